#!/usr/bin/env python3
"""Which call changes host-side state it should not?  The flow of tools/flaky_validate.py with rtx_debug_host_checksums after every API call: between two commits nothing the
host holds (mesh indices and vertices, materials, the leaf order, the binary tree, the wide mirror, shade records) may change; the first three never after the hand-over.
usage: python3 tools/flaky_bisect.py [reps=4000] [seeds=817,148]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
args = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
reps, seeds = int(args.get("reps", 4000)), [int(x) for x in args.get("seeds", "817,148").split(",")]
NAMES = ("mesh indices", "mesh vertices", "materials", "instances", "leaf order", "binary tree", "wide mirror", "shade + objtris + slots")
W, H = 48, 32
scenes = {}
for seed in seeds:
    sc = tgp.RandomTinyScene(rt, 9000 + 200000 + seed, max_tris=[200, 800, 3000][seed % 3])
    M = np.eye(4); M[:3, :3] = np.diag([1.1, 0.9, -1.05]) @ np.array([[np.cos(.3), 0, np.sin(.3)], [0, 1, 0], [-np.sin(.3), 0, np.cos(.3)]]); M[:3, 3] = (0.05, -0.02, 0.03)
    inst = len(sc.instances) - 1
    M2 = (M @ np.asarray(sc.instances[inst][1], np.float64).reshape(4, 4).T).T.astype(np.float32).reshape(16)
    scenes[seed] = (sc, inst, M2, tgp.random_rays(3000, seed, -1.2, 1.2))
found, t0, ref = 0, time.time(), {}
for rep in range(reps):
    for seed in seeds:
        sc, inst, M2, rays = scenes[seed]
        p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1 + seed % 2, flags=seed & 1, frame_seed=seed)
        c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, 0)
        trail = []
        def mark(label, frozen=range(8)):
            h = c.host_checksums()
            if trail:
                changed = [NAMES[k] for k in frozen if h[k] != trail[-1][1][k]]
                if changed:
                    global found; found += 1
                    print(f"rep {rep} seed {seed}: {changed} changed during '{label}' (previous mark: '{trail[-1][0]}')", flush=True)
            trail.append((label, h))
        c.upload(sc, W / H); mark("upload + commit", ())
        key = (seed, "built")
        if key not in ref: ref[key] = trail[-1][1]
        elif ref[key] != trail[-1][1]: found += 1; print(f"rep {rep} seed {seed}: state after the first commit differs from the first run's in {[NAMES[k] for k in range(8) if ref[key][k] != trail[-1][1][k]]}", flush=True)
        c.clear(W, H); mark("clear")
        c.render(p); mark("render")
        c.trace_closest(rays); mark("trace_closest")
        c.validate_bvh(); mark("validate")
        c.stats(); mark("stats")
        c.set_instance_transform(inst, M2); mark("set_instance_transform", (0, 1, 2, 4, 5, 6, 7))
        c.commit(); mark("commit (host refit)", (0, 1, 2, 3, 4))
        key = (seed, "refit")
        if key not in ref: ref[key] = trail[-1][1]
        elif ref[key] != trail[-1][1]: found += 1; print(f"rep {rep} seed {seed}: state after the refit differs from the first run's in {[NAMES[k] for k in range(8) if ref[key][k] != trail[-1][1][k]]}", flush=True)
        c.clear(W, H); c.render(p); mark("render 2")
        c.validate_bvh(); c.tree_hash(); mark("validate + hash")
        c.close()
    if rep % 500 == 499: print(f"  {rep + 1} reps x {len(seeds)} scenes, {found} findings, {time.time() - t0:.0f} s", flush=True)
print(f"flaky_bisect: {reps} reps x {len(seeds)} scenes: {found} findings")
