#!/usr/bin/env python3
"""Does any result depend on the ORDER in which the persistent traversal kernels work?  (GPU tool; the evidence behind the determinant floor of the hit definition.)

A bit-exact renderer must return the same image, ray counts and hit records however its waves are scheduled.  Which rays share a wave of k_trace_closest / k_trace_shadow depends
on timing (the four waves of a workgroup draw from one LDS cursor), and the speculative schedule lets the wave's vote decide when a lane's pending triangles shorten its ray —
i.e. which boxes it still visits.  A "hit" that only some visiting orders find (float Moeller-Trumbore's 0 / 0 for a ray in a triangle's plane) therefore shows up as a rare,
non-reproducible mismatch.  This tool makes it reproducible: every random mid-size scene (tests/test_gpu_parity.py: RandomTinyScene, 65 .. 3000 triangles, slivers included) is
rendered through the general BVH path in several configurations that change nothing but the order of work — the default twice, refill thresholds 1 and 48, 8 sub-queues per CU,
the while-while and the voted wave schedules instead of the speculative one — and the image bits, the three ray counts and 4 500 closest-hit records are compared across them.

usage: python3 tools/order_fuzz.py [scenes=400] [seed=200000] [w=96 h=64 spp=4] [lib=<path of a librtx_hip variant>]
  make -C royaltracer-dx_amd VARIANT=nofloor VARFLAGS=-DRTX_DET_REL=0.0f   builds the hit definition of rounds 1-4 (det != 0) for the A/B: lib=royaltracer-dx_amd/librtx_hip_nofloor.so
Exit code 1 if any scene differs between two configurations."""
import hashlib, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
if "lib" in args:
    os.environ["RTX_LIB_PATH"] = os.path.abspath(args["lib"])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
nscenes, seed0 = int(args.get("scenes", 400)), int(args.get("seed", 200000))
W, H = int(args.get("w", 96)), int(args.get("h", 64))
CONFIGS = [("default", {}), ("default again", {}), ("refill at 1 idle lane", {rt.OPT_REFILL_MIN: 1}), ("refill at 48, 8 sub-queues per CU", {rt.OPT_REFILL_MIN: 48, rt.OPT_BLOCKS_PER_CU: 8}),
           ("while-while schedule", {rt.OPT_TRACE_SCHED: 0}), ("voted schedule", {rt.OPT_TRACE_SCHED: 2}), ("no shadow overlap, no taper", {rt.OPT_OVERLAP_SHADOW: 0, rt.OPT_TAPER: 0})]
bad, t0 = [], time.time()
for s in range(nscenes):
    sc = tgp.RandomTinyScene(rt, 9000 + seed0 + s, max_tris=[200, 800, 3000][s % 3])
    p = rt.Params(width=W, height=H, spp=int(args.get("spp", 4)), max_bounces=5, nee_samples=1 + s % 2, flags=s & 1, frame_seed=s)
    rays = np.concatenate([tgp.random_rays(3000, s, -1.2, 1.2), tgp.random_rays(1500, s + 7, -0.4, 0.4)])
    sh = rays.copy(); sh[:, 7] = np.random.default_rng(s).uniform(0.05, 2.0, len(sh)).astype(np.float32)
    sigs = []
    for name, opts in CONFIGS:
        c = rt.Context(0)
        for k, v in opts.items():
            c.set_option(k, v)
        c.upload(sc, W / H); c.clear(W, H); c.render(p); st = c.stats()
        h = hashlib.sha1(c.read_accum().tobytes()); h.update(repr((st.rays_primary, st.rays_extension, st.rays_shadow)).encode())
        h.update(c.trace_closest(rays).tobytes()); h.update(np.asarray(c.trace_any(sh)).tobytes())
        sigs.append(h.hexdigest()); c.close()
    if len(set(sigs)) > 1:
        bad.append((s, [CONFIGS[i][0] for i in range(len(sigs)) if sigs[i] != sigs[0]]))
        print(f"scene {s}: differs from the default in {bad[-1][1]}", flush=True)
    if s % 100 == 99:
        print(f"  {s + 1} scenes, {len(bad)} order-dependent, {time.time() - t0:.0f} s", flush=True)
print(f"order_fuzz: lib {os.path.basename(rt.LIB_PATH)}, {nscenes} scenes from seed {seed0}, {len(CONFIGS)} configurations each: {len(bad)} scenes whose results depend on the order of work {bad[:20]}")
sys.exit(1 if bad else 0)
