"""Commit times of the large procedural scenes on the GPU box, host builder against RTX_OPT_GPU_BUILD: python tools/build_time.py [hard=1] [frames=3]
   first commit       scene hand-over (rtx_set_materials / add_mesh / add_instance) + rtx_commit_scene: flatten + shade records + lights on the host, the tree, uploads
   geometry change    one small mesh added to the resident scene + rtx_commit_scene: everything is re-derived (what a deforming or streamed-in mesh costs per change)
   frame              BASELINE's frame of that scene on the tree just built (1080p, 16 spp, 8 bounces), ms — the tree's quality as the renderer sees it
RTX_BUILD_TIMES=1 in the environment prints the phases of every commit to stderr."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as g  # noqa: E402
rt = g.load_package()
args = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
hard, frames = int(args.get("hard", 0)), int(args.get("frames", 3))
W, H = 1920, 1080
print("| scene | builder | first commit s | geometry-change commit s | GPU build phases ms (boxes + keys, sort, PLOC, top on the host, layout) | PLOC rounds / clusters to the host | wide nodes | frame ms (min of %d) |" % frames)
print("|---|---|---|---|---|---|---|---|")
for kind in ("sponza", "bistro"):
    sc = (rt.Scene.sponza_class if kind == "sponza" else rt.Scene.bistro_class)(hard=bool(hard))
    flags = 1 if kind == "sponza" else 4
    p = rt.Params(width=W, height=H, spp=16, max_bounces=8, nee_samples=1, rr_start=3, flags=flags)
    quad = np.array([[-0.1, 0.5, -0.1], [0.1, 0.5, -0.1], [0.1, 0.5, 0.1], [-0.1, 0.5, -0.1], [0.1, 0.5, 0.1], [-0.1, 0.5, 0.1]], np.float32)
    v = np.zeros((6, 7), np.float32); v[:, 0:3] = quad; v[:, 6] = sum(len(m[2]) for m in sc.meshes)
    for gpu in (0, 1):
        c = rt.Context(0); c.set_option(rt.OPT_GPU_BUILD, gpu)
        t0 = time.perf_counter(); c.upload(sc, W / H); first = time.perf_counter() - t0
        info = c.build_info()
        c.clear(W, H); c.render(p)
        ms = []
        for k in range(frames):
            c.clear(W, H); t0 = time.perf_counter(); c.render(p); ms.append((time.perf_counter() - t0) * 1e3)
        mesh = c.add_mesh(v, np.arange(6, dtype=np.uint32), np.full(6, 1, np.uint32)); c.add_instance(mesh, np.eye(4, dtype=np.float32).reshape(16))
        t0 = time.perf_counter(); c.commit(); change = time.perf_counter() - t0
        info2 = c.build_info()
        print(f"| {kind}{' (hard)' if hard else ''} {sc.num_triangles} triangles | {'GPU (RTX_OPT_GPU_BUILD)' if gpu else 'host'} | {first:.3f} | {change:.3f} | "
              f"{' '.join('%.1f' % x for x in info2['ms']) if gpu else '-'} | {('%d / %d' % (info2['ploc_rounds'], info2['clusters_top'])) if gpu else '-'} | {info['nodes']} | {min(ms):.2f} |", flush=True)
        c.close()
