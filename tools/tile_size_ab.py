"""Frame time of BASELINE's single-GPU frames against rtx_params.tile_size (the path-slot -> pixel mapping of an unsharded frame): python tools/tile_size_ab.py [sizes=32,64,128,256]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, __graft_entry__ as g
rt = g.load_package()
args = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
sizes = [int(x) for x in args.get("sizes", "32,64,128,256").split(",")]
W, H = 1920, 1080
for name, sc, spp, flags in (("cornell", rt.Scene.cornell(), 64, 1), ("sponza", rt.Scene.sponza_class(), 16, 1), ("bistro", rt.Scene.bistro_class(), 16, 4)):
    c = rt.Context(0); c.upload(sc, W / H)
    for rep in range(2):
        for ts in sizes:
            p = rt.Params(width=W, height=H, spp=spp, max_bounces=8, nee_samples=1, rr_start=3, flags=flags, tile_size=ts)
            c.clear(W, H); c.render(p)
            ms = []
            for k in range(4):
                c.render(p); ms.append(c.stats().render_ms)
            print(f"{name} tile {ts}: {min(ms):.3f} ms (min of 4, events) median {sorted(ms)[2]:.3f}", flush=True)
    c.close()
