"""Frame time of a BVH workload against the visiting order of any-hit rays (RTX_OPT_ANYHIT_ORDER: -1 = what the commit-time probe chose): python tools/anyorder_ab.py [hard=1] [gpu=1]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import __graft_entry__ as g
rt = g.load_package()
args = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
hard, gpu = int(args.get("hard", 0)), int(args.get("gpu", 0))
W, H = 1920, 1080
for name in ("sponza", "bistro"):
    sc = (rt.Scene.sponza_class if name == "sponza" else rt.Scene.bistro_class)(hard=bool(hard))
    c = rt.Context(0); c.set_option(rt.OPT_GPU_BUILD, gpu); c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=16, max_bounces=8, nee_samples=1, rr_start=3, flags=4 if name == "bistro" else 1)
    for order in (-1, 0, 1, 2, -1):
        c.set_option(rt.OPT_ANYHIT_ORDER, order)
        c.clear(W, H); c.render(p); c.render(p)
        ms = []
        for k in range(4):
            c.render(p); ms.append(c.stats().render_ms)
        print(f"{name}{' hard' if hard else ''}{' (GPU-built)' if gpu else ''} any-hit order {order if order >= 0 else 'probe'}: min {min(ms):.3f} median {sorted(ms)[2]:.3f} ms", flush=True)
    c.close()
