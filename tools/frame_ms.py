"""Untimed frames (no per-kernel events: the shadow launches overlap the next closest-hit launch) of BASELINE's BVH workloads: python tools/frame_ms.py [sponza|bistro ...] [hard=1] [frames=6] [option_id=value ...]   (38=1: GPU-built tree; 39=0: the whole traversal stack in LDS)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import __graft_entry__ as g
rt = g.load_package()
names = [a for a in sys.argv[1:] if a in ("sponza", "bistro", "cornell")] or ["sponza", "bistro"]
frames = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("frames=")] or [6])[0]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a and a.split("=")[0].isdigit()]
hard = any(a == "hard=1" for a in sys.argv[1:])
W, H = 1920, 1080
for name in names:
    sc = rt.Scene.cornell() if name == "cornell" else rt.Scene.sponza_class(hard=hard) if name == "sponza" else rt.Scene.bistro_class(hard=hard)
    c = rt.Context(0)
    for k, v in opts: c.set_option(int(k), int(v))
    c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=64 if name == "cornell" else 16, max_bounces=8, nee_samples=1, rr_start=3, flags=4 if name == "bistro" else 1)
    c.clear(W, H); c.render(p); c.render(p)
    ms = []
    for k in range(frames):
        c.render(p); ms.append(c.stats().render_ms)
    print(f"{name}{' hard' if hard else ''} {' '.join('='.join(o) for o in opts)} stack bound {c.build_info()['nodes'] and c.stats().bvh_nodes} nodes: min {min(ms):.3f} median {sorted(ms)[len(ms) // 2]:.3f} ms", flush=True)
    c.close()
