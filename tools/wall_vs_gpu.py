import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, __graft_entry__ as g
rt = g.load_package()
W, H = 1920, 1080
for name, sc, spp, flags in (("cornell", rt.Scene.cornell(), 64, 1), ("sponza", rt.Scene.sponza_class(), 16, 1)):
    c = rt.Context(0); c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=spp, max_bounces=8, nee_samples=1, rr_start=3, flags=flags)
    c.clear(W, H); c.render(p); c.render(p)
    for k in range(5):
        t0 = time.perf_counter(); c.render(p); wall = (time.perf_counter() - t0) * 1e3
        print(name, "wall %.3f ms  render_ms (events) %.3f  host overhead %.3f" % (wall, c.stats().render_ms, wall - c.stats().render_ms), flush=True)
    c.close()
