"""Frame time of the reference's ReSTIR pipeline (pass 1 + 2 + 3) at 1080p on three scenes.  python tools/restir_time.py"""
import sys, time, os; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
gd = 'tests/golden'
scenes = {'cornell': rt.Scene.cornell(), 'garage': rt.Scene.from_obj([os.path.join(gd, "garage.obj"), os.path.join(gd, "monke.obj")], gd + "/"), 'sponza': rt.Scene.sponza_class()}
W, H = 1920, 1080
for name, sc in scenes.items():
    c = rt.Context(0); c.set_option(rt.OPT_KERNEL_TIMING, 1); c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0)
    c.restir_reset(); c.clear(W, H)
    for f in range(4):
        t0 = time.time(); c.render_restir(p.copy(frame_seed=10 + f)); dt = time.time() - t0
        st = c.stats()
        print(name, 'frame', f, 'wall %.2f ms' % (dt * 1e3), 'render_ms %.2f' % st.render_ms, 'rays', st.rays_primary, st.rays_extension, st.rays_shadow, 'Mrays/s %.0f' % ((st.rays_primary + st.rays_extension + st.rays_shadow) / st.render_ms / 1e3))
    c.close()
