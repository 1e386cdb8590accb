"""Frame time of the reference's ReSTIR pipeline (pass 1 + 2 + 3) at 1080p.
python tools/restir_time.py [cornell|garage|sponza ...] [frames=N] [option_id=value ...]
Under `rocprofv3 --kernel-trace --stats -- python3 tools/restir_time.py sponza` the kernel table is that scene's alone."""
import sys, time, os; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
gd = 'tests/golden'
ctors = {'cornell': lambda: rt.Scene.cornell(),
         'garage': lambda: rt.Scene.from_obj([os.path.join(gd, "garage.obj"), os.path.join(gd, "monke.obj")], gd + "/"),
         'sponza': lambda: rt.Scene.sponza_class(), 'bistro': lambda: rt.Scene.bistro_class()}
names = [a for a in sys.argv[1:] if a in ctors] or ['cornell', 'garage', 'sponza']
frames = ([int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('frames=')] or [4])[0]
opts = [a.split('=') for a in sys.argv[1:] if '=' in a and a.split('=')[0].isdigit()]
W, H = 1920, 1080
for name in names:
    sc = ctors[name]()
    c = rt.Context(0)
    for k, v in opts:
        c.set_option(int(k), int(v))
    c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0)
    c.restir_reset(); c.clear(W, H)
    for f in range(frames):
        t0 = time.time(); c.render_restir(p.copy(frame_seed=10 + f)); dt = time.time() - t0
        st = c.stats()
        print(name, 'frame', f, 'wall %.2f ms' % (dt * 1e3), 'render_ms %.2f' % st.render_ms, 'rays', st.rays_primary, st.rays_extension, st.rays_shadow, 'Mrays/s %.0f' % ((st.rays_primary + st.rays_extension + st.rays_shadow) / st.render_ms / 1e3), flush=True)
    c.close()
