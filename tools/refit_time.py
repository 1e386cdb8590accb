"""Wall time of transform-only commits: GPU refit vs host refit + re-collapse + upload.  python tools/refit_time.py"""
import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
for name, sc in (('sponza 262k', rt.Scene.sponza_class()), ('bistro 3.8M', rt.Scene.bistro_class())):
    for mode in (1, 0):
        c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, mode)
        t0 = time.time(); c.upload(sc, 16/9); t_build = time.time() - t0
        ts = []
        for k in range(3):
            m = np.eye(4, dtype=np.float32); m[3, 0] = 0.01 * (k + 1)
            c.set_instance_transform(0, m.reshape(16)); t0 = time.time(); c.commit(); ts.append(time.time() - t0)
        print(name, 'gpu_refit' if mode else 'host_refit', 'build+upload %.2f s' % t_build, 'refit commits (s):', [round(t, 4) for t in ts], 'valid', c.validate_bvh())
        c.close()
