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


# ---- one SMALL instance moving inside a large static scene (the reference's own loop: Renderer.cpp:444-452): partial refit (RTX_OPT_PARTIAL_REFIT, default) vs full
import os
class _Two:
    def __init__(self, big, small, place):
        nm = len(big.materials)
        self.materials = np.concatenate([np.asarray(big.materials, np.float32), np.asarray(small.materials, np.float32)])
        self.meshes = list(big.meshes); base = sum(len(m) for _, _, m in big.meshes)
        for v, i, m in small.meshes:
            v = np.array(v, np.float32, copy=True).reshape(-1, 7); v[:, 6] = float(base)
            self.meshes.append((v, i, np.asarray(m, np.uint32) + np.uint32(nm))); base += len(m)
        self.instances = list(big.instances) + [(len(big.meshes) + mesh, place) for mesh, _ in small.instances]
        self._big = big
    def view_proj(self, aspect): return self._big.view_proj(aspect)
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
monke = rt.Scene.from_obj([os.path.join(gold, 'monke.obj')], gold + '/')
for name, big in (('sponza 262k + monke', rt.Scene.sponza_class()), ('bistro 3.8M + monke', rt.Scene.bistro_class())):
    sc = _Two(big, monke, np.eye(4, dtype=np.float32).reshape(16))
    for partial in (1, 0):
        c = rt.Context(0); c.set_option(rt.OPT_PARTIAL_REFIT, partial); c.upload(sc, 16 / 9)
        ts = []
        for k in range(6):
            m = np.eye(4, dtype=np.float32); m[3, 1] = 0.01 * (k + 1)
            c.set_instance_transform(len(sc.instances) - 1, m.reshape(16)); t0 = time.time(); c.commit(); ts.append(time.time() - t0)
        print(name, 'partial' if partial else 'full   ', 'refit commits (ms):', [round(t * 1e3, 3) for t in ts], 'valid', c.validate_bvh())
        c.close()
