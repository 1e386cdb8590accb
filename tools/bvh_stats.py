"""Node steps and triangle tests per ray of the resident wide BVH (rtx_debug_trace_stats) for primary and secondary rays.
Run on a GPU box: python tools/bvh_stats.py"""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
import test_gpu_parity as T
for name, sc in (('sponza', rt.Scene.sponza_class()), ('bistro', rt.Scene.bistro_class())):
    c = rt.Context(0); c.upload(sc, 16/9)
    p = rt.Params(width=480, height=270)
    prim = c.primary_rays(p)
    h = c.trace_closest(prim); hit = T.bits(h)[:,3] != 0xFFFFFFFF
    rng = np.random.default_rng(1)
    sec = T.random_rays(int(hit.sum()), 4); sec[:, 0:3] = prim[hit, 0:3] + h[hit, 0:1] * prim[hit, 4:7]; sec[:, 3] = 2e-5
    # flip directions into the hemisphere facing the camera-ish: keep as is (random)
    for label, rays in (('primary', prim), ('secondary', sec)):
        s = c.trace_stats(rays)
        assert np.array_equal(T.bits(s)[:, 3], T.bits(c.trace_closest(rays))[:, 3])
        print(name, label, 'rays', len(rays), 'node steps/ray %.2f' % s[:,1].mean(), 'tri tests/ray %.2f' % s[:,2].mean(), 'p90 steps', np.percentile(s[:,1], 90), 'max', s[:,1].max(), 'hit frac %.2f' % (T.bits(s)[:,3] != 0xFFFFFFFF).mean(), 'nodes', c.stats().bvh_nodes)
    c.close()
