"""Does running two half-frames concurrently (two contexts, two streams, two host threads) beat one whole frame?  If the HBM-bound shading kernels of one half overlap
the VALU-bound traversal kernels of the other, the pair finishes sooner than the sum.  usage: python tools/overlap_probe.py [sponza|bistro]"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
rt = g.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "sponza"
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H, spp, flags = 1920, 1080, 16, (1 if kind == "sponza" else 4)
def mk():
    c = rt.Context(0); c.upload(sc, W / H); c.clear(W, H); return c
one = mk()
p = rt.Params(width=W, height=H, spp=spp, max_bounces=8, nee_samples=1, flags=flags)
one.render(p)
t = []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); one.render(p); t.append(time.perf_counter() - t0)
print(kind, "one context, %d spp: %.2f ms" % (spp, 1e3 * min(t)))
a, b = mk(), mk()
ph = [rt.Params(width=W, height=H, spp=spp // 2, sample_base=1, max_bounces=8, nee_samples=1, flags=flags), rt.Params(width=W, height=H, spp=spp // 2, sample_base=1 + spp // 2, max_bounces=8, nee_samples=1, flags=flags)]
for c, q in ((a, ph[0]), (b, ph[1])): c.render(q)
t = []
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=c.render, args=(q,)) for c, q in ((a, ph[0]), (b, ph[1]))]
    for x in th: x.start()
    for x in th: x.join()
    t.append(time.perf_counter() - t0)
print(kind, "two contexts x %d spp concurrently: %.2f ms" % (spp // 2, 1e3 * min(t)))
t = []
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a.render(ph[0]); b.render(ph[1]); t.append(time.perf_counter() - t0)
print(kind, "two contexts x %d spp one after the other: %.2f ms" % (spp // 2, 1e3 * min(t)))
