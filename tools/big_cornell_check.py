"""The headline scene at any size, GPU (fused tiny-scene kernels and general BVH kernels) vs the oracle, bit for bit.
usage: python tools/big_cornell_check.py 1920 1080 64 [cornell|sponza|bistro] [shard nshards]   (the full configs[1] frame: the oracle needs ~10 s on 16 threads)"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as g
rt = g.load_package(); orc = g.load_oracle()
from test_gpu_parity import bits
W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "cornell"          # cornell | sponza | bistro
shard, nshards = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 1)   # one rank's pixel tiles only
sc = {"cornell": rt.Scene.cornell, "sponza": rt.Scene.sponza_class, "bistro": rt.Scene.bistro_class}[kind]()
p = rt.Params(width=W, height=H, spp=spp, max_bounces=8, nee_samples=1, rr_start=3, flags=0 if kind == "bistro" else 1, frame_seed=5,
              tile_size=64, shard_rank=shard, shard_count=nshards)
o = orc.Oracle().load(sc, W / H); o.set_threads(16)
t0 = time.time(); oa, oc = o.render(p); print("oracle %.1f s, rays %s" % (time.time() - t0, oc), flush=True)
for small in ((1, 0) if kind == "cornell" else (0,)):
    c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.upload(sc, W / H)
    c.clear(W, H); c.render(p); st = c.stats(); im = c.read_accum()
    d = (bits(im) != bits(oa)).any(-1)
    print("small", small, "diff px", int(d.sum()), "of", W * H, "counts equal", (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, flush=True)
    for (y, x) in np.argwhere(d)[:5]: print("   px", x, y, im[y, x], oa[y, x])
    c.close()
