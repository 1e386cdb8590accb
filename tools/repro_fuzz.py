import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["RTX_FUZZ_SEED"] = "200000"
import numpy as np
import __graft_entry__ as g
rt = g.load_package()
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
sys.path.insert(0, os.path.join(ROOT, 'oracle')); import orc
bits = tgp.bits
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 817
W, H = 48, 32
sc = tgp.RandomTinyScene(rt, 9000 + 200000 + seed, max_tris=[200, 800, 3000][seed % 3])
p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1 + seed % 2, flags=seed & 1, frame_seed=seed)
o = orc.Oracle().load(sc, W / H); oa, oc = o.render(p)
rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H), 1), tgp.random_rays(3000, seed, -1.2, 1.2)])
ob = o.trace_closest(rays, 1)
M = np.eye(4); M[:3, :3] = np.diag([1.1, 0.9, -1.05]) @ np.array([[np.cos(.3), 0, np.sin(.3)], [0, 1, 0], [-np.sin(.3), 0, np.cos(.3)]]); M[:3, 3] = (0.05, -0.02, 0.03)
inst = len(sc.instances) - 1
M2 = (M @ np.asarray(sc.instances[inst][1], np.float64).reshape(4, 4).T).T.astype(np.float32).reshape(16)
o2 = orc.Oracle().load(sc, W / H); o2.set_instance_transform(inst, M2); oa2, oc2 = o2.render(p)
ob2 = o2.trace_closest(rays, 1)
for refit in (1, 0):
    for bvh in ("", "threads=1"):
        if bvh: rt.bvh_option("threads", 1)
        else: rt.bvh_option("threads", 0)
        c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, refit); c.upload(sc, W / H)
        c.clear(W, H); c.render(p); st = c.stats()
        a = np.array_equal(bits(c.read_accum()), bits(oa)); b = (st.rays_primary, st.rays_extension, st.rays_shadow) == oc; d = np.array_equal(bits(c.trace_closest(rays)), bits(ob))
        c.set_instance_transform(inst, M2); c.commit()
        v = c.validate_bvh()
        g2 = c.trace_closest(rays)
        nd = int((bits(g2) != bits(ob2)).any(axis=1).sum())
        c.clear(W, H); c.render(p); st2 = c.stats()
        e = np.array_equal(bits(c.read_accum()), bits(oa2)); f = (st2.rays_primary, st2.rays_extension, st2.rays_shadow) == oc2
        print("refit", refit, bvh, "tris", st.triangles, "before: img", a, "counts", b, "closest", d, "| after: validate", v, "closest diffs", nd, "img", e, "counts", f, st2.rays_shadow, oc2)
        if nd:
            w = np.nonzero((bits(g2) != bits(ob2)).any(axis=1))[0][:5]
            for i in w: print("  ray", i, rays[i], "gpu", g2[i], bits(g2)[i, 3], "oracle", ob2[i], bits(ob2)[i, 3])
        c.close()
