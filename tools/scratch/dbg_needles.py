import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import __graft_entry__ as g
rt = g.load_package(); orc = g.load_oracle()
import test_gpu_parity as T
kind = "needles"
rng = np.random.default_rng(sum(map(ord, kind)) + 7)
t = T.soup(kind, 6000, rng)
ray = np.array([[1.6235447e+00, -6.5197051e-01, 1.3057612e+00, 9.9999997e-06, -8.8360572e-01, -3.7004687e-02, -4.6676713e-01, 1.0e+30]], np.float32)
# recover exact ray: regenerate as in the test
sc = T.SoupScene(t)
lo, hi = t.reshape(-1, 3).min(0), t.reshape(-1, 3).max(0); ext = float((hi - lo).max()); m = 60000
org = rng.uniform(lo - 0.1 * ext, hi + 0.1 * ext, (m, 3))
d = rng.normal(size=(m, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
k = m // 3
pick = rng.integers(0, len(t), k); w = rng.dirichlet((1, 1, 1), k); w[: k // 4] = np.eye(3)[rng.integers(0, 3, k // 4)]
w[k // 4: k // 2, 2] = 0; w[k // 4: k // 2, :2] /= np.maximum(w[k // 4: k // 2, :2].sum(1, keepdims=True), 1e-9)
tgt = (t[pick] * w[:, :, None]).sum(1)
d[:k] = tgt - org[:k]; d[:k] /= np.maximum(np.linalg.norm(d[:k], axis=1, keepdims=True), 1e-30)
d[k: k + 2000] = np.eye(3)[rng.integers(0, 3, 2000)] * rng.choice([-1.0, 1.0], (2000, 1))
rays = np.zeros((m, 8), np.float32)
rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = org, 1e-5, d, 1e30
o = orc.Oracle().load(sc, 1.0)
c = rt.Context(0); c.upload(sc, 1.0)
gh = c.trace_closest(rays); bh = o.trace_closest(rays, mode=0)
bad = np.nonzero(T.bits(gh)[:, 3] != T.bits(bh)[:, 3])[0]
print("bad rays", bad)
for i in bad:
    r = rays[i:i+1]
    print("ray", r.tolist(), "gpu", gh[i], T.bits(gh)[i, 3], "cpu", bh[i], T.bits(bh)[i, 3], "cpu bvh", o.trace_closest(r, mode=1))
    pid = int(T.bits(bh)[i, 3])
    print("tri", t[pid].tolist())
    # only that triangle (+ neighbours in id) on the GPU
    for sub in ([pid], list(range(pid - 3, pid + 4)), list(range(max(0, pid - 500), pid + 500))):
        s2 = T.SoupScene(t[sub]); c2 = rt.Context(0); c2.set_option(rt.OPT_SMALL_SCENE, 0); c2.upload(s2, 1.0)
        o2 = orc.Oracle().load(s2, 1.0)
        print(len(sub), "gpu", c2.trace_closest(r), "cpu", o2.trace_closest(r, mode=0)); c2.close()
