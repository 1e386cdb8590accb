"""Generates tests/golden/ggx_table.json: the 64-entry (N, V, L, material, seed) table of SURVEY 8(c) item 4 with the values the float64
restatement of the reference's HLSL (tests/ggx_ref64.py — written from GGX_v6.hlsl / BRDF_v6.hlsl / Lambertian_v6.hlsl, not from the oracle)
gives for them.  Run from the repository root:  python tests/golden/make_ggx_table.py
The Ess LUT of each material comes from the product's deterministic host generator (rtxh_generate_ess_lut); it is INPUT data here."""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft      # noqa: E402
import ggx_ref64 as R                # noqa: E402


def unit(v):
    return v / np.linalg.norm(v)


def hemi(rng, n, cmin):
    while True:
        d = unit(rng.normal(size=3))
        if np.dot(d, n) >= cmin:
            return d


def expected(e):
    m = R.Mat(e["Kd"], e["Ks"], e["roughness"], e["metallic"], e["lut"])
    N, V, L = (np.array(e[k], np.float32).astype(np.float64) for k in ("N", "V", "L"))
    F, P, pd, ps = R.mixture(m, N, L, V)
    s0, s1 = e["seed"]
    r, s0, s1 = R.tea(s0, s1)
    strat = int(R.select_strategy(m, V, N, r))
    u1, s0, s1 = R.tea(s0, s1)
    u2, s0, s1 = R.tea(s0, s1)
    wi = R.sample_ggx(m, V, N, u1, u2)[0] if strat == 1 else R.sample_lambert(N, u1, u2)
    return {"F": [float(x) for x in F], "P": float(P), "p_d": float(pd), "p_s": float(ps), "r": r, "u1": u1, "u2": u2,
            "strategy": strat, "wi": [float(x) for x in wi], "seed_out": [s0, s1],
            "D": float(R.d_ggx(R.dot(N, R.normalize(V + L)), m.Pr)), "G1": float(R.g1_smith(R.dot(N, V), m.Pr * m.Pr)),
            "G2": float(R.g2_smith(R.dot(N, V), R.dot(N, L), m.Pr * m.Pr)), "f_ggx": [float(x) for x in R.ggx_eval(m, N, L, V)],
            "pdf_ggx": float(R.ggx_pdf(m, N, L, V))}


def main():
    rt = graft.load_package()
    rng = np.random.default_rng(20261003)
    entries = []
    rough = [0.1, 0.15, 0.25, 0.35, 0.5, 0.65, 0.8, 1.0]
    for i in range(64):
        N = unit(rng.normal(size=3)) if i % 4 else np.array([0.0, 0.0, 1.0] if i % 8 else [0.0, 1.0, 0.0])   # also the CoordinateSystem branch |N.z| >= 0.999
        V, L = hemi(rng, N, 0.1), hemi(rng, N, 0.05)
        r = rough[i % 8] if i < 56 else 0.03                          # the last 8: roughness < 0.04 -> strategy 0 even when r <= p_s
        e = {"N": [float(np.float32(x)) for x in N], "V": [float(np.float32(x)) for x in V], "L": [float(np.float32(x)) for x in L],
             "Kd": [float(x) for x in rng.uniform(0.05, 0.95, 3)], "Ks": [float(x) for x in rng.uniform(0.0, 1.0, 3)] if i % 5 else [0.0, 0.0, 0.0],
             "roughness": r, "metallic": [0.0, 0.0, 0.5, 1.0][i % 4], "seed": [int(x) for x in rng.integers(0, 2 ** 32, 2)]}
        e["lut"] = [float(x) for x in rt.generate_ess_lut(float(np.float32(np.float16(r))))]
        e["expect"] = expected(e)
        entries.append(e)
    with open(os.path.join(ROOT, "tests", "golden", "ggx_table.json"), "w") as f:
        json.dump({"note": "inputs + float64 values of tests/ggx_ref64.py (restated from the reference's HLSL text); see make_ggx_table.py", "entries": entries}, f, indent=0)
    print("wrote", len(entries), "entries; strategies:", [e["expect"]["strategy"] for e in entries])


if __name__ == "__main__":
    main()
