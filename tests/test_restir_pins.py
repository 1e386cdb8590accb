"""Independent pins of the ReSTIR passes — 2 and 3 since round 3, pass 1 (SampleRIS + SamplePathSimple) since round 4 (VERDICT r02 "what's weak" 1 / r03 item 6): until then the temporal and spatial passes — pairwise MIS, the
reconnection Jacobian, the reject predicates, the reservoir merges — were pinned only by "GPU == oracle".  tests/restir_ref64.py restates both passes in float64 numpy
from the HLSL text (not from the oracle); here it re-computes EVERY pixel of a frame of the oracle from the oracle's own input buffers and must arrive at the same
reservoirs and radiance.  The GPU reproduces these oracle buffers byte for byte (tests/test_gpu_parity.py), so the pin carries over to the kernels."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ggx_ref64 as R          # noqa: E402
import restir_ref64 as X       # noqa: E402


def _col(m):
    return np.asarray(m, np.float64).reshape(4, 4).T           # 16 floats, element (r, c) at m[c * 4 + r]


def _rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-30)


@pytest.fixture(scope="module")
def frames(rt, orc, golden_dir):
    """three ReSTIR frames of the oracle on garage.obj + monke.obj (GGX + Lambert, two instances) with a moving camera; of the third frame: the pass-1 output, the history
    it started from, the buffers after passes 2 / 3 and the radiance it added"""
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 48, 28
    o = orc.Oracle().load(sc, W / H)
    proj = rt.perspective_fov_rh(np.float32(np.pi / 3), W / H, 0.1, 1000.0)
    views = [rt.lookat(e, (0, 1, 0), (0, 1, 0)) for e in [(-1.5, 1.5, 3.5), (-1.45, 1.5, 3.5), (-1.4, 1.52, 3.48)]]
    acc, st, hist, p1, before = np.zeros((H, W, 4), np.float32), None, None, None, None
    for k, v in enumerate(views):
        o.set_camera(v, proj)
        p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=31 + k)
        if k == 2:
            hist, before = tuple(b.copy() for b in st[3:6]), acc.copy()
            _, p1, _ = o.render_v6_pass1(p)
        acc, st, _ = o.restir_frames(p, acc, st)

    def visible(x1, n1, x2):                                   # VisibilityCheck (Sampler_v6.hlsl:86-104) answered by the oracle's any-hit query
        d = x2 - x1; dist = np.sqrt((d * d).sum()); org = x1 + X.normalize(n1) * X.S_BIAS
        ray = np.array([[*org, 0.0, *(d / dist if dist > 0 else d), max(dist - 10 * X.S_BIAS, 2 * X.S_BIAS)]], np.float32)
        return 0.0 if o.trace_any(ray)[0] else 1.0
    def closest(org, d):                                       # TraceRay + ClosestHit of a TMin = s_bias ray, answered by the oracle's closest-hit query and its ClosestHit restatement
        ray = np.array([[*org, X.S_BIAS, *d, 10000.0]], np.float32)
        hit = o.trace_closest(ray, 1); sf = o.surface(ray, hit)[0]
        mid = int(sf[3:4].view(np.uint32)[0])
        return dict(hit=mid != 0xFFFFFFFE, pos=sf[0:3].astype(np.float64), normal=sf[4:7].astype(np.float64), mID=mid)

    def occluded(org, d, tmin, tmax):                          # the shadow ray type (ShadowRay.hlsl)
        return bool(o.trace_any(np.array([[*org, tmin, *d, tmax]], np.float32))[0])
    lights = [X.Light(r) for r in o.lights()]
    return dict(W=W, H=H, seed=33, cam=np.linalg.inv(_col(views[2]))[:3, 3], prev_view=_col(views[1]), proj=_col(proj), closest=closest, occluded=occluded, lights=lights,
                ke=[np.asarray(m[8:11], np.float32).astype(np.float64) for m in sc.materials], o2w=[_col(m) for _, m in sc.instances],
                inst=[(np.linalg.inv(_col(m)), _col(m)) for _, m in sc.instances], mats=[R.Mat(m[0:3], m[4:7], m[12], m[13], m[16:32]) for m in sc.materials],
                pass1=p1, hist=hist, after=st, radiance=(acc - before)[..., :3], visible=visible)


def test_pass1_matches_the_float64_restatement(frames):
    """VERDICT r03 item 6 — the last row pinned only by "GPU == oracle": RayGen_v6_pass1.hlsl:48-190 per pixel = SampleRIS (4 light + 1 BSDF candidates, balance-heuristic RIS
    weights, reservoir updates that consume random numbers: Sampler_v6.hlsl:653-736 with SampleLightNEE / SampleLightBSDF), GetP_Hat with its visibility ray and GetW, and
    SamplePathSimple (Path_Sampler_v6.hlsl:3-285: first BSDF bounce, then per bounce four unshadowed SampleLightNEE_GI + one SampleLightBSDF_GI, two strategy draws, the final
    reservoir shadow ray), restated in float64 from the HLSL text (tests/restir_ref64.py: pass1_pixel).  Every pixel of a frame, from the oracle's own primary hit and with the
    oracle answering the rays (the witness has no ray tracer).
      DI reservoir (SampleRIS + visibility + W): EVERY pixel holds the same sample and agrees in w_sum / W within 1e-5.
      GI reservoir + the pixel's estimate (SamplePathSimple: up to five dependent rays, 12 + 4 reservoir updates): the same selected sample (E3) and w_sum / W / sdata.debug within
      1e-5 for >= 96.5 % of the pixels, within 1e-3 for >= 99 %.  The remainder are ill-conditioned PATHS, not a different algorithm: traced one by one they bounce inside a room
      corner, between two coincident floor sheets, or leave a crease 2e-5 from where they arrived — a hit position that differs in the 7th digit (the witness samples its
      directions in float64) sends the next ray to another surface.  The frame's summed estimate agrees within 1e-3."""
    F = frames
    di_buf, gi_buf, sd_buf = F["pass1"]
    n = di_valid = gi_valid = same_e3 = 0
    worst_di, gi_err, sum_w, sum_o = 0.0, [], np.zeros(3), np.zeros(3)

    def same(a, b):          # which sample: its position; normals up to sign (SampleLightNEE turns a light's normal towards the shading point: for a point IN the light's plane
        return np.allclose(a["x2"], b["x2"], rtol=0.0, atol=2e-5) and (abs(float(np.dot(a["n2"], b["n2"]))) > 1.0 - 1e-6 or (not np.any(a["n2"]) and not np.any(b["n2"])))       # the sign of that ~0 cosine is rounding; every consumer re-orients n2)
    for y in range(F["H"]):
        for x in range(F["W"]):
            slot = X.map_pixel_id(F["W"], x, y)
            sd = X.load_sd(sd_buf, slot)
            r = X.pass1_pixel(x, y, F["W"], F["H"], F["seed"], sd, F["mats"], F["ke"], F["lights"], F["o2w"], F["closest"], F["occluded"], F["visible"])
            od, og = X.load_res(di_buf, slot), X.load_res(gi_buf, slot)
            if r is None:
                assert od["w_sum"] == 0.0 and og["w_sum"] == 0.0, (x, y)
                continue
            n += 1; di_valid += od["w_sum"] > 0; gi_valid += og["w_sum"] > 0
            assert same(r["di"], od) and np.allclose(r["di"]["L2"], od["L2"], rtol=2e-3, atol=1e-6), ("DI sample", x, y, r["di"], od)
            assert (r["di"]["M"], r["gi"]["M"]) == (od["M"], og["M"]) == (1, 1), (x, y)
            worst_di = max(worst_di, _rel(r["di"]["w_sum"], od["w_sum"]), _rel(r["di"]["W"], od["W"]))
            dbg = sd_buf[slot].view(np.float32)[12:15].astype(np.float64)
            e3 = bool(np.allclose(r["gi"]["L2"], og["L2"], rtol=2e-3, atol=1e-6)) and same(r["gi"], og)
            same_e3 += e3
            gi_err.append(max(_rel(r["gi"]["w_sum"], og["w_sum"]), _rel(r["gi"]["W"], og["W"]), float(np.abs(r["debug"] - dbg).max() / max(np.abs(dbg).max(), 1e-6))) if e3 else 1.0)
            sum_w += r["debug"]; sum_o += dbg
    gi_err = np.array(gi_err)
    assert n > 1200 and di_valid > 1000 and gi_valid > 900, (n, di_valid, gi_valid)
    assert worst_di < 1e-5, worst_di
    assert same_e3 >= 0.995 * n, (same_e3, n)
    assert (gi_err < 1e-5).mean() >= 0.965 and (gi_err < 1e-3).mean() >= 0.99, ((gi_err < 1e-5).mean(), (gi_err < 1e-3).mean())
    assert np.abs(sum_w - sum_o).max() / np.abs(sum_o).max() < 1e-3, (sum_w, sum_o)


def test_temporal_pass_matches_the_float64_restatement(frames):
    """RayGen_v6_pass2.hlsl:46-204: motion reprojection, acceptance predicates, the temporal pairwise-MIS weights (MIS_v6.hlsl:61-81, MIS_GI_v6.hlsl:78-110), both merges
    and the final W: every pixel's DI and GI reservoir after pass 2 — which sample it holds (exact), M (exact), w_sum and W (1e-5) — as re-computed in float64 from the HLSL"""
    F = frames
    n = nd = ng = moved = 0
    worst = 0.0
    for y in range(F["H"]):
        for x in range(F["W"]):
            r = X.temporal_pass_pixel(x, y, F["W"], F["H"], F["seed"], F["cam"], F["prev_view"], F["proj"], F["inst"], F["mats"], *F["pass1"], *F["hist"], F["visible"])
            if r is None:
                continue
            n += 1; nd += r["acc_di"]; ng += r["acc_gi"]; moved += r["pixel"] != (x, y)
            slot = X.map_pixel_id(F["W"], x, y)
            od, og = X.load_res(F["after"][0], slot), X.load_res(F["after"][1], slot)
            assert np.array_equal(np.float32(r["di"]["x2"]), np.float32(od["x2"])) and np.array_equal(np.float32(r["gi"]["x2"]), np.float32(og["x2"])), (x, y)
            assert (r["di"]["M"], r["gi"]["M"]) == (od["M"], og["M"]), (x, y)
            worst = max(worst, _rel(r["di"]["w_sum"], od["w_sum"]), _rel(r["gi"]["w_sum"], og["w_sum"]), _rel(r["di"]["W"], od["W"]), _rel(r["gi"]["W"], og["W"]))
    assert n > 1200 and nd > 900 and ng > 1100 and moved > 5            # the merges and the reprojection are really exercised
    assert worst < 1e-5, worst


def test_spatial_pass_matches_the_float64_restatement(frames):
    """RayGen_v6_pass3.hlsl:46-372: the neighbour search with its reject predicates (Common_v6.hlsl:246-350), the generalized pairwise MIS (MIS_v6.hlsl:2-59,
    MIS_GI_v6.hlsl:2-75), the reconnection Jacobian (Sampler_v6.hlsl:48-68), the merges and the final shade ReconnectDI * W + f_gi * W_gi: every pixel's history record and the
    radiance the frame added, as re-computed in float64 from the HLSL"""
    F = frames
    n = with_di = with_gi = picked = 0
    worst_w = worst_r = 0.0
    for y in range(F["H"]):
        for x in range(F["W"]):
            r = X.spatial_pass_pixel(x, y, F["W"], F["H"], F["seed"], F["cam"], F["mats"], F["after"][0], F["after"][1], F["after"][2], F["visible"])
            if r is None:
                continue
            n += 1; with_di += r["n_di"] > 0; with_gi += r["n_gi"] > 0; picked += any(r["picks_di"]) or any(r["picks_gi"])
            slot = X.map_pixel_id(F["W"], x, y)
            od, og = X.load_res(F["after"][3], slot), X.load_res(F["after"][4], slot)
            assert np.array_equal(np.float32(r["di"]["x2"]), np.float32(od["x2"])) and np.array_equal(np.float32(r["gi"]["x2"]), np.float32(og["x2"])), (x, y)
            assert (r["di"]["M"], r["gi"]["M"]) == (od["M"], og["M"]), (x, y)
            worst_w = max(worst_w, _rel(r["di"]["w_sum"], od["w_sum"]), _rel(r["gi"]["w_sum"], og["w_sum"]), _rel(r["di"]["W"], od["W"]), _rel(r["gi"]["W"], og["W"]))
            rad = F["radiance"][y, x]
            worst_r = max(worst_r, float(np.abs(r["radiance"] - rad).max() / max(np.abs(rad).max(), 1e-6)))
            assert 0.0 <= r["weights"]["mi_c"] <= 1.0 + 1e-9 and 0.0 <= r["weights"]["mi_c_gi"] <= 1.0
    assert n > 1200 and with_di > 1000 and with_gi > 900 and picked > 300
    assert worst_w < 1e-5 and worst_r < 1e-5, (worst_w, worst_r)


def test_pairwise_mis_weights_sum_to_one():
    """MIS_v6.hlsl:2-59 on a table of (M, p_hat) tuples: for ONE sample y, seen with target p_c(y) from the canonical pixel and p_j(y) from neighbour j, the canonical weight
    plus the non-canonical weights that the neighbours' own merges would give the same y is 1 — the property that makes the combined estimator unbiased; it fails for a
    transposed M ratio, a missing (M_sum - M_c) factor or weights normalised by the wrong count.  The temporal variant (one neighbour, p_hat left out by the reference:
    MIS_v6.hlsl:61-81) sums to one as well.  float64, restated from the HLSL."""
    rng = np.random.default_rng(7)
    for _ in range(2000):
        k = int(rng.integers(0, 4))
        Mc = float(rng.integers(1, 300)); Mn = rng.integers(1, 300, k).astype(float)
        cap = 128.0
        cM, nM = min(cap, Mc), np.minimum(cap, Mn)
        M_sum = cM + nM.sum()
        p_c, p_n = float(rng.uniform(0.0, 3.0)), rng.uniform(0.0, 3.0, k)
        if rng.random() < 0.2:
            p_n[rng.random(k) < 0.5] = 0.0                               # a neighbour that cannot see the sample
        if p_c == 0.0 and not p_n.any():
            continue
        m_c = cM / M_sum                                                 # GenPairwiseMIS_canonical
        for j in range(k):
            den = cM * p_c + (M_sum - cM) * p_n[j]
            if den > 0.0:
                m_c += (nM[j] / M_sum) * (cM * p_c / den)
        m_n = 0.0                                                        # GenPairwiseMIS_noncanonical, once per neighbour
        for j in range(k):
            num = (M_sum - cM) * p_n[j]; den = num + cM * p_c
            if den > 0.0:
                m_n += (nM[j] / M_sum) * (num / den)
        assert abs(m_c + m_n - 1.0) < 1e-12, (Mc, Mn, p_c, p_n)
        if k == 1:                                                       # temporal pair, caps of 16
            tc, tl = min(16.0, Mc), min(16.0, Mn[0]); ts = tc + tl
            mi_c = tc / ts + (tl / ts) * (tc / (tc + (ts - tc)))
            mi_t = (tl / ts) * (ts - tc) / ((ts - tc) + tc)
            assert abs(mi_c + mi_t - 1.0) < 1e-12
