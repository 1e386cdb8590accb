"""Strategy 3 — rough dielectric transmission — is an EXTENSION: the reference names it and leaves a stub (BRDF_v6.hlsl:5,28-29,44-47,85-87,
102-104,120-122; Material.Ni is never filled, ObjLoader.h:428-435).  There is nothing to restate, so parity for it is unpinned by
definition; what pins it here is (1) a float64 restatement of Walter et al. 2007 (tests/ggx_ref64.py) for the evaluation, (2) properties:
Snell's law about the sampled microfacet, normalisation of the pdf, sample histograms, throughput weights in [0, 1], the strategy split
p_d alpha / p_d (1 - alpha), (3) an end-to-end pane scene, and — on the GPU — bit-equality with the oracle like everything else.
With RTX_FLAG_TRANSMISSION off, or for opaque materials, nothing changes (the golden fixtures of every other test are the witness)."""
import numpy as np
import pytest

import ggx_ref64 as R
from test_ggx_pins import _MaterialScene, _h_grid

FLAG_T = 4
NI = 1.5
GLASS = [(0.2, 0.0, 0.1), (0.5, 0.25, 0.3), (0.8, 0.5, 0.15), (0.1, 0.0, 0.04)]        # (roughness, dissolve alpha, Ks)


def _glass_scene(rt):
    m = np.zeros((len(GLASS) + 1, 32), np.float32)
    for k, (r, a, ks) in enumerate(GLASS):
        m[k, 0:4] = (0.6, 0.5, 0.4, a); m[k, 4:7] = ks; m[k, 7] = NI; m[k, 12] = r; m[k, 13] = 0.0
        m[k, 16:32] = rt.generate_ess_lut(float(np.float32(np.float16(r))))
    m[len(GLASS)] = m[1]; m[len(GLASS), 3] = 1.0                                         # the opaque twin of material 1
    return _MaterialScene(rt, m)


def _mat64(rt, k):
    r, a, ks = GLASS[k]
    m = R.Mat((0.6, 0.5, 0.4), (ks, ks, ks), r, 0.0, rt.generate_ess_lut(float(np.float32(np.float16(r)))))
    m.alpha = float(np.float32(np.float16(a)))
    return m


def _ulps(a, b):
    sp = np.spacing(np.abs(np.float32(b))).astype(np.float64)
    return np.abs(np.float64(a) - np.float64(b)) / np.maximum(sp, 1e-45)


def _check_eval(backend, rt):
    rng = np.random.default_rng(3)
    worst = 0.0
    for k in range(len(GLASS)):
        m = _mat64(rt, k)
        for trial in range(60):
            N = R.normalize(rng.normal(size=3))
            V = R.normalize(rng.normal(size=3)); L = R.normalize(rng.normal(size=3))
            side = float(R.dot(N, V))                                     # V on either side of the geometric normal: entering and leaving
            if abs(side) < 0.15:
                continue
            nf = N if side > 0 else -N
            if float(R.dot(nf, L)) > -0.1:
                L = L - 2.0 * float(R.dot(nf, L)) * nf                    # put L on the far side
                if float(R.dot(nf, L)) > -0.1:
                    continue
            eta_p = NI                                                     # thin-pane model: every crossing is air -> Ni
            q = np.concatenate([N, V, L]).astype(np.float32)
            N32, V32, L32 = (q[0:3].astype(np.float64), q[3:6].astype(np.float64), q[6:9].astype(np.float64))
            nf32 = N32 if side > 0 else -N32
            f, pdf = R.btdf_eval(m, nf32, L32, V32, eta_p)
            pd, ps = R.strategy_probs(m, V32, nf32)
            pt = pd * (1.0 - m.alpha)
            got = backend.bsdf_eval(k, FLAG_T, q[None])[0].astype(np.float64)
            assert abs(got[6] - eta_p) < 1e-6 and abs(got[5] - ps) < 1e-6 and abs(got[4] - pd * m.alpha) < 1e-6
            exp = np.array(list(pt * f) + [pt * pdf])
            if exp[3] == 0.0:
                assert (got[:4] == 0.0).all()
                continue
            # conditioning as in test_ggx_pins: 4 ulp + 32 x the effect of +-1 ulp input perturbations on the float64 value
            dev = np.zeros(4)
            for _ in range(16):
                pert = lambda a: (a.astype(np.float32) + np.spacing(np.abs(a.astype(np.float32))) * rng.integers(-1, 2, 3)).astype(np.float64)
                Vp, Lp, Np = pert(V32), pert(L32), pert(nf32)
                f2, pdf2 = R.btdf_eval(m, Np, Lp, Vp, eta_p)
                pd2, _ = R.strategy_probs(m, Vp, Np)
                dev = np.maximum(dev, np.abs(np.array(list(pd2 * (1.0 - m.alpha) * f2) + [pd2 * (1.0 - m.alpha) * pdf2]) - exp))
            ulp = np.spacing(np.abs(exp).astype(np.float32)).astype(np.float64)
            err = np.abs(got[:4] - exp)
            assert (err <= 6.0 * ulp + 32.0 * dev).all(), (k, trial, err / ulp, dev / ulp)
            worst = max(worst, float((err / (6.0 * ulp + 32.0 * dev)).max()))
        # the opaque twin and the flag-less call: no transmission anywhere
    q = np.array([[0, 0, 1, 0.3, 0.1, 0.9, 0.2, -0.1, -0.95]], np.float32)
    assert (backend.bsdf_eval(len(GLASS), FLAG_T, q)[0][:4] == backend.bsdf_eval(len(GLASS), 0, q)[0][:4]).all()
    assert np.array_equal(backend.bsdf_eval(1, 0, q)[0][:6], backend.bsdf_eval(len(GLASS), 0, q)[0][:6])          # without the flag dissolve plays no role
    return worst


def test_btdf_evaluation_equals_walter_2007_in_float64(rt, orc):
    o = orc.Oracle().load(_glass_scene(rt), 1.0)
    print("worst error / tolerance:", round(_check_eval(o, rt), 3))


def _samples(backend, k, N, V, n, rng):
    q = np.zeros((n, 8), np.float32); q[:, 0:3] = N; q[:, 3:6] = V
    q[:, 6:8] = rng.integers(0, 2 ** 32, size=(n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    s = backend.bsdf_sample(k, FLAG_T, q)
    return s[:, :3].astype(np.float64), s[:, 3].copy().view(np.uint32)


def test_strategy_split_snell_and_weights(rt, orc):
    """strategy 3 is drawn with probability (1 - p_s)(1 - alpha); its directions obey Snell's law about a microfacet normal of the
    visible distribution, lie on the far side, and carry a throughput f |cos| / pdf = (1 - F) G2 / G1 in [0, 1]"""
    o = orc.Oracle().load(_glass_scene(rt), 1.0)
    rng = np.random.default_rng(5)
    for k in range(len(GLASS)):
        m = _mat64(rt, k)
        for side in (+1.0, -1.0):                                          # hit from the front / from behind
            N = np.array([0.0, 0.0, 1.0]); c = 0.8
            V = np.array([np.sqrt(1 - c * c), 0.0, side * c])
            nf = side * N; eta_p = NI
            n = 60000
            wi, st = _samples(o, k, N, V, n, rng)
            pd, ps = R.strategy_probs(m, V, nf)
            p3 = float(pd * (1.0 - m.alpha))
            is3 = st == 3
            assert abs(is3.mean() - p3) < 4.0 * np.sqrt(p3 * (1 - p3) / n) + 1e-4, (k, side, is3.mean(), p3)
            assert abs((st == 1).mean() - float(ps)) < 4.0 * np.sqrt(float(ps) * (1 - float(ps)) / n) + 1e-4
            w3 = wi[is3]
            tir = (np.abs(w3).sum(1) == 0.0)
            assert not tir.any()                                           # every crossing enters the denser medium: no total internal reflection
            w3 = w3[~tir]
            assert np.allclose(np.linalg.norm(w3, axis=1), 1.0, atol=2e-6) and (w3 @ nf < 0.0).all()
            H = R.normalize(V + eta_p * w3); H = np.where((H @ nf)[:, None] < 0, -H, H)
            lhs = np.linalg.norm(V - (H @ V)[:, None] * H, axis=1); rhs = eta_p * np.linalg.norm(w3 - (w3 * H).sum(1)[:, None] * H, axis=1)
            assert np.allclose(lhs, rhs, atol=5e-6)                        # sin(theta_i) = eta_p sin(theta_t) about H
            e = o.bsdf_eval(k, FLAG_T, np.concatenate([np.tile(N, (len(w3), 1)), np.tile(V, (len(w3), 1)), w3], 1).astype(np.float32)).astype(np.float64)
            wgt = e[:, 0] * np.abs(w3 @ nf) / np.maximum(e[:, 3], 1e-300)
            assert (e[:, 3] > 0).all() and (wgt >= 0).all() and wgt.max() <= 1.0 + 1e-4, (k, side, wgt.max())
            assert wgt.mean() > 0.3                                        # and most of the energy goes through


def test_btdf_pdf_is_normalised_and_samples_follow_it(rt, orc):
    """over the half-vector hemisphere  int pdf(L(H)) / |dh/dwi| dH = 1 - P(total internal reflection)  (= 1 when entering); a chi-square of
    the sampled directions against the float64 pdf on the far hemisphere"""
    o = orc.Oracle().load(_glass_scene(rt), 1.0)
    Hh, w = _h_grid(900, 128)
    rng = np.random.default_rng(9)
    for k in (0, 1, 2):
        m = _mat64(rt, k)
        for side, c in ((+1.0, 0.9), (+1.0, 0.4), (-1.0, 0.9)):
            N = np.array([0.0, 0.0, 1.0]); V = np.array([np.sqrt(1 - c * c), 0.0, side * c]); nf = side * N
            eta_p = NI
            H = Hh * side                                                   # hemisphere around nf
            VH = H @ V
            eta = 1.0 / eta_p
            s2 = eta * eta * (1.0 - VH * VH)
            ok = (VH > 0) & (s2 < 1.0)
            L = (eta * VH[ok] - np.sqrt(1.0 - s2[ok]))[:, None] * H[ok] - eta * V
            q = np.concatenate([np.tile(N, (len(L), 1)), np.tile(V, (len(L), 1)), L], 1).astype(np.float32)
            e = o.bsdf_eval(k, FLAG_T, q).astype(np.float64)
            pd, ps = R.strategy_probs(m, V, nf); pt = float(pd * (1.0 - m.alpha))
            LH = (L * H[ok]).sum(1)
            jac = eta_p ** 2 * np.abs(LH) / (VH[ok] + eta_p * LH) ** 2
            total = float((e[:, 3] / pt / jac * w[ok]).sum())
            # D_V integrates to one over ALL visible normals; those that refract are the part counted here
            m1 = R.Mat((0.5, 0.5, 0.5), (1, 1, 1), GLASS[k][0], 1.0, np.ones(16))
            Lr = 2.0 * VH[VH > 0][:, None] * H[VH > 0] - V
            full = float((R.ggx_pdf(m1, nf, Lr, V) * 4.0 * VH[VH > 0] * w[VH > 0]).sum())
            part = float((R.ggx_pdf(m1, nf, 2.0 * VH[ok][:, None] * H[ok] - V, V) * 4.0 * VH[ok] * w[ok]).sum())
            assert abs(full - 1.0) < 3e-3 and abs(total - part) < 4e-3, (k, side, c, total, part, full)
            assert abs(total - 1.0) < 4e-3                                  # no total internal reflection: the refracted part is everything
            # histogram of the sampled transmitted directions (cos theta x phi bins on the far hemisphere)
            n = 120000
            wi, st = _samples(o, k, N, V, n, rng)
            w3 = wi[(st == 3) & (np.abs(wi).sum(1) > 0)]
            nb_c, nb_p, sub = 6, 12, 20
            cz = (np.arange(nb_c * sub) + 0.5) / (nb_c * sub); ph = (np.arange(nb_p * sub) + 0.5) / (nb_p * sub) * 2 * np.pi - np.pi
            CZ, PH = np.meshgrid(cz, ph, indexing="ij"); sn = np.sqrt(1 - CZ * CZ)
            Ld = np.stack([sn * np.cos(PH), sn * np.sin(PH), -side * CZ], -1).reshape(-1, 3)
            dens = np.array([R.btdf_eval(m, nf, l, V, eta_p)[1] for l in Ld])
            prob = (dens * (2 * np.pi / (nb_p * sub)) * (1.0 / (nb_c * sub))).reshape(nb_c, sub, nb_p, sub).sum((1, 3)).reshape(-1)
            ci = np.minimum((np.abs(w3[:, 2]) * nb_c).astype(int), nb_c - 1)
            pi_ = np.minimum(((np.arctan2(w3[:, 1], w3[:, 0]) + np.pi) / (2 * np.pi) * nb_p).astype(int), nb_p - 1)
            obs = np.bincount(ci * nb_p + pi_, minlength=nb_c * nb_p).astype(np.float64)
            exp = prob / prob.sum() * len(w3)
            big = exp >= 10
            chi2 = float((((obs - exp)[big] ** 2) / exp[big]).sum()); dof = int(big.sum()) - 1
            assert chi2 < dof + 5.0 * np.sqrt(2.0 * dof), (k, side, c, chi2, dof)


class _PaneScene:
    """a lit diffuse wall seen through a glass pane (or through nothing): floor-less, light behind the camera"""
    def __init__(self, rt, with_pane, alpha):
        m = np.zeros((4, 32), np.float32)
        m[0, 0:4] = (1, 1, 1, 1); m[0, 12] = 1
        m[1, 0:4] = (0.7, 0.7, 0.7, 1); m[1, 12] = 1                        # wall
        m[2, 0:4] = (0, 0, 0, 1); m[2, 8:11] = (6, 6, 6); m[2, 12] = 1      # light
        m[3, 0:4] = (0.1, 0.1, 0.1, alpha); m[3, 4:7] = 0.04; m[3, 7] = NI; m[3, 12] = 0.05; m[3, 16:32] = rt.generate_ess_lut(float(np.float32(np.float16(0.05))))
        self.materials = m
        quads = [((-3, -3, -2), (3, -3, -2), (3, 3, -2), (-3, 3, -2), 1),   # wall at z = -2, normal +z
                 ((-1, 1.9, 1.0), (-1, 1.9, 3.0), (1, 1.9, 3.0), (1, 1.9, 1.0), 2)]   # a light panel above and behind the pane, facing down
        if with_pane:
            quads.append(((-3, -3, 0), (3, -3, 0), (3, 3, 0), (-3, 3, 0), 3))          # the pane at z = 0, normal +z (towards the camera)
        v, idx, mid = [], [], []
        for a, b, c, d, mat in quads:
            base = len(v); v += [(*a, 0, 0, 0, 0), (*b, 0, 0, 0, 0), (*c, 0, 0, 0, 0), (*d, 0, 0, 0, 0)]
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]; mid += [mat] * 6
        self.meshes = [(np.array(v, np.float32), np.array(idx, np.uint32), np.array(mid, np.uint32))]
        self.instances = [(0, np.eye(4, dtype=np.float32).reshape(16))]
        self._vp = (rt.lookat((0, 0, 4), (0, 0, 0), (0, 1, 0)), rt.perspective_fov_rh(np.radians(40.0), 1.0, 0.1, 100.0))

    def view_proj(self, aspect):
        return self._vp


def test_a_clear_pane_barely_dims_the_wall_behind_it(rt, orc):
    """end to end on the oracle: a wall lit from the camera side, seen directly and through a nearly clear, nearly smooth pane (alpha 0, F0 0.04):
    with transmission the wall keeps most of its brightness (two passages through the interface lose ~2 x 4-5 % to Fresnel reflection, plus
    the samples the rough interface scatters away); without the flag the same pane is an opaque dark plate"""
    W = H = 48
    p = lambda flags: rt.Params(width=W, height=H, spp=192, max_bounces=6, nee_samples=1, flags=flags, frame_seed=2)
    mean = lambda a: float((a[H // 4:3 * H // 4, W // 4:3 * W // 4, :3] / np.maximum(a[H // 4:3 * H // 4, W // 4:3 * W // 4, 3:4], 1)).mean())
    bare = mean(orc.Oracle().load(_PaneScene(rt, False, 0.0), 1.0).render(p(FLAG_T))[0])
    o = orc.Oracle().load(_PaneScene(rt, True, 0.0), 1.0)
    through = mean(o.render(p(FLAG_T))[0]); blocked = mean(o.render(p(0))[0])
    print("wall radiance: bare %.4f, through the pane %.4f, flag off %.4f" % (bare, through, blocked))
    assert bare > 0.05
    assert 0.72 * bare < through < 0.95 * bare                           # measured 0.81
    assert blocked < 0.6 * bare                                           # (what is seen then is the lit dark plate itself, Kd 0.1)


@pytest.mark.gpu
def test_gpu_transmission_equals_the_oracle(rt, orc):
    """the device's strategy 3 — evaluation, sampling, and whole frames through the fused tiny-scene kernels and the general BVH path — bit for bit"""
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    sc = _glass_scene(rt)
    c = rt.Context(0); c.upload(sc, 1.0)
    o = orc.Oracle().load(sc, 1.0)
    _check_eval(c, rt)
    rng = np.random.default_rng(21)
    n = 20000
    for k in range(len(GLASS) + 1):
        q = np.zeros((n, 9), np.float32)
        q[:, 0:3] = R.normalize(rng.normal(size=(n, 3))); q[:, 3:6] = R.normalize(rng.normal(size=(n, 3))); q[:, 6:9] = R.normalize(rng.normal(size=(n, 3)))
        for flags in (FLAG_T, 0):
            assert np.array_equal(bits(c.bsdf_eval(k, flags, q)), bits(o.bsdf_eval(k, flags, q))), (k, flags)
            s8 = q[:, :8].copy(); s8[:, 6:8] = rng.integers(0, 2 ** 32, size=(n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
            assert np.array_equal(bits(c.bsdf_sample(k, flags, s8)), bits(o.bsdf_sample(k, flags, s8))), (k, flags)
    c.close()
    W = H = 64
    for alpha in (0.0, 0.4):
        ps = _PaneScene(rt, True, alpha)
        oo = orc.Oracle().load(ps, 1.0)
        for flags in (FLAG_T, 0):
            p = rt.Params(width=W, height=H, spp=8, max_bounces=6, nee_samples=1, flags=flags, frame_seed=4)
            ref, cnt = oo.render(p)
            for small in (1, 0):
                g = rt.Context(0); g.set_option(rt.OPT_SMALL_SCENE, small); g.upload(ps, 1.0); g.clear(W, H); g.render(p); st = g.stats()
                assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt, (alpha, flags, small)
                assert np.array_equal(bits(g.read_accum()), bits(ref)), (alpha, flags, small)
                g.close()
