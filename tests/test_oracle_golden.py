"""CPU suite, part 1: the oracle against (a) independent re-derivations written here in numpy / pure
Python, (b) the committed golden vectors, (c) physical properties of the math it restates."""
import os
import sys
import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_golden.npz"))


# ---- independent known answers ----------------------------------------------------------------
def tea_py(v0, v1):
    """TEA-4 written from Common_v6.hlsl:119-138 with Python integers (hand-checkable)."""
    M, s = 0xFFFFFFFF, 0
    for _ in range(4):
        s = (s + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C & M) ^ ((v1 + s) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D & M) ^ ((v0 + s) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    return v0, v1


def test_tea_matches_pure_python(orc):
    for seed in [(0, 0), (1, 2), (0xDEADBEEF, 0x12345678), (0xFFFFFFFF, 0xFFFFFFFF)]:
        vals, end = orc.tea(seed, 16)
        v0, v1 = seed
        for k in range(16):
            v0, v1 = tea_py(v0, v1)
            assert bits(vals[k]) == bits(np.float32(v0) / np.float32(4294967296.0))
        assert end == (v0, v1)


def test_tea_can_return_one(orc):
    # float(v0) rounds to nearest: v0 >= 0xFFFFFF80 gives exactly 1.0f
    assert np.float32(0xFFFFFF80) / np.float32(4294967296.0) == np.float32(1.0)


def test_seed_formula(orc):
    M = 0xFFFFFFFF
    for x, y, s, t in [(0, 0, 1, 0), (1919, 1079, 1, 12345), (7, 3, 2, 0xFFFFFFFF), (123, 456, 64, 9)]:
        e0 = ((y * 73856093) & M) ^ ((x * 19349663) & M) ^ ((s * 83492791) & M) ^ ((t * 293803) & M)
        e1 = ((x * 37623481) & M) ^ ((y * 51964263) & M) ^ ((s * 68250729) & M) ^ ((t * 423977) & M)
        assert orc.seed_init(x, y, s, t) == (e0, e1)


def test_half_round_matches_numpy_float16(orc):
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-70000, 70000, 2000), rng.uniform(-1, 1, 2000), 10.0 ** rng.uniform(-9, 5, 2000),
                         [0.0, 0.6, 0.73, 5.0, 17.0, 65504.0, 65519.0, 65520.0, 1e-8, 6e-8, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -25]]).astype(np.float32)
    with np.errstate(over="ignore"):
        expect = xs.astype(np.float16).astype(np.float32)
    got = np.array([orc.half_round(float(x)) for x in xs], np.float32)
    assert np.array_equal(bits(got), bits(expect))


def test_sincos_and_pow_accuracy(orc):
    xs = np.linspace(0.0, 6.2831855, 4001, dtype=np.float32)
    sc = np.array([orc.sincos(float(x)) for x in xs])
    assert np.abs(sc[:, 0] - np.sin(xs.astype(np.float64))).max() < 3e-7
    assert np.abs(sc[:, 1] - np.cos(xs.astype(np.float64))).max() < 3e-7
    xs = np.concatenate([np.linspace(0.0032, 1.0, 500), np.linspace(1.0, 40.0, 200)]).astype(np.float32)
    got = np.array([orc.pow_(float(x), 1.0 / 2.4) for x in xs])
    assert np.abs(got / np.power(xs.astype(np.float64), 1.0 / 2.4) - 1.0).max() < 2e-6


def test_rsqrt_det_is_within_one_ulp(orc):
    """the deterministic rsqrt behind normalize() (integer seed + three Newton steps, executed identically by the kernels) against
    float64: HLSL's rsqrt is a 1-ULP approximation, this one must be at least as good over the whole normal range"""
    rng = np.random.default_rng(5)
    xs = np.concatenate([np.exp(rng.uniform(np.log(1e-30), np.log(1e30), 200000)), rng.uniform(0.25, 4.0, 200000),
                         [1.0, 2.0, 4.0, 0.5, 3.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 2.0 - 2.0 ** -23, 1e-37, 3e38]]).astype(np.float32)
    got = orc.rsqrt(xs).astype(np.float64)
    ref = 1.0 / np.sqrt(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
    err = np.abs(got - ref) / ulp
    assert err.max() < 1.0, err.max()
    # unit vectors stay unit vectors: |normalize(v)| = 1 within 2 ulp
    v = rng.normal(size=(100000, 3)).astype(np.float32)
    n2 = (v.astype(np.float64) ** 2).sum(1)
    y = orc.rsqrt((v[:, 2] * v[:, 2] + (v[:, 1] * v[:, 1] + v[:, 0] * v[:, 0])).astype(np.float32)).astype(np.float64)
    assert np.abs(np.sqrt(n2) * y - 1.0).max() < 3e-7


def test_squared_length_threshold_is_exact():
    """csrc/rtx_shade.hpp tests `dot(n, n) >= 0x322bcc78` where the shader (and the oracle) say `length(n) > 0.0001f`: identical for
    every float, because correctly rounded sqrt is monotonic and 0x322bcc78 is the smallest float whose root rounds above 1e-4f"""
    c = np.float32(1e-4)
    T = np.array([0x322bcc78], np.uint32).view(np.float32)[0]
    u = np.arange(0x322bcc78 - 2_000_000, 0x322bcc78 + 2_000_000, dtype=np.uint32)
    xs = u.view(np.float32)
    assert np.array_equal(np.sqrt(xs) > c, xs >= T)
    ys = np.exp(np.random.default_rng(0).uniform(np.log(1e-14), np.log(1e-2), 1_000_000)).astype(np.float32)
    assert np.array_equal(np.sqrt(ys) > c, ys >= T)
    sp = np.array([0.0, -0.0, np.inf, np.nan, 1e-45, 3e38], np.float32)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(np.sqrt(sp) > c, sp >= T)


def test_mat4_inverse_vs_numpy(orc):
    rng = np.random.default_rng(1)
    for _ in range(50):
        m = rng.normal(size=(4, 4)).astype(np.float32)
        inv = orc.mat4_inverse(m.T.reshape(16)).reshape(4, 4).T      # column-major storage
        assert np.allclose(inv, np.linalg.inv(m.astype(np.float64)), rtol=2e-4, atol=2e-5)


def test_srgb8(orc):
    acc = np.zeros((1, 6, 4), np.float32)
    acc[0, :, 3] = 2.0
    acc[0, 0, :3] = 0.0; acc[0, 1, :3] = 2.0 * 0.0031308; acc[0, 2, :3] = 2.0 * 0.5; acc[0, 3, :3] = 2.0 * 7.0
    acc[0, 4, 0] = np.nan; acc[0, 5, 1] = np.inf
    out = orc.srgb8(acc)[0]
    lin = np.array([0.0, 0.0031308, 0.5, 1.0])
    enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    assert np.array_equal(out[:4, 0], np.floor(enc * 255 + 0.5).astype(np.uint8))
    assert tuple(out[4, :3]) == (255, 0, 255) and tuple(out[5, :3]) == (0, 255, 255)     # NaN -> magenta, Inf -> cyan
    assert (out[:, 3] == 255).all()


# ---- committed golden vectors ------------------------------------------------------------------
def test_golden_tea(orc, gold):
    for i in range(3):
        vals, end = orc.tea(tuple(int(v) for v in gold[f"tea{i}_seed"]), 8)
        assert np.array_equal(bits(vals), bits(gold[f"tea{i}_vals"])) and end == tuple(int(v) for v in gold[f"tea{i}_end"])


@pytest.fixture(scope="module")
def cornell_oracle(rt, orc, cornell):
    return orc.Oracle().load(cornell, 16 / 9)


@pytest.fixture(scope="module")
def garage_scene(rt, golden_dir):
    return rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")


@pytest.fixture(scope="module")
def garage_oracle(orc, garage_scene):
    return orc.Oracle().load(garage_scene, 16 / 9)


def test_golden_cornell_geometry(rt, cornell_oracle, gold):
    o = cornell_oracle
    assert np.array_equal(bits(o.lights()), bits(gold["cornell_lights"]))
    rays = o.primary_rays(rt.Params(width=1920, height=1080), 1)
    assert np.array_equal(bits(rays[[0, 960 + 540 * 1920, 1919 + 1079 * 1920]]), bits(gold["cornell_primary_pick"]))
    r = gold["cornell_rays"]
    for mode in (0, 1):                                # brute force and the oracle's BVH agree with the fixture
        assert np.array_equal(bits(o.trace_closest(r, mode)), bits(gold["cornell_hits"]))
        assert np.array_equal(o.trace_any(gold["cornell_shadow_rays"], mode), gold["cornell_shadow_occ"])
    assert np.array_equal(bits(o.surface(r, gold["cornell_hits"])), bits(gold["cornell_surface"]))


@pytest.mark.parametrize("tag,kw", [("c1", dict(spp=1, max_bounces=4)), ("c2", dict(spp=4, max_bounces=8))])
def test_golden_cornell_images(rt, orc, cornell, gold, tag, kw):
    o = orc.Oracle().load(cornell, 64 / 36)
    acc, cnt = o.render(rt.Params(width=64, height=36, nee_samples=1, flags=1, **kw))
    assert np.array_equal(bits(acc), bits(gold[f"cornell_{tag}_accum"]))
    assert tuple(int(v) for v in gold[f"cornell_{tag}_rays"]) == cnt


def test_golden_garage(rt, orc, garage_scene, garage_oracle, gold):
    og = garage_oracle
    assert len(og.lights()) == int(gold["garage_num_lights"][0])
    assert np.array_equal(bits(og.lights()[:8]), bits(gold["garage_lights_head"]))
    assert np.array_equal(bits(og.trace_closest(gold["garage_rays"], 0)), bits(gold["garage_hits"]))
    assert np.array_equal(bits(og.trace_closest(gold["garage_rays"], 1)), bits(gold["garage_hits"]))
    assert np.array_equal(bits(og.surface(gold["garage_rays"], gold["garage_hits"])), bits(gold["garage_surface"]))
    for flags in (1, 0):
        for mat in (1, 2, 5):
            assert np.array_equal(bits(og.bsdf_eval(mat, flags, gold["bsdf_in_eval"])), bits(gold[f"bsdf_eval_f{flags}_m{mat}"]))
            assert np.array_equal(bits(og.bsdf_sample(mat, flags, gold["bsdf_in_sample"])), bits(gold[f"bsdf_sample_f{flags}_m{mat}"]))
    o2 = orc.Oracle().load(garage_scene, 64 / 36)
    acc, cnt = o2.render(rt.Params(width=64, height=36, spp=2, max_bounces=6, nee_samples=2, flags=0))
    assert np.array_equal(bits(acc), bits(gold["garage_accum"])) and tuple(int(v) for v in gold["garage_rays_count"]) == cnt


# ---- properties --------------------------------------------------------------------------------
def test_bvh_equals_brute_force_random_rays(rt, cornell_oracle, garage_oracle):
    rng = np.random.default_rng(3)
    for o, lo, hi in ((cornell_oracle, -0.3, 1.3), (garage_oracle, -6, 6)):
        n = 20000
        r = np.zeros((n, 8), np.float32)
        r[:, 0:3] = rng.uniform(lo, hi, (n, 3)); d = rng.normal(size=(n, 3)); r[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
        r[:, 3], r[:, 7] = 1e-4, 1e4
        assert np.array_equal(bits(o.trace_closest(r, 0)), bits(o.trace_closest(r, 1)))
        r[:, 7] = 0.5 * (hi - lo)
        assert np.array_equal(o.trace_any(r, 0), o.trace_any(r, 1))


def test_axis_aligned_and_degenerate_rays(rt, cornell_oracle):
    """zero direction components, rays along edges / in wall planes: BVH must still equal brute force"""
    o = cornell_oracle
    rays = []
    for org in [(0.5, 0.5, -0.5), (0.5, 0.0, 0.5), (0.0, 0.5, 0.5), (0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)]:
        for d in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (-1, 1, 0)]:
            dn = np.array(d, np.float64) / np.linalg.norm(d)
            rays.append(list(org) + [1e-4] + list(dn) + [1e4])
    r = np.array(rays, np.float32)
    assert np.array_equal(bits(o.trace_closest(r, 0)), bits(o.trace_closest(r, 1)))
    assert np.array_equal(o.trace_any(r, 0), o.trace_any(r, 1))


def test_empty_scene_and_no_lights(rt, orc):
    class Empty:
        materials = np.zeros((1, 32), np.float32); meshes = []; instances = []
        def view_proj(self, aspect):
            return rt.lookat((0, 0, 1), (0, 0, 0), (0, 1, 0)), rt.perspective_fov_rh(1.0, aspect, 0.1, 100.0)
    o = orc.Oracle().load(Empty(), 1.0)
    acc, cnt = o.render(rt.Params(width=8, height=8, spp=2, max_bounces=3))
    assert cnt == (128, 0, 0) and not acc[..., :3].any() and (acc[..., 3] == 2).all()
    assert o.num_triangles == 0 and len(o.lights()) == 0


def test_lambert_sampling_is_cosine_weighted(rt, cornell_oracle):
    """E[1/pdf] over the sampler = hemisphere solid angle 2*pi*(3.1415/pi): pdf uses PI = 3.1415 (Common_v6.hlsl:1)"""
    o = cornell_oracle
    rng = np.random.default_rng(4)
    n = 40000
    nrm = np.tile(np.array([[0.0, 0.6, 0.8]], np.float32), (n, 1)); wo = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (n, 1))
    seeds = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    s = o.bsdf_sample(1, 1, np.concatenate([nrm, wo, seeds], 1))
    wi = s[:, :3]
    assert np.allclose(np.linalg.norm(wi, axis=1), 1.0, atol=1e-5) and ((wi * nrm).sum(1) >= 0).all()
    e = o.bsdf_eval(1, 1, np.concatenate([nrm, wo, wi], 1))
    f, pdf = e[:, :3], e[:, 3]
    cos = (wi * nrm).sum(1)
    assert np.allclose(pdf, np.maximum(cos, 1e-6) / np.float32(3.1415), rtol=1e-5)
    assert abs((1.0 / pdf).mean() / (2 * 3.1415) - 1.0) < 0.02                      # mean of 1/pdf = solid angle
    kd = np.float32(np.float16(0.73))                                                # fp16-rounded Kd (MaterialOptimized)
    assert np.allclose(f, kd / np.float32(3.1415), rtol=1e-6)
    assert np.allclose((f[:, 0] * cos / pdf), kd, rtol=1e-4)                          # albedo estimator is exact for Lambert


def test_ggx_mixture_is_finite_and_energy_bounded(rt, garage_oracle):
    og = garage_oracle
    rng = np.random.default_rng(5)
    n = 20000
    nrm = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (n, 1))
    wo = rng.normal(size=(n, 3)); wo[:, 1] = np.abs(wo[:, 1]) + 0.05; wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    seeds = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    for mat in (1, 2, 5):
        s = og.bsdf_sample(mat, 0, np.concatenate([nrm, wo, seeds], 1))
        wi = s[:, :3]
        assert np.isfinite(wi).all() and ((wi * nrm).sum(1) >= -1e-6).all()
        e = og.bsdf_eval(mat, 0, np.concatenate([nrm, wo, wi], 1))
        assert np.isfinite(e).all() and (e[:, :4] >= 0).all()
        assert np.allclose(e[:, 4] + e[:, 5], 1.0, atol=1e-6)                        # p_d + p_s = 1
        w = e[:, :3] * (wi * nrm).sum(1)[:, None] / np.maximum(e[:, 3:4], 1e-12)
        assert np.median(w) < 1.5                                                    # throughput weights are O(1)


class _FloorAndLight:
    """a diffuse floor (y = 0) under a rectangular light (y = 1), nothing else: the reference's data model built by hand"""
    KD, KE = (0.5, 0.25, 0.75), (4.0, 2.0, 1.0)                      # binary16-exact, so MaterialOptimized changes nothing
    LX, LZ, LY = 0.5, 0.25, 1.0                                       # light half extents and height

    def __init__(self):
        m = np.zeros((3, 32), np.float32)
        m[0, 0:4] = (1, 1, 1, 1); m[0, 12] = 1.0                      # default material (ObjLoader.h:415-417)
        m[1, 0:4] = (*self.KD, 1); m[1, 12] = 1.0                     # floor: Kd, Ks = 0, roughness 1
        m[2, 0:4] = (0, 0, 0, 1); m[2, 8:11] = self.KE; m[2, 12] = 1.0  # light: Ke
        self.materials = m
        F, lx, lz, ly = 6.0, self.LX, self.LZ, self.LY
        quads = [((-F, 0, -F), (-F, 0, F), (F, 0, F), (F, 0, -F), 1),          # floor, normal +y
                 ((-lx, ly, -lz), (lx, ly, -lz), (lx, ly, lz), (-lx, ly, lz), 2)]  # light
        v, idx, mid = [], [], []
        for a, b, c, d, mat in quads:
            base = len(v)
            v += [(*a, 0, 0, 0, 0), (*b, 0, 0, 0, 0), (*c, 0, 0, 0, 0), (*d, 0, 0, 0, 0)]     # normal (0,0,0) = flat, material base 0
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]; mid += [mat] * 6
        self.meshes = [(np.array(v, np.float32), np.array(idx, np.uint32), np.array(mid, np.uint32))]
        self.instances = [(0, np.eye(4, dtype=np.float32).reshape(16))]

    def view_proj(self, aspect, rt=None):
        return self._v, self._p


def test_direct_light_of_a_rectangle_matches_the_analytic_irradiance(rt, orc):
    """An ANALYTIC pin of the integrator, independent of this repository's arithmetic: a Lambert floor under a rectangular emitter
    with nothing else in the scene has outgoing radiance (Kd / pi) E(x), E the irradiance of the rectangle (numerical quadrature
    here).  With max_bounces = 2 the estimate is NEE at the floor (light CDF, area pdf, MIS weight) plus the BSDF-sampled ray that
    reaches the light (emissive-hit MIS weight); a wrong pdf convention, MIS weight, cosine or throughput shows up as a bias."""
    sc = _FloorAndLight()
    W, H = 24, 16
    sc._v = rt.lookat((2.5, 1.6, 1.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    sc._p = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    o = orc.Oracle().load(sc, W / H)
    p = rt.Params(width=W, height=H, spp=600, max_bounces=2, nee_samples=1, flags=1)
    acc, cnt = o.render(p)
    img = acc[..., :3] / np.maximum(acc[..., 3:4], 1.0)
    rays = o.primary_rays(rt.Params(width=W, height=H), 1)
    hits = o.trace_closest(rays, 0)
    prim = hits[:, 3].view(np.uint32).reshape(H, W)
    floor = (prim == 0) | (prim == 1)
    assert floor.sum() > 150 and ((prim == 2) | (prim == 3)).sum() > 3        # the view shows floor and light
    X = (rays[:, 0:3] + hits[:, 0:1] * rays[:, 4:7]).reshape(H, W, 3).astype(np.float64)
    # irradiance of the rectangle at floor points: E = Le * integral of cos_x cos_y / r^2 dA, midpoint rule 400 x 200
    nx, nz = 400, 200
    gx = (np.arange(nx) + 0.5) / nx * 2 * sc.LX - sc.LX
    gz = (np.arange(nz) + 0.5) / nz * 2 * sc.LZ - sc.LZ
    dA = (2 * sc.LX / nx) * (2 * sc.LZ / nz)
    ff = np.zeros((H, W))
    for yy, xx in zip(*np.nonzero(floor)):
        dx, dz = gx[:, None] - X[yy, xx, 0], gz[None, :] - X[yy, xx, 2]
        r2 = dx * dx + dz * dz + sc.LY ** 2
        ff[yy, xx] = (sc.LY * sc.LY / (r2 * r2)).sum() * dA                    # cos_x = cos_y = LY / r
    expect = ff[..., None] * (np.array(sc.KD) / np.pi)[None, None, :] * np.array(sc.KE)[None, None, :]
    got, ref = img[floor], expect[floor]
    # the whole floor region: 600 spp x ~250 pixels, Monte-Carlo error far below 1 %
    assert abs(got.sum() / ref.sum() - 1.0) < 0.01, got.sum() / ref.sum()
    for ch in range(3):
        assert abs(got[:, ch].sum() / ref[:, ch].sum() - 1.0) < 0.012
    # per pixel: no bias pattern (bright pixels under the light and dim ones far away alike)
    bright = ref[:, 0] > 0.25 * ref[:, 0].max()
    assert bright.sum() > 10 and np.abs(got[bright] / ref[bright] - 1.0).max() < 0.15
    dim = ~bright
    assert abs(got[dim].sum() / ref[dim].sum() - 1.0) < 0.02
    # pixels that see the emitter itself show its radiance exactly
    assert np.allclose(img[(prim == 2) | (prim == 3)], sc.KE)


@pytest.mark.parametrize("rough", [0.3, 0.7])
def test_ggx_floor_under_a_rectangle_light_matches_float64_quadrature(rt, orc, rough):
    """the GGX + NEE + MIS integrator pinned against an independent float64 quadrature (the Lambert-only rectangle test above cannot see a wrong
    microfacet pdf measure or MIS weight).  Floor material Kd 0.25, Ks 0.5, roughness 0.3 / 0.7, its Ess LUT from the host generator; the BSDF is the reference's mixture
    F = p_d f_lambert + p_s f_ggx (Sampler_v6.hlsl:443-457) restated in tests/ggx_ref64.py from the HLSL text.  Per floor pixel, with V = the camera direction:
      * FULL:      L_o = integral over the light of F(L, V) Le cos_x cos_y / r^2 dA       <-> max_bounces = 2: NEE at the floor + the BSDF-sampled ray that reaches the light.
                   Any pair of MIS weights that sums to one gives this; a pdf in the wrong measure (area vs solid angle) on either side breaks the sum and shows as a bias.
      * NEE HALF:  the same integrand times w = p_l / (p_l + P), p_l = r^2 / (A cos_y) the light's pdf per solid angle, P the mixture pdf (Path_Sampler_v6.hlsl:164)
                   <-> max_bounces = 1 (the path ends after its first vertex: only NEE contributes).  Pins the weight itself, restated in float64.
    Agreement within 3 sigma of the Monte-Carlo error (8 independent batches give sigma) plus 0.3 % for the midpoint rule."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ggx_ref64 as R
    sc = _FloorAndLight()
    kd, ks = (0.25, 0.25, 0.25), (0.5, 0.5, 0.5)
    lut = rt.generate_ess_lut(rough)
    sc.materials[1, 0:4] = (*kd, 1); sc.materials[1, 4:7] = ks; sc.materials[1, 12] = rough; sc.materials[1, 13] = 0.0; sc.materials[1, 16:32] = lut
    W, H = 24, 16
    sc._v = rt.lookat((2.5, 1.6, 1.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    sc._p = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    o = orc.Oracle().load(sc, W / H)
    rays = o.primary_rays(rt.Params(width=W, height=H), 1)
    hits = o.trace_closest(rays, 0)
    prim = hits[:, 3].view(np.uint32).reshape(H, W)
    floor = (prim == 0) | (prim == 1)
    assert floor.sum() > 150
    X = (rays[:, 0:3] + hits[:, 0:1] * rays[:, 4:7]).reshape(H, W, 3).astype(np.float64)
    V = -rays[:, 4:7].reshape(H, W, 3).astype(np.float64)
    m = R.Mat(kd, ks, rough, 0.0, lut)
    N = np.array([0.0, 1.0, 0.0])
    nx, nz = 240, 120
    gx = (np.arange(nx) + 0.5) / nx * 2 * sc.LX - sc.LX
    gz = (np.arange(nz) + 0.5) / nz * 2 * sc.LZ - sc.LZ
    dA = (2 * sc.LX / nx) * (2 * sc.LZ / nz); A = 4 * sc.LX * sc.LZ
    P_l = np.stack(np.broadcast_arrays(gx[:, None], np.full((1, 1), sc.LY), gz[None, :]), -1).reshape(-1, 3)       # light points
    full, half = np.zeros((H, W, 3)), np.zeros((H, W, 3))
    Ke = np.array(sc.KE, np.float64)
    for yy, xx in zip(*np.nonzero(floor)):
        d = P_l - X[yy, xx]; r2 = (d * d).sum(-1); L = d / np.sqrt(r2)[:, None]
        cos_x, cos_y = L[:, 1], L[:, 1]                                                                        # floor normal +y, light normal -y: both cosines are L.y
        F, Pm, _, _ = R.mixture(m, N, L, V[yy, xx])
        g = cos_x * cos_y / r2 * dA
        p_l = r2 / (A * cos_y)
        full[yy, xx] = (F * (g[:, None])).sum(0) * Ke
        half[yy, xx] = (F * (g * p_l / (p_l + Pm))[:, None]).sum(0) * Ke
    base = dict(width=W, height=H, spp=150, nee_samples=1, flags=0)
    for bounces, expect in ((2, full), (1, half)):
        sums = []
        for b in range(8):
            acc, _ = o.render(rt.Params(max_bounces=bounces, sample_base=1 + 150 * b, **base))
            sums.append((acc[..., :3] / np.maximum(acc[..., 3:4], 1.0))[floor].sum(0))
        sums = np.array(sums)
        mean, sem = sums.mean(0), sums.std(0, ddof=1) / np.sqrt(len(sums))
        ref = expect[floor].sum(0)
        z = np.abs(mean - ref) / (3.0 * sem + 0.003 * ref)
        assert (z < 1.0).all(), (rough, bounces, mean / ref, sem / ref)
        assert (sem / ref < 0.02).all()                                                                       # the test has the power to see a 3-5 % error
    assert (half[floor].sum(0) < 0.98 * full[floor].sum(0)).all()                                             # the BSDF-sampled half is not negligible: the weight is really exercised


def test_light_list_follows_reference_rules(rt, garage_oracle, garage_scene):
    L = garage_oracle.lights()
    n = len(L)
    assert n > 0 and (L.view(np.uint32)[:, 15] == n).all()                           # triCount on every record
    w = L[:, 11]
    assert (np.diff(w) <= 1e-12).all()                                               # sorted by weight, descending
    assert abs(w.sum() - 1.0) < 1e-4 and L[-1, 3] == 1.0                             # normalised; last cdf forced to 1
    assert np.allclose(np.cumsum(w)[:-1], L[:-1, 3], atol=1e-5)
    assert (L[:, 12:15] == 5.0).all() and (L[:, 16] == L[0, 16]).all()               # `lights` material Ke = 5; totalWeight everywhere


def test_progressive_accumulation_is_order_stable(rt, orc, cornell):
    o = orc.Oracle().load(cornell, 48 / 27)
    base = dict(width=48, height=27, max_bounces=5, nee_samples=1, flags=1)
    one, _ = o.render(rt.Params(spp=4, **base))
    two, _ = o.render(rt.Params(spp=1, sample_base=1, **base))
    two, _ = o.render(rt.Params(spp=3, sample_base=2, **base), two)
    assert np.array_equal(bits(one), bits(two))
    o.set_threads(1); a1, _ = o.render(rt.Params(spp=2, **base))
    o.set_threads(4); a4, _ = o.render(rt.Params(spp=2, **base))
    assert np.array_equal(bits(a1), bits(a4))                                        # thread count does not change bits


def test_shards_partition_the_image(rt, orc, cornell):
    o = orc.Oracle().load(cornell, 100 / 60)
    base = dict(width=100, height=60, spp=1, max_bounces=3, nee_samples=1, flags=1, tile_size=16)
    whole, cw = o.render(rt.Params(**base))
    parts = np.zeros_like(whole); tot = np.zeros(3, np.uint64)
    for r in range(3):
        a, c = o.render(rt.Params(shard_rank=r, shard_count=3, **base))
        assert not (parts[..., 3] * a[..., 3]).any()                                 # disjoint ownership
        parts += a; tot += np.array(c, np.uint64)
    assert np.array_equal(bits(parts), bits(whole)) and tuple(int(v) for v in tot) == cw


def test_v6_pass1_oracle_properties(rt, orc, cornell):
    """the literal pass-1 estimator: MapPixelID layout, reservoir invariants, and agreement in the mean with the
    unbiased bounce-loop estimator (pass 1 adds unshadowed NEE inside SamplePathSimple, so it is brighter)"""
    W, H = 64, 36
    o = orc.Oracle().load(cornell, W / H)
    p = rt.Params(width=W, height=H, spp=8, max_bounces=3, nee_samples=4, flags=1)
    acc, (di, gi, sd), cnt = o.render_v6_pass1(p)
    assert np.isfinite(acc).all() and (acc[..., 3] == 8).all()
    slots = ((W + 3) // 4) * ((H + 3) // 4) * 16
    assert len(di) == len(gi) == len(sd) == slots
    assert orc.lib.orc_map_pixel_id(W, 5, 6) == ((6 // 4) * 16 + 5 // 4) * 16 + (6 % 4) * 4 + 5 % 4      # Common_v6.hlsl:173-198
    M = di.view(np.uint16).reshape(slots, 20)[:, 19]
    sdf = sd.copy().view(np.uint8)
    mid = sd[:, 12:14].copy().view(np.uint16)[:, 0]
    hit_nonemissive = (mid != 0xFFFE) & (mid != 4)
    assert set(np.unique(M[hit_nonemissive])) <= {1} and (M[mid == 0xFFFE] == 0).all()             # reservoir.M = 1 after SampleRIS
    wsum = di[:, 12:16].copy().view(np.float32)[:, 0]
    assert (wsum[hit_nonemissive] >= 0).all() and wsum[hit_nonemissive].mean() > 0
    pt, _ = o.render(rt.Params(width=W, height=H, spp=8, max_bounces=5, nee_samples=1, flags=1))
    m1, m2 = acc[..., :3].mean() , pt[..., :3].mean()
    assert 0.9 < m1 / m2 < 1.5


def test_golden_reference_pipeline(rt, orc, cornell, gold):
    """pass-1 estimator and two ReSTIR frames against the committed vectors"""
    o = orc.Oracle().load(cornell, 48 / 28); o.set_camera(*cornell.view_proj(48 / 28))
    acc, (di, gi, sd), cnt = o.render_v6_pass1(rt.Params(width=48, height=28, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=3))
    assert np.array_equal(bits(acc), bits(gold["pass1_accum"])) and np.array_equal(di, gold["pass1_di"]) and np.array_equal(gi, gold["pass1_gi"]) and np.array_equal(sd, gold["pass1_sd"])
    assert cnt == tuple(int(v) for v in gold["pass1_rays"])
    acc, st, cnt = o.restir_frames(rt.Params(width=48, height=28, spp=2, max_bounces=3, nee_samples=4, flags=0, frame_seed=3))
    assert np.array_equal(bits(acc), bits(gold["restir_accum"])) and cnt == tuple(int(v) for v in gold["restir_rays"])
    assert np.array_equal(st[3], gold["restir_last_di"]) and np.array_equal(st[4], gold["restir_last_gi"]) and np.array_equal(st[5], gold["restir_last_sd"])


def test_restir_agrees_with_the_path_tracer_in_the_mean(rt, orc, cornell):
    """ReSTIR DI + GI (pass 1-3) and the bounce-loop estimator estimate the same image: means within a few percent"""
    W, H = 64, 36
    o = orc.Oracle().load(cornell, W / H); o.set_camera(*cornell.view_proj(W / H))
    acc, st, _ = o.restir_frames(rt.Params(width=W, height=H, spp=6, max_bounces=3, nee_samples=4, flags=1, frame_seed=1))
    pt, _ = o.render(rt.Params(width=W, height=H, spp=24, max_bounces=5, nee_samples=1, flags=1))
    a, b = acc[..., :3].sum((0, 1)) / 6, pt[..., :3].sum((0, 1)) / 24
    assert np.allclose(a, b, rtol=0.08), (a, b)
    M = st[3].view(np.uint16).reshape(len(st[3]), 20)[:, 19]
    assert M.max() > 16                                  # history accumulates beyond one frame's cap


def test_c1_cpu_reference_path_at_full_size(rt, orc, cornell):
    """BASELINE.json configs[0]: Cornell Box, 1920 x 1080, 1 spp, 4 bounces, Lambertian only, on the CPU reference path (the oracle; plumbing, no GPU).
    The whole frame in a few seconds; invariants that need no second implementation: one primary ray per pixel, every pixel gets one sample, the ray
    budget (<= 3 extension rays and <= 4 shadow rays per path), a black background outside the box opening and the light seen at its own radiance,
    determinism across thread counts, and the 128 x 72 crop of the same settings that the committed golden fixture holds."""
    W, H = 1920, 1080
    p = rt.Params(width=W, height=H, spp=1, max_bounces=4, nee_samples=1, rr_start=3, flags=rt.FLAG_LAMBERT_ONLY, frame_seed=1)
    o = orc.Oracle().load(cornell, W / H)
    o.set_threads(max(1, len(os.sched_getaffinity(0))))
    img, cnt = o.render(p)
    assert cnt[0] == W * H and 0 < cnt[1] <= 3 * W * H and 0 < cnt[2] <= 4 * W * H
    assert (img[..., 3] == 1.0).all() and np.isfinite(img).all() and (img[..., :3] >= 0.0).all()
    assert not img[:, :200, :3].any() and not img[:, -200:, :3].any()           # left and right of the box opening: the camera sees nothing (Miss -> black)
    light = img[:, :, :3].reshape(-1, 3).max(0)
    assert np.allclose(light, [17.0, 12.0, 4.0])                                  # the light seen directly carries Ke (fp16-exact values), Hit.hlsl:128-131
    o.set_threads(1)
    again, cnt1 = o.render(rt.Params(width=W // 8, height=H // 8, spp=1, max_bounces=4, nee_samples=1, rr_start=3, flags=1, frame_seed=1))
    o.set_threads(3)
    again3, cnt3 = o.render(rt.Params(width=W // 8, height=H // 8, spp=1, max_bounces=4, nee_samples=1, rr_start=3, flags=1, frame_seed=1))
    assert cnt1 == cnt3 and np.array_equal(again.view(np.uint32), again3.view(np.uint32))
