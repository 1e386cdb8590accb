"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Integer / index work is compared bit-exactly; radiance within north_star's 1e-4 relative L2 (in
practice bit-exact, which the tests also report)."""
import os
import numpy as np
import pytest

FUZZ_SEED = int(os.environ.get("RTX_FUZZ_SEED", "0"))       # offset of every random test's seeds: RTX_FUZZ_SEED=100000 RTX_FUZZ_SCENES=5000 explores new scenes

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-4   # BASELINE.json north_star: "within 1e-4 relative per-pixel L2"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rel_l2(a, b):
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


@pytest.fixture(scope="module")
def ctx(rt):
    c = rt.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=["fused_small", "separate_small", "separate_bvh", "fused_bvh", "steal_bvh", "inplace_bvh"])
def cornell_pair(request, rt, orc, cornell):
    """Cornell through every kernel path: fused bounce kernel (default for tiny scenes), separate
    trace / shade / shadow kernels with the tiny-scene traversal, and the general BVH traversal with separate kernels
    (the default for general scenes), with the fused k_bounce_bvh, and with work stealing between the sub-queues"""
    c = rt.Context(0)
    c.set_option(rt.OPT_SMALL_SCENE, 0 if request.param.endswith("_bvh") else 1)
    c.set_option(rt.OPT_WORK_STEALING, 1 if request.param == "steal_bvh" else 0)
    c.set_option(rt.OPT_COMPACT_STATE, 0 if request.param == "inplace_bvh" else 1)      # path state by path id, in place (the default keeps it by queue position)
    c.set_option(rt.OPT_FUSED_BVH, 1 if request.param == "fused_bvh" else 0)
    c.set_option(rt.OPT_FUSED_BOUNCE, 1 if request.param == "fused_small" else 0)
    c.upload(cornell, 16 / 9)
    yield c, orc.Oracle().load(cornell, 16 / 9)
    c.close()


def random_rays(n, seed, lo=-0.2, hi=1.2, tmax=1e4):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] = o, 1e-4, d.astype(np.float32), tmax
    return r


def test_tea_rng_bit_exact(rt, orc, ctx):
    for seed in [(0, 0), (1234567, 89), (0xFFFFFFFF, 0x80000000)]:
        g, gs = ctx.tea(seed, 64)
        c, cs = orc.tea(seed, 64)
        assert np.array_equal(bits(g), bits(c)) and gs == cs


def test_primary_rays_bit_exact(rt, cornell_pair):
    ctx, o = cornell_pair
    for flags in (0, rt.FLAG_JITTER):
        p = rt.Params(width=96, height=54, flags=flags, frame_seed=7)
        assert np.array_equal(bits(ctx.primary_rays(p, 3)), bits(o.primary_rays(p, 3)))


@pytest.fixture(params=["small_scene_path", "bvh_path"])
def cornell_variant(request, rt, orc, cornell):
    """Cornell through both traversal paths: the tiny-scene brute-force pre-test and the general BVH"""
    c = rt.Context(0)
    c.set_option(rt.OPT_SMALL_SCENE, 1 if request.param == "small_scene_path" else 0)
    c.upload(cornell, 16 / 9)
    yield c, orc.Oracle().load(cornell, 16 / 9)
    c.close()


def test_trace_closest_equals_brute_force(rt, cornell_variant):
    ctx, o = cornell_variant
    p = rt.Params(width=160, height=90)
    rng = np.random.default_rng(17)
    # rays that start ON surfaces (like every secondary ray) and graze / run inside wall planes
    on = o.primary_rays(p); h0 = o.trace_closest(on, 1); hit0 = bits(h0)[:, 3] != 0xFFFFFFFF
    sec = random_rays(int(hit0.sum()), 4)
    sec[:, 0:3] = on[hit0, 0:3] + h0[hit0, 0:1] * on[hit0, 4:7]; sec[:, 3] = 2e-5
    graz = random_rays(4000, 5); graz[:, 5] = rng.uniform(-1e-4, 1e-4, 4000).astype(np.float32); graz[:, 1] = rng.choice([0.0, 0.3, 0.98883], 4000)
    graz[:, 4:7] /= np.linalg.norm(graz[:, 4:7], axis=1, keepdims=True)
    rays = np.concatenate([on, random_rays(200000, 1), sec, graz])
    g = ctx.trace_closest(rays)
    c = o.trace_closest(rays, mode=0)       # brute force over all triangles
    assert np.array_equal(bits(g)[:, 3], bits(c)[:, 3]), "hit triangle ids differ"
    hit = bits(c)[:, 3] != 0xFFFFFFFF
    assert np.array_equal(bits(g)[hit], bits(c)[hit]), "t/u/v differ"
    assert hit.mean() > 0.3


def test_trace_any_equals_brute_force(rt, cornell_variant):
    ctx, o = cornell_variant
    rays = np.concatenate([random_rays(200000, 2, lo=0.05, hi=0.95, tmax=0.6), random_rays(50000, 6, lo=-0.1, hi=1.1, tmax=0.05)])
    g = ctx.trace_any(rays)
    c = o.trace_any(rays, mode=0)
    assert np.array_equal(g, c)
    assert 0.05 < c.mean() < 0.95


def test_surface_reconstruction_bit_exact(rt, cornell_pair):
    ctx, o = cornell_pair
    rays = random_rays(20000, 3)
    hits = o.trace_closest(rays, mode=1)
    assert np.array_equal(bits(ctx.surface(rays, hits)), bits(o.surface(rays, hits)))


@pytest.mark.parametrize("flags", [1, 0])
def test_bsdf_eval_and_sample_bit_exact(rt, cornell_pair, flags):
    ctx, o = cornell_pair
    rng = np.random.default_rng(5)
    n = 4096
    def unit(k):
        v = rng.normal(size=(k, 3)); return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    nrm, wo, wi = unit(n), unit(n), unit(n)
    flip = (nrm * wo).sum(1) < 0
    wo[flip] = -wo[flip]
    for mat in (1, 2, 0):
        q = np.concatenate([nrm, wo, wi], axis=1)
        ge, ce = ctx.bsdf_eval(mat, flags, q), o.bsdf_eval(mat, flags, q)
        assert np.array_equal(bits(ge), bits(ce)), f"eval differs for material {mat}"
        seeds = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
        q2 = np.concatenate([nrm, wo, seeds], axis=1)
        gs, cs = ctx.bsdf_sample(mat, flags, q2), o.bsdf_sample(mat, flags, q2)
        assert np.array_equal(bits(gs), bits(cs)), f"sample differs for material {mat}"
        # the form k_shade runs since round 4 (MixView: the view-dependent terms computed once per shading point and shared by the NEE samples, the strategy draw and the
        # continuation): bit 31 of the debug flags selects it; same bits
        assert np.array_equal(bits(ctx.bsdf_eval(mat, flags | 0x80000000, q)), bits(ge)), f"MixView eval differs for material {mat}"
        assert np.array_equal(bits(ctx.bsdf_sample(mat, flags | 0x80000000, q2)), bits(gs)), f"MixView sample differs for material {mat}"


@pytest.mark.parametrize("cfg", [
    dict(width=128, height=72, spp=1, max_bounces=4, nee_samples=1, flags=1),     # C1 settings, small
    dict(width=128, height=72, spp=4, max_bounces=8, nee_samples=1, flags=1),     # C2 settings, small
    dict(width=96, height=54, spp=2, max_bounces=8, nee_samples=4, flags=1),      # v6 nee_samples = 4
    dict(width=96, height=54, spp=2, max_bounces=8, nee_samples=0, flags=1),      # BSDF sampling only
    dict(width=96, height=54, spp=2, max_bounces=6, nee_samples=1, flags=0),      # full strategy selection (GGX lobe)
    dict(width=100, height=50, spp=3, max_bounces=8, nee_samples=1, flags=3, sample_base=5, frame_seed=99),  # jitter, ragged size
])
def test_render_parity_cornell(rt, cornell_pair, cfg):
    ctx, o = cornell_pair
    p = rt.Params(**cfg)
    aspect = p.width / p.height
    ctx.set_camera(*rt.Scene.cornell().view_proj(aspect)); o.set_camera(*rt.Scene.cornell().view_proj(aspect))
    ctx.clear(p.width, p.height)
    ctx.render(p)
    g = ctx.read_accum()
    c, cnt = o.render(p)
    st = ctx.stats()
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt
    assert np.array_equal(g[..., 3], c[..., 3])
    r = rel_l2(g[..., :3], c[..., :3])
    nbad = int((bits(g) != bits(c)).any(axis=-1).sum())
    print(f"rel_l2={r:.3e} pixels_not_bit_exact={nbad}/{p.width * p.height}")
    assert r <= REL_L2_TOL
    # per-pixel: relative error of every pixel with signal
    num = np.sqrt(((g[..., :3] - c[..., :3]) ** 2).sum(-1)); den = np.sqrt((c[..., :3] ** 2).sum(-1))
    assert (num <= REL_L2_TOL * np.maximum(den, 1e-3)).all()
    assert np.array_equal(ctx.read_srgb8(), orc_srgb(o, c))


class XformedScene:
    """a Scene with its instance transforms replaced (same duck type as rt.Scene for Context.upload / Oracle.load)"""
    def __init__(self, base, mats):
        self.materials, self.meshes, self._base = base.materials, base.meshes, base
        self.instances = [(mesh, np.asarray(m, np.float32).reshape(16)) for (mesh, _), m in zip(base.instances, mats)]

    def view_proj(self, aspect):
        return self._base.view_proj(aspect)


@pytest.mark.parametrize("variant", ["fused_small", "separate_small", "separate_bvh", "fused_bvh"])
@pytest.mark.parametrize("kind", ["rotated_sheared", "mirrored"])
def test_tiny_scene_paths_on_a_skewed_room(rt, orc, cornell, variant, kind):
    """the tiny-scene machinery (planar-quad merging, conservative pre-test, convex-hull faces skipped by NEE segments) on a room
    that is NOT axis-aligned: the Cornell instance under a rotation + shear + non-uniform scale, and under a mirroring transform
    (flipped winding).  Closest / any-hit records against the oracle's brute force, images bit for bit."""
    th, ph = 0.37, -0.21
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]) @ np.array([[1, 0, 0], [0, np.cos(ph), -np.sin(ph)], [0, np.sin(ph), np.cos(ph)]])
    S = np.array([[1.1, 0.15, 0.0], [0.0, 0.9, 0.1], [0.05, 0.0, 1.2]])
    A = R @ S if kind == "rotated_sheared" else np.diag([-1.0, 1.0, 1.0])
    M = np.eye(4); M[:3, :3] = A; M[:3, 3] = (0.5, 0.5, 0.5) - A @ np.array([0.5, 0.5, 0.5]) + (0.02, -0.01, 0.03)   # about the room's centre
    sc = XformedScene(cornell, [M.T.reshape(16)])                      # column-major storage of a column-vector matrix
    c = rt.Context(0)
    c.set_option(rt.OPT_SMALL_SCENE, 0 if variant.endswith("_bvh") else 1); c.set_option(rt.OPT_FUSED_BVH, 1 if variant == "fused_bvh" else 0); c.set_option(rt.OPT_FUSED_BOUNCE, 1 if variant == "fused_small" else 0)
    c.upload(sc, 16 / 9)
    o = orc.Oracle().load(sc, 16 / 9)
    p = rt.Params(width=112, height=63, spp=3, max_bounces=8, nee_samples=2, flags=1)
    on = o.primary_rays(p); h0 = o.trace_closest(on, 0); hit0 = bits(h0)[:, 3] != 0xFFFFFFFF
    sec = random_rays(int(hit0.sum()), 4); sec[:, 0:3] = on[hit0, 0:3] + h0[hit0, 0:1] * on[hit0, 4:7]; sec[:, 3] = 2e-5
    rays = np.concatenate([on, random_rays(60000, 1, -0.3, 1.3), sec])
    g, b = c.trace_closest(rays), o.trace_closest(rays, mode=0)
    assert np.array_equal(bits(g)[:, 3], bits(b)[:, 3])
    hit = bits(b)[:, 3] != 0xFFFFFFFF
    assert np.array_equal(bits(g)[hit], bits(b)[hit]) and hit.mean() > 0.3
    sh = random_rays(60000, 2, 0.0, 1.0, tmax=0.7)
    assert np.array_equal(c.trace_any(sh), o.trace_any(sh, mode=0))
    c.clear(p.width, p.height); c.render(p)
    ga = c.read_accum(); ca, cnt = o.render(p)
    st = c.stats()
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt and st.rays_shadow > 0
    assert np.array_equal(bits(ga), bits(ca)), f"{int((bits(ga) != bits(ca)).any(-1).sum())} pixels differ"
    assert ga[..., :3].mean() > 0.01
    c.close()


def orc_srgb(o, acc):
    import __graft_entry__ as graft
    return graft.load_oracle().srgb8(acc)


def test_progressive_accumulation_and_batches(rt, cornell_pair):
    """spp split over calls / batches gives the same sums as one call (fixed per-pixel order)."""
    ctx, o = cornell_pair
    base = dict(width=64, height=36, max_bounces=5, nee_samples=1, flags=1)
    ctx.set_camera(*rt.Scene.cornell().view_proj(64 / 36)); o.set_camera(*rt.Scene.cornell().view_proj(64 / 36))
    ctx.clear(64, 36); ctx.render(rt.Params(spp=6, **base)); one = ctx.read_accum()
    ctx.clear(64, 36)
    ctx.render(rt.Params(spp=2, sample_base=1, **base)); ctx.render(rt.Params(spp=4, sample_base=3, **base))
    two = ctx.read_accum()
    assert np.array_equal(bits(one), bits(two))
    ctx.set_option(rt.OPT_PATHS_PER_BATCH, 4096)     # force many small batches
    ctx.clear(64, 36); ctx.render(rt.Params(spp=6, **base)); three = ctx.read_accum()
    ctx.set_option(rt.OPT_PATHS_PER_BATCH, 128 << 20)
    assert np.array_equal(bits(one), bits(three))
    c, _ = o.render(rt.Params(spp=6, **base))
    assert rel_l2(one[..., :3], c[..., :3]) <= REL_L2_TOL


def test_shards_reassemble_bit_identically(rt, cornell_pair):
    """pixel-tile sharding: any shard count gives the same image (seeds depend on x, y, s only)."""
    ctx, o = cornell_pair
    base = dict(width=200, height=120, spp=2, max_bounces=5, nee_samples=1, flags=1, tile_size=32)
    ctx.set_camera(*rt.Scene.cornell().view_proj(200 / 120)); o.set_camera(*rt.Scene.cornell().view_proj(200 / 120))
    ctx.clear(200, 120); ctx.render(rt.Params(**base)); whole = ctx.read_accum()
    ctx.clear(200, 120)
    for r in range(3):
        ctx.render(rt.Params(shard_rank=r, shard_count=3, **base))
    assert np.array_equal(bits(ctx.read_accum()), bits(whole))
    c, _ = o.render(rt.Params(shard_rank=1, shard_count=3, **base))
    ctx.clear(200, 120); ctx.render(rt.Params(shard_rank=1, shard_count=3, **base))
    assert rel_l2(ctx.read_accum()[..., :3], c[..., :3]) <= REL_L2_TOL


def test_pack_unpack_kernels_match_host_layout(rt, cornell_pair):
    """rtx_pack_tiles / rtx_unpack_tiles against the numpy slab layout the gloo tests use"""
    import torch
    from royaltracer_dx_amd import sharding
    ctx, o = cornell_pair
    W, H, TS, world = 200, 120, 32, 3
    base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=1, flags=1, tile_size=TS)
    ctx.set_camera(*rt.Scene.cornell().view_proj(W / H))
    ctx.clear(W, H); ctx.render(rt.Params(**base)); whole = ctx.read_accum()
    slabs = []
    for r in range(world):
        p = rt.Params(shard_rank=r, shard_count=world, **base)
        n = ctx.slab_bytes(p) // 4
        t = torch.empty(n, dtype=torch.float32, device="cuda")
        ctx.pack_tiles(p, t.data_ptr()); torch.cuda.synchronize()
        got = t.cpu().numpy().reshape(-1, 4)
        assert np.array_equal(bits(got), bits(sharding.pack(whole, TS, r, world)))
        slabs.append(t)
    allslabs = torch.cat(slabs); torch.cuda.synchronize()               # (torch's stream vs the context's own stream)
    ctx.clear(W, H)
    ctx.unpack_tiles(rt.Params(shard_rank=0, shard_count=world, **base), allslabs.data_ptr())
    assert np.array_equal(bits(ctx.read_accum()), bits(whole))


def test_garage_scene_parity(rt, orc, golden_dir):
    """the reference's own startup scene (garage.obj + monke.obj, two instances, GGX materials, smooth normals)"""
    import os
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    c = rt.Context(0); c.upload(sc, 96 / 54)
    o = orc.Oracle().load(sc, 96 / 54)
    rays = np.concatenate([o.primary_rays(rt.Params(width=96, height=54)), random_rays(30000, 8, -6, 6)])
    g, b = c.trace_closest(rays), o.trace_closest(rays, 0)
    assert np.array_equal(bits(g)[:, 3], bits(b)[:, 3])
    hit = bits(b)[:, 3] != 0xFFFFFFFF
    assert np.array_equal(bits(g)[hit], bits(b)[hit])
    assert np.array_equal(bits(c.surface(rays, b)), bits(o.surface(rays, b)))
    assert np.array_equal(bits(c.lights()), bits(o.lights()))
    for cfg in (dict(spp=2, max_bounces=6, nee_samples=2, flags=0), dict(spp=1, max_bounces=8, nee_samples=1, flags=2)):
        p = rt.Params(width=96, height=54, **cfg)
        c.clear(96, 54); c.render(p)
        ga = c.read_accum(); ca, cnt = o.render(p)
        st = c.stats()
        assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt
        r = rel_l2(ga[..., :3], ca[..., :3])
        print("garage rel_l2", r, "not bit exact:", int((bits(ga) != bits(ca)).any(-1).sum()))
        assert r <= REL_L2_TOL
    c.close()


def test_golden_fixtures_on_gpu(rt, golden_dir, cornell_pair):
    """the committed golden vectors (tests/golden/oracle_golden.npz) reproduced by the HIP path alone"""
    import os
    ctx, _ = cornell_pair
    g = np.load(os.path.join(golden_dir, "oracle_golden.npz"))
    ctx.set_camera(*rt.Scene.cornell().view_proj(16 / 9))
    hits = ctx.trace_closest(g["cornell_rays"])
    assert np.array_equal(bits(hits)[:, 3], bits(g["cornell_hits"])[:, 3])
    h = bits(g["cornell_hits"])[:, 3] != 0xFFFFFFFF
    assert np.array_equal(bits(hits)[h], bits(g["cornell_hits"])[h])
    assert np.array_equal(ctx.trace_any(g["cornell_shadow_rays"]), g["cornell_shadow_occ"])
    assert np.array_equal(bits(ctx.surface(g["cornell_rays"], g["cornell_hits"])), bits(g["cornell_surface"]))
    assert np.array_equal(bits(ctx.lights()), bits(g["cornell_lights"]))
    for i in range(3):
        vals, end = ctx.tea(tuple(int(v) for v in g[f"tea{i}_seed"]), 8)
        assert np.array_equal(bits(vals), bits(g[f"tea{i}_vals"])) and end == tuple(int(v) for v in g[f"tea{i}_end"])
    vp = rt.Scene.cornell().view_proj(48 / 28)
    ctx.set_camera(*vp); ctx.set_camera(*vp)
    ctx.clear(48, 28); ctx.render_v6_pass1(rt.Params(width=48, height=28, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=3))
    di, gi, sd = ctx.read_pass1_buffers()
    assert np.array_equal(bits(ctx.read_accum()), bits(g["pass1_accum"])) and np.array_equal(di, g["pass1_di"]) and np.array_equal(gi, g["pass1_gi"]) and np.array_equal(sd, g["pass1_sd"])
    ctx.restir_reset(); ctx.clear(48, 28); ctx.render_restir(rt.Params(width=48, height=28, spp=2, max_bounces=3, nee_samples=4, flags=0, frame_seed=3))
    ld, lg, ls = ctx.read_restir_last()
    assert np.array_equal(bits(ctx.read_accum()), bits(g["restir_accum"]))
    assert np.array_equal(ld, g["restir_last_di"]) and np.array_equal(lg, g["restir_last_gi"]) and np.array_equal(ls, g["restir_last_sd"])
    ctx.set_camera(*rt.Scene.cornell().view_proj(64 / 36))
    for tag, kw in (("c1", dict(spp=1, max_bounces=4)), ("c2", dict(spp=4, max_bounces=8))):
        ctx.clear(64, 36); ctx.render(rt.Params(width=64, height=36, nee_samples=1, flags=1, **kw))
        assert np.array_equal(bits(ctx.read_accum()), bits(g[f"cornell_{tag}_accum"]))
        st = ctx.stats()
        assert (st.rays_primary, st.rays_extension, st.rays_shadow) == tuple(int(v) for v in g[f"cornell_{tag}_rays"])


def test_renderer_facade_lifecycle(rt, cornell):
    """OnInit / OnUpdate / OnRender of the headless Renderer (Renderer.h:46-51): progressive accumulation"""
    r = rt.Renderer(64, 36, "test", 0)
    r.set_scene(cornell)
    r.params.spp = 2; r.params.flags = 1; r.params.max_bounces = 4
    r.on_init()
    for _ in range(3):
        r.on_update(); r.on_render()
    acc = r.read_accum()
    assert (acc[..., 3] == 6).all() and acc[..., :3].sum() > 0
    out = r.read_output()
    assert out.shape == (36, 64, 4) and (out[..., 3] == 255).all()
    r.close()


def test_renderer_facade_issues_the_restir_frame(rt, orc, golden_dir):
    """the reference's shipping frame — three DispatchRays per OnRender (Renderer.cpp:646-673) — reachable from the C++ host facade:
    Renderer::SetMode(ReSTIR) makes OnRender issue rtx_render_restir with the reference's defines (nee 4, bounces 3), history carried between frames.  Three frames
    on garage.obj + monke.obj (the reference's own start-up scene and camera) equal the oracle's three frames bit for bit, as wavefront stages and literally."""
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 96, 56
    o = orc.Oracle().load(sc, W / H)
    acc_o, st = np.zeros((H, W, 4), np.float32), None
    for k in range(3):
        o.set_camera(*sc.view_proj(W / H))                   # OnUpdate sets the camera every frame (previous view = this view from the second frame on)
        acc_o, st, _ = o.restir_frames(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=k + 1), acc_o, st)
    for wave in (1, 0):
        r = rt.Renderer(W, H, "restir", 0)
        r.set_scene(sc); r.set_mode(1)
        assert (r.restir_params.nee_samples, r.restir_params.max_bounces, r.restir_params.spp) == (4, 3, 1)
        r.on_init(); r.set_option(rt.OPT_RESTIR_WAVEFRONT, wave)
        for _ in range(3):
            r.on_update(); r.on_render()                     # frame_seed = m_time = 1, 2, 3
        acc = r.read_accum()
        assert np.array_equal(bits(acc), bits(acc_o)), wave
        assert (acc[..., 3] == 3).all()
        r.close()


def test_debug_output_layers(rt, orc, cornell, golden_dir):
    """the reference's 30-layer gOutput and its 'C' key (Renderer.h:298-299, Renderer.cpp:690-698, 748-754): layer 0 is the image, layers 10-17 are
    first-hit attributes (recomputed here from the oracle's primary rays / closest hits / ClosestHit surfaces, in float32 like the kernel), the rest is black"""
    garage = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    f32 = np.float32

    def rgb8(v):
        v = np.clip(np.asarray(v, np.float32), f32(0), f32(1))
        return (v * f32(255.0) + f32(0.5)).astype(np.int32).astype(np.uint8)
    for sc, small in ((cornell, 1), (cornell, 0), (garage, 0)):
        W, H = 96, 54
        c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.upload(sc, W / H)
        o = orc.Oracle().load(sc, W / H)
        p = rt.Params(width=W, height=H, spp=1, flags=1)
        c.clear(W, H); c.render(p)
        assert np.array_equal(c.read_layer(0), c.read_srgb8())
        rays = o.primary_rays(p); hits = o.trace_closest(rays, 1); sf = o.surface(rays, hits)
        hit = bits(hits)[:, 3] != 0xFFFFFFFF
        mat = bits(sf)[:, 3]; kd = np.array([[rt.half_round(float(x)) for x in sc.materials[m][0:3]] if h else [0, 0, 0] for m, h in zip(mat, hit)], np.float32)
        t, u, v = hits[:, 0], hits[:, 1], hits[:, 2]
        expect = {10: rgb8(sf[:, 4:7] * f32(0.5) + f32(0.5)), 11: rgb8(np.repeat((t / (f32(1.0) + t))[:, None], 3, 1)), 13: rgb8(kd),
                  15: rgb8(np.stack([f32(1.0) - u - v, u, v], 1))}
        for layer, rgb in expect.items():
            got = c.read_layer(layer).reshape(-1, 4)
            assert (got[:, 3] == 255).all() and (got[~hit, :3] == 0).all(), layer
            assert np.array_equal(got[hit, :3], rgb[hit]), (layer, small)
        ids = c.read_layer(12).reshape(-1, 4)
        for m in np.unique(mat[hit]):                                  # one colour per material id, different ids differ
            assert len(np.unique(ids[hit & (mat == m)][:, :3], axis=0)) == 1
        assert len(np.unique(ids[hit][:, :3], axis=0)) == len(np.unique(mat[hit]))
        for layer in (1, 9, 20, 28, 29):
            g = c.read_layer(layer); assert (g[..., :3] == 0).all() and (g[..., 3] == 255).all()
        with pytest.raises(rt.RtxError):
            c.read_layer(30)
        c.close()
    r = rt.Renderer(64, 36, "layers"); r.set_scene(cornell); r.params.flags = 1; r.on_init(); r.on_update(); r.on_render()
    seen = [r.display_layer]
    img0 = r.read_output()
    for _ in range(18):
        r.on_key_up("C"); seen.append(r.display_layer)
    assert seen == [0, 10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 23, 24, 25, 26, 27, 28, 0]      # m_displayLevels, wrapping around
    assert np.array_equal(r.read_output(), img0)
    r.on_key_up("C"); n10 = r.read_output()
    assert r.display_layer == 10 and not np.array_equal(n10, img0) and (n10[..., 3] == 255).all()
    r.close()


def test_error_paths(rt, cornell):
    c = rt.Context(0)
    with pytest.raises(rt.RtxError):
        c.render(rt.Params(width=8, height=8))                       # not committed
    c.upload(cornell, 1.0)
    with pytest.raises(rt.RtxError):
        c.render(rt.Params(width=8, height=8, max_bounces=0))
    with pytest.raises(rt.RtxError):
        c.render(rt.Params(width=8, height=8, tile_size=24))         # not a power of two
    with pytest.raises(rt.RtxError):
        c.add_mesh(np.zeros((3, 7), np.float32), [0, 1, 5], [0, 0, 0])   # index out of range
    c.render(rt.Params(width=8, height=8, spp=0))                    # empty work is fine
    c.close()


class SoupScene:
    """a raw triangle soup in the reference's data model (one mesh, one instance, one grey material)"""
    def __init__(self, tris, o2w=None):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 3, 3)
        n = len(tris)
        v = np.zeros((n * 3, 7), np.float32); v[:, 0:3] = tris.reshape(-1, 3)          # normal (0,0,0) = flat, material base 0
        m = np.zeros((1, 32), np.float32); m[0, 0:4] = (0.7, 0.7, 0.7, 1); m[0, 4:8] = (1, 1, 1, 1)
        self.materials = m
        self.meshes = [(v, np.arange(n * 3, dtype=np.uint32), np.zeros(n * 3, np.uint32))]
        self.instances = [(0, np.eye(4, dtype=np.float32).reshape(16) if o2w is None else np.asarray(o2w, np.float32).reshape(16))]
        self.tris = tris

    def view_proj(self, aspect):
        e = np.eye(4, dtype=np.float32).reshape(16)
        return e, e


def soup(kind, n, rng):
    c = rng.uniform(-1, 1, (n, 1, 3))
    if kind == "random":                 # small triangles everywhere
        t = c + rng.normal(scale=0.03, size=(n, 3, 3))
    elif kind == "coplanar":             # everything in the plane z = 0.25: zero-extent node axis, rays inside the plane
        t = c + rng.normal(scale=0.05, size=(n, 3, 3)); t[..., 2] = 0.25
    elif kind == "far_offset":           # a small model far from the origin: float cancellation in (p - o) * idir
        t = c * 0.5 + rng.normal(scale=0.02, size=(n, 3, 3)) + np.array([1000.0, -2000.0, 500.0])
    elif kind == "needles":              # long thin triangles: big boxes, heavy overlap
        a = rng.uniform(-1, 1, (n, 3)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        t = np.stack([a, a + 1.5 * d, a + 1.5 * d + rng.normal(scale=0.002, size=(n, 3))], axis=1)
    elif kind == "duplicates":           # many identical triangles (ties -> lowest id) + degenerate ones
        base = c[: n // 8] + rng.normal(scale=0.05, size=(n // 8, 3, 3))
        t = np.concatenate([np.tile(base, (7, 1, 1)), np.repeat(c[: n - 7 * (n // 8)], 3, axis=1)])
    elif kind == "mixed_scale":          # 1e-4-sized clusters inside a 100-unit scene: deep exponent range in one tree
        big = rng.uniform(-50, 50, (n // 2, 1, 3)) + rng.normal(scale=3.0, size=(n // 2, 3, 3))
        small = rng.uniform(-1, 1, (n - n // 2, 1, 3)) * 0.01 + rng.normal(scale=1e-4, size=(n - n // 2, 3, 3))
        t = np.concatenate([big, small])
    return t.astype(np.float32)


@pytest.mark.parametrize("builder", ["default", "split", "gpu"])
@pytest.mark.parametrize("kind", ["random", "coplanar", "far_offset", "needles", "duplicates", "mixed_scale"])
def test_wide_bvh_equals_brute_force_on_hostile_soups(rt, orc, kind, builder):
    """the compressed 8-wide BVH (byte-quantised child boxes, octant-ordered traversal) must return exactly the brute-force
    minimum over all triangles, ties to the lowest id, on geometry chosen to stress the quantisation and its margins —
    with the default builder and with spatial splits on (a triangle referenced from several leaves, each with the box of its part)"""
    rng = np.random.default_rng(sum(map(ord, kind)) + 7 + FUZZ_SEED)
    n = 6000
    t = soup(kind, n, rng)
    sc = SoupScene(t)
    c = rt.Context(0)
    if builder == "split":
        c.set_option(rt.OPT_BVH_SPLIT, 1000); c.set_option(rt.OPT_BVH_REINSERT, 3)      # overlap threshold 1e-6 of the scene's area, three re-insertion passes
    if builder == "gpu":
        c.set_option(rt.OPT_GPU_BUILD, 1)                                                # the tree built on the device (csrc/rtx_build.hip)
    c.upload(sc, 1.0)
    o = orc.Oracle().load(sc, 1.0)
    assert c.stats().triangles == len(t) and c.stats().bvh_refs >= len(t)                # leaf entries: spatial splits add references
    assert (c.stats().bvh_refs == len(t)) if builder != "split" else (c.stats().bvh_refs > len(t) or kind != "needles")
    assert (c.build_info()["clusters_top"] > 0) == (builder == "gpu")           # (6 000 triangles are below ploc_top: no PLOC round, the top-down builder over single-triangle clusters)
    assert c.validate_bvh() == 0
    lo, hi = t.reshape(-1, 3).min(0), t.reshape(-1, 3).max(0)
    ext = float((hi - lo).max())
    m = 60000
    org = rng.uniform(lo - 0.1 * ext, hi + 0.1 * ext, (m, 3))
    d = rng.normal(size=(m, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a third of the rays aim at triangle interiors / vertices / edges from nearby points, some are axis-parallel
    k = m // 3
    pick = rng.integers(0, len(t), k); w = rng.dirichlet((1, 1, 1), k); w[: k // 4] = np.eye(3)[rng.integers(0, 3, k // 4)]      # exact vertices
    w[k // 4: k // 2, 2] = 0; w[k // 4: k // 2, :2] /= np.maximum(w[k // 4: k // 2, :2].sum(1, keepdims=True), 1e-9)             # on an edge
    tgt = (t[pick] * w[:, :, None]).sum(1)
    d[:k] = tgt - org[:k]; d[:k] /= np.maximum(np.linalg.norm(d[:k], axis=1, keepdims=True), 1e-30)
    d[k: k + 2000] = np.eye(3)[rng.integers(0, 3, 2000)] * rng.choice([-1.0, 1.0], (2000, 1))                                   # zero components
    if kind == "coplanar":
        org[-4000:, 2] = 0.25; d[-4000:, 2] = 0.0; d[-4000:] /= np.maximum(np.linalg.norm(d[-4000:], axis=1, keepdims=True), 1e-30)  # rays IN the plane
    else:                                                                                                                       # ... and, for every other soup, rays lying in the plane of ONE triangle: Moeller-Trumbore's 0 / 0 case
        pl = rng.integers(0, len(t), 4000); T = t[pl].astype(np.float64); a = rng.normal(scale=3.0, size=(4000, 2)); bb = rng.normal(size=(4000, 2))
        org[-4000:] = T[:, 0] + a[:, :1] * (T[:, 1] - T[:, 0]) + a[:, 1:] * (T[:, 2] - T[:, 0])
        d[-4000:] = bb[:, :1] * (T[:, 1] - T[:, 0]) + bb[:, 1:] * (T[:, 2] - T[:, 0]); d[-4000:] /= np.maximum(np.linalg.norm(d[-4000:], axis=1, keepdims=True), 1e-30)
    rays = np.zeros((m, 8), np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = org, 1e-5, d, 1e30
    g = c.trace_closest(rays)
    b = o.trace_closest(rays, mode=0)                                      # brute force over all triangles
    hit = bits(b)[:, 3] != 0xFFFFFFFF
    # EVERY ray must agree.  Until round 4 this test counted and excluded the rays for which float Moeller-Trumbore ACCEPTS a point nowhere near a sliver (a ray in the
    # triangle's plane: |det| is rounding noise, u = v = -0, t arbitrary — brute force reported it, every tree culled it); the hit definition now carries a determinant floor
    # (csrc/rtx_math.hpp: tri_det_floor, oracle/rt_oracle.c: tri_hit) and such a "hit" no longer exists on either side.  What is left must be consistent: the reported point
    # o + t d lies on the reported triangle to within 2 % of the scene for all rays (in float64, no code of the repository).
    pid = bits(b)[hit, 3].astype(np.int64)
    P = rays[hit, 0:3].astype(np.float64) + b[hit, 0:1].astype(np.float64) * rays[hit, 4:7].astype(np.float64)
    tv = t[pid].astype(np.float64)
    Q = tv[:, 0] + b[hit, 1:2].astype(np.float64) * (tv[:, 1] - tv[:, 0]) + b[hit, 2:3].astype(np.float64) * (tv[:, 2] - tv[:, 0])
    assert (np.abs(P - Q).max(1) <= 0.02 * ext).all(), f"{kind}: a brute-force hit lies nowhere near its triangle"
    bad = bits(g)[:, 3] != bits(b)[:, 3]
    assert not bad.any(), f"{kind}: {int(bad.sum())} hit ids differ, first ray {rays[bad][0]}, gpu {g[bad][0]}, cpu {b[bad][0]}"
    assert np.array_equal(bits(g)[hit], bits(b)[hit]), "t/u/v differ"
    assert np.array_equal(bits(g), bits(o.trace_closest(rays, mode=1))), "GPU and the oracle's own BVH differ"
    assert hit.mean() > 0.05
    sh = rays.copy(); sh[:, 7] = rng.uniform(0.05, 1.0, m).astype(np.float32) * ext
    ga, ba = c.trace_any(sh), o.trace_any(sh, mode=0)
    assert np.array_equal(ga, ba) and np.array_equal(ga, o.trace_any(sh, mode=1))
    for order in (0, 1, 2):                                               # any-hit is existence: the visiting order of a node's children changes no answer
        c.set_option(rt.OPT_ANYHIT_ORDER, order)
        assert np.array_equal(c.trace_any(sh), ga), f"any-hit order {order}"
    c.close()


@pytest.mark.parametrize("kind", ["atrium", "needles"])
def test_traversal_stack_overflow_columns_change_nothing(rt, orc, kind):
    """RTX_OPT_STACK_CAP (round 5): traversal-stack entries beyond the cap live in per-lane columns in global memory instead of LDS.  With a cap of 4 (every tree here needs more)
    the persistent closest-hit and any-hit kernels, the ReSTIR stages' visibility kernel and the debug queries all run through the overflow path: image, ray counts, closest hits
    and occlusion answers must equal the uncapped context's and the oracle's."""
    rng = np.random.default_rng(5)
    sc = rt.Scene.sponza_class(60000, 260) if kind == "atrium" else SoupScene(soup("needles", 20000, rng))
    W, H = 96, 54
    vp = sc.view_proj(W / H) if kind == "atrium" else (rt.lookat((0.2, 0.3, 2.6), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0))
    p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1, flags=1)
    o = orc.Oracle().load(sc, W / H); o.set_camera(*vp)
    oa, oc = o.render(p)
    rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H), 1)[:2000], random_rays(3000, 5, -1.5, 1.5)])
    ob = o.trace_closest(rays, 1)
    sh = rays.copy(); sh[:, 7] = rng.uniform(0.05, 1.0, len(sh)).astype(np.float32) * 3.0
    oany = o.trace_any(sh, mode=1)
    for cap in (0, 4, 6):
        c = rt.Context(0); c.set_option(rt.OPT_STACK_CAP, cap); c.upload(sc, W / H); c.set_camera(*vp)
        assert c.validate_bvh() == 0
        c.clear(W, H); c.render(p); st = c.stats()
        assert np.array_equal(bits(c.read_accum()), bits(oa)) and (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, f"cap {cap}"
        assert np.array_equal(bits(c.trace_closest(rays)), bits(ob)) and np.array_equal(c.trace_any(sh), oany), f"cap {cap}"
        if kind == "atrium":                                 # a ReSTIR frame: its stages' traversal launches take the same columns
            pr = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=2, flags=0)
            c.restir_reset(); c.clear(W, H); c.render_restir(pr)
            img = c.read_accum()
            if cap == 0: ref_restir = img
            else: assert np.array_equal(bits(img), bits(ref_restir)), f"ReSTIR frame, cap {cap}"
        c.close()


def test_host_state_survives_hundreds_of_contexts(rt):
    """Round 5 (profiles/r05_determinism.md): the once-in-700-contexts mismatch of the fuzz tests was two words of a live 912-byte host vector (a mesh's index array, the
    builder's leaf order) changing during a later rtx_commit_scene — a stray write that came with hipStreamDestroy in a process that creates and destroys many contexts.  The
    library pools its streams now.  This is tools/flaky_bisect.py in small: the failing flow on the two scenes it hit (their 228-entry vectors), RTX_CONTEXT_REPS contexts each
    (default 400; the hunt ran 20 000), host checksums after every call: nothing the host holds may change between commits, and every context ends with the same tree."""
    reps = int(os.environ.get("RTX_CONTEXT_REPS", 400))
    W, H = 48, 32
    for seed in (817, 148):
        sc = RandomTinyScene(rt, 9000 + 200000 + seed, max_tris=[200, 800, 3000][seed % 3])
        M = np.eye(4); M[:3, :3] = np.diag([1.1, 0.9, -1.05]) @ np.array([[np.cos(.3), 0, np.sin(.3)], [0, 1, 0], [-np.sin(.3), 0, np.cos(.3)]]); M[:3, 3] = (0.05, -0.02, 0.03)
        inst = len(sc.instances) - 1
        M2 = (M @ np.asarray(sc.instances[inst][1], np.float64).reshape(4, 4).T).T.astype(np.float32).reshape(16)
        rays = random_rays(3000, seed, -1.2, 1.2)
        p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1 + seed % 2, flags=seed & 1, frame_seed=seed)
        built = refit = tree = None
        for rep in range(reps):
            c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, 0); c.upload(sc, W / H)
            h0 = c.host_checksums()
            c.clear(W, H); c.render(p); c.trace_closest(rays); assert c.validate_bvh() == 0
            assert c.host_checksums() == h0, f"seed {seed} context {rep}: host state changed between two commits"
            c.set_instance_transform(inst, M2); c.commit()
            h1 = c.host_checksums()
            assert h1[:3] == h0[:3] and h1[4] == h0[4], f"seed {seed} context {rep}: meshes, materials or the leaf order changed in a transform-only commit"
            c.clear(W, H); c.render(p); assert c.validate_bvh() == 0
            t = c.tree_hash()
            if rep == 0: built, refit, tree = h0, h1, t
            assert (h0, h1, t) == (built, refit, tree), f"seed {seed} context {rep}: not the state of the first context"
            c.close()


@pytest.mark.parametrize("kind", ["atrium", "atrium_hard", "street", "soup", "needles"])
def test_gpu_build_equals_its_host_twin_and_renders_the_oracle_image(rt, orc, kind):
    """RTX_OPT_GPU_BUILD (csrc/rtx_build.hip; VERDICT r04 item 6: the reference's driver builds its BLAS / TLAS on the device, BottomLevelASGenerator.cpp:178-247,
    TopLevelASGenerator.cpp:149-250): a geometry-changing commit builds the wide tree ON THE GPU — Morton sort, PLOC rounds, the top over <= 16 384 clusters by the host's SAH builder,
    SAH collapse and layout on the device, boxes by the refit kernels.  (1) The tree is valid (every triangle in exactly one leaf slot, inside every decoded box above it).
    (2) It is the tree its HOST TWIN builds (the host builder with BvhBuildOptions::ploc_radius, same decisions from shared code: rtx_wide.hpp) node for node and triangle
    for triangle, once both have been quantised by the same refit kernels.  (3) Image, ray counts and closest-hit records equal the oracle's — results do not depend on the
    tree.  (4) A second geometry change (a mesh added) rebuilds on the GPU and is exact again."""
    rng = np.random.default_rng(23)
    if kind == "atrium": sc = rt.Scene.sponza_class(60000, 260)
    elif kind == "atrium_hard": sc = rt.Scene.sponza_class(60000, 260, hard=True)
    elif kind == "street": sc = rt.Scene.bistro_class(300000, 3800)
    else:
        sc = SoupScene(soup("random" if kind == "soup" else "needles", 20000, rng))
        sc.materials = np.concatenate([sc.materials, sc.materials]); sc.materials[1, 8:11] = (5.0, 4.0, 3.0)           # an emissive material for the added mesh of step (4)
    W, H = 96, 54
    p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1, flags=0 if kind == "street" else 1)
    vp = sc.view_proj(W / H) if kind in ("atrium", "atrium_hard", "street") else (rt.lookat((0.2, 0.3, 2.6), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0))
    a = rt.Context(0); a.set_option(rt.OPT_GPU_BUILD, 1); a.upload(sc, W / H); a.set_camera(*vp)
    info = a.build_info()
    assert info["ploc_rounds"] >= (2 if a.stats().triangles > 40000 else 0) and 1 <= info["clusters_top"] <= 16384 and info["refs"] == a.stats().triangles and sum(info["ms"]) > 0
    assert a.validate_bvh() == 0
    rt.bvh_option("ploc", 16)
    try:
        b = rt.Context(0)
    finally:
        rt.bvh_option("ploc", 0)
    b.upload(sc, W / H); b.set_camera(*vp)
    assert b.build_info()["clusters_top"] == 0                                     # built on the host
    b.set_instance_transform(0, sc.instances[0][1]); b.commit()                    # a full refit on the GPU: both trees quantised by the same kernels
    assert b.stats().bvh_refits == 1
    assert a.tree_hash() == b.tree_hash(), "the GPU-built tree differs from its host twin"
    o = orc.Oracle().load(sc, W / H); o.set_camera(*vp)
    oa, oc = o.render(p)
    rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H), 1)[:2000], random_rays(3000, 5, -1.5, 1.5)])
    ob = o.trace_closest(rays, 1)
    for c in (a, b):
        c.clear(W, H); c.render(p); st = c.stats()
        assert np.array_equal(bits(c.read_accum()), bits(oa)) and (st.rays_primary, st.rays_extension, st.rays_shadow) == oc
        assert np.array_equal(bits(c.trace_closest(rays)), bits(ob))
    b.close()
    # (4) a geometry change: one more mesh (an emissive quad hanging in the scene) -> GPU rebuild
    quad = np.array([[-0.3, 0.6, -0.3], [0.3, 0.6, -0.3], [0.3, 0.6, 0.3], [-0.3, 0.6, -0.3], [0.3, 0.6, 0.3], [-0.3, 0.6, 0.3]], np.float32) * (1.0 if kind in ("soup", "needles") else 0.5)
    nmid = sum(len(m[2]) for m in sc.meshes)
    v = np.zeros((6, 7), np.float32); v[:, 0:3] = quad; v[:, 6] = nmid
    lamp = len(sc.materials) - 1 if kind in ("soup", "needles") else {"atrium": 12, "atrium_hard": 12, "street": 39}[kind]
    mesh = a.add_mesh(v, np.arange(6, dtype=np.uint32), np.full(6, lamp, np.uint32)); a.add_instance(mesh, np.eye(4, dtype=np.float32).reshape(16)); a.commit()
    assert a.validate_bvh() == 0 and a.build_info()["refs"] == info["refs"] + 2 and a.stats().bvh_refits == 0

    class Plus:
        materials = sc.materials; meshes = list(sc.meshes) + [(v, np.arange(6, dtype=np.uint32), np.full(6, lamp, np.uint32))]
        instances = list(sc.instances) + [(len(sc.meshes), np.eye(4, dtype=np.float32).reshape(16))]
        def view_proj(self, aspect): return vp
    o2 = orc.Oracle().load(Plus(), W / H)
    oa2, oc2 = o2.render(p)
    a.clear(W, H); a.render(p); st = a.stats()
    assert np.array_equal(bits(a.read_accum()), bits(oa2)) and (st.rays_primary, st.rays_extension, st.rays_shadow) == oc2
    a.close()


def test_gpu_built_tree_renders_the_full_size_c3_frame_of_the_host_built_one(rt):
    """BASELINE configs[2] at full size (Sponza-class, 1080p, 16 spp, 8 bounces: 264 M rays) on the GPU-built tree and on the host-built one: the same bits, the same ray
    counts (test_full_size_baseline_configs_are_bit_identical ties the host-built frame to the oracle), and a frame time within 10 % (the bench's extra record has the figures)."""
    import time
    sc = rt.Scene.sponza_class()
    W, H = 1920, 1080
    p = rt.Params(width=W, height=H, spp=16, max_bounces=8, nee_samples=1, rr_start=3, flags=1)
    out = []
    for gpu in (0, 1):
        c = rt.Context(0); c.set_option(rt.OPT_GPU_BUILD, gpu)
        t0 = time.perf_counter(); c.upload(sc, W / H); commit_s = time.perf_counter() - t0
        c.clear(W, H); c.render(p)                                   # warm-up
        ms = []
        for k in range(3):
            c.clear(W, H); t0 = time.perf_counter(); c.render(p); ms.append((time.perf_counter() - t0) * 1e3)
        st = c.stats()
        out.append((c.read_accum(), (st.rays_primary, st.rays_extension, st.rays_shadow), min(ms), commit_s, c.build_info()))
        c.close()
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][1] == out[1][1]
    print(f"C3 frame: host-built tree {out[0][2]:.2f} ms (commit {out[0][3]:.2f} s), GPU-built {out[1][2]:.2f} ms (commit {out[1][3]:.2f} s, {out[1][4]})")
    assert out[1][2] <= 1.10 * out[0][2]


@pytest.mark.parametrize("kind,tris,cfg", [
    ("sponza", 262144, dict(width=160, height=90, spp=2, max_bounces=8, nee_samples=1, flags=1)),          # C3 settings, small image
    ("bistro", 300000, dict(width=128, height=72, spp=2, max_bounces=8, nee_samples=1, flags=0)),          # C5 materials (GGX + NEE), reduced triangle count
    ("bistro", 300000, dict(width=128, height=72, spp=4, max_bounces=8, nee_samples=1, flags=4)),          # ... with the dielectric panes transmitting (strategy-3 extension)
])
def test_large_scene_parity(rt, orc, kind, tris, cfg):
    """deep-BVH scenes through the general traversal path: hit records equal the oracle's, images within 1e-4"""
    sc = rt.Scene.sponza_class(tris, 260) if kind == "sponza" else rt.Scene.bistro_class(tris, 3800)
    p = rt.Params(**cfg)
    aspect = p.width / p.height
    c = rt.Context(0); c.upload(sc, aspect)
    o = orc.Oracle().load(sc, aspect)
    st0 = None
    rays = np.concatenate([o.primary_rays(p), random_rays(40000, 31, -1.5, 1.5)])
    g, b = c.trace_closest(rays), o.trace_closest(rays, 1)
    assert np.array_equal(bits(g)[:, 3], bits(b)[:, 3]), f"{int((bits(g)[:, 3] != bits(b)[:, 3]).sum())} hit ids differ"
    hit = bits(b)[:, 3] != 0xFFFFFFFF
    assert np.array_equal(bits(g)[hit], bits(b)[hit])
    sr = random_rays(40000, 32, -1.0, 1.0, tmax=0.7)
    assert np.array_equal(c.trace_any(sr), o.trace_any(sr, 1))
    assert np.array_equal(bits(c.surface(rays, b)), bits(o.surface(rays, b)))
    c.clear(p.width, p.height); c.render(p)
    ga = c.read_accum(); ca, cnt = o.render(p)
    st = c.stats()
    print(kind, "tris", st.triangles, "nodes", st.bvh_nodes, "lights", st.lights, "rays", cnt, "render_ms", round(st.render_ms, 2))
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt
    r = rel_l2(ga[..., :3], ca[..., :3])
    print("rel_l2", r, "pixels not bit exact:", int((bits(ga) != bits(ca)).any(-1).sum()))
    assert r <= REL_L2_TOL
    c.close()


RESTIR_FORMS = [pytest.param(1, id="wavefront"), pytest.param(0, id="literal")]     # RTX_OPT_RESTIR_WAVEFRONT: stage kernels + persistent traversal | one thread per pixel and pass


@pytest.mark.parametrize("wave", RESTIR_FORMS)
@pytest.mark.parametrize("flags,nee,bounces", [(1, 4, 3), (0, 4, 3), (0, 1, 2), (1, 0, 3), (0, 2, 0)])
def test_v6_pass1_estimator_parity(rt, cornell_pair, flags, nee, bounces, wave):
    """the reference's own pass 1 (SampleRIS + SamplePathSimple, RayGen_v6_pass1.hlsl:48-190): reservoirs,
    sample data (reference byte layouts, MapPixelID order) and radiance, bit for bit against the oracle — as wavefront stages (the default) and literally"""
    ctx, o = cornell_pair
    ctx.set_option(rt.OPT_RESTIR_WAVEFRONT, wave)
    W, H = 100, 58                                        # not a multiple of the 4x4 MapPixelID tile
    p = rt.Params(width=W, height=H, spp=2, max_bounces=bounces, nee_samples=nee, flags=flags, frame_seed=5)
    ctx.set_camera(*rt.Scene.cornell().view_proj(W / H)); o.set_camera(*rt.Scene.cornell().view_proj(W / H))
    ctx.clear(W, H); ctx.render_v6_pass1(p)
    ctx.set_option(rt.OPT_RESTIR_WAVEFRONT, 1)
    g = ctx.read_accum(); gd, gg, gs = ctx.read_pass1_buffers()
    c, (cd, cg, cs), cnt = o.render_v6_pass1(p)
    st = ctx.stats()
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt
    assert np.array_equal(gs, cs), f"SampleData differs in {int((gs != cs).any(1).sum())} records"
    assert np.array_equal(gd, cd), f"Reservoir_DI differs in {int((gd != cd).any(1).sum())} records"
    assert np.array_equal(gg, cg), f"Reservoir_GI differs in {int((gg != cg).any(1).sum())} records"
    assert np.array_equal(bits(g), bits(c))
    assert g[..., :3].sum() > 0 and (g[..., 3] == 2).all()


@pytest.mark.parametrize("wave", RESTIR_FORMS)
def test_v6_pass1_garage(rt, orc, golden_dir, wave):
    import os
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    c = rt.Context(0); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, 96 / 54)
    o = orc.Oracle().load(sc, 96 / 54)
    p = rt.Params(width=96, height=54, spp=1, max_bounces=3, nee_samples=4, flags=0)     # the reference's defines (Common_v6.hlsl:8-12)
    c.clear(96, 54); c.render_v6_pass1(p)
    g = c.read_accum(); gd, gg, gs = c.read_pass1_buffers()
    ca, (cd, cg, cs), cnt = o.render_v6_pass1(p)
    st = c.stats()
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == cnt
    assert np.array_equal(gs, cs) and np.array_equal(gd, cd) and np.array_equal(gg, cg)
    assert rel_l2(g[..., :3], ca[..., :3]) <= REL_L2_TOL
    c.close()


@pytest.mark.parametrize("wave", RESTIR_FORMS)
@pytest.mark.parametrize("flags", [1, 0])
def test_restir_frames_parity(rt, cornell_pair, flags, wave):
    """the reference's shipping pipeline: pass 1 + temporal reuse (pass 2) + spatial reuse and shade (pass 3), three
    consecutive frames so that the temporal history is exercised; all six reservoir / sample buffers and the image"""
    ctx, o = cornell_pair
    ctx.set_option(rt.OPT_RESTIR_WAVEFRONT, wave)
    W, H = 96, 56
    vp = rt.Scene.cornell().view_proj(W / H)
    ctx.set_camera(*vp); ctx.set_camera(*vp); o.set_camera(*vp); o.set_camera(*vp)      # previous view = current view
    p = rt.Params(width=W, height=H, spp=3, max_bounces=3, nee_samples=4, flags=flags, frame_seed=11)
    ctx.restir_reset(); ctx.clear(W, H); ctx.render_restir(p)
    ctx.set_option(rt.OPT_RESTIR_WAVEFRONT, 1)
    g = ctx.read_accum(); gd, gg, gs = ctx.read_pass1_buffers(); ld, lg, ls = ctx.read_restir_last()
    c, st, cnt = o.restir_frames(p)
    s = ctx.stats()
    assert (s.rays_primary, s.rays_extension, s.rays_shadow) == cnt
    for name, a, b in (("cur_di", gd, st[0]), ("cur_gi", gg, st[1]), ("cur_sd", gs, st[2]), ("last_di", ld, st[3]), ("last_gi", lg, st[4]), ("last_sd", ls, st[5])):
        assert np.array_equal(a, b), f"{name} differs in {int((a != b).any(1).sum())} records"
    assert np.array_equal(bits(g), bits(c))
    M = ld.view(np.uint16).reshape(len(ld), 20)[:, 19]
    assert M.max() > 1                                               # temporal / spatial merges happened
    assert g[..., :3].sum() > 0


@pytest.mark.parametrize("wave", RESTIR_FORMS)
def test_restir_garage_with_camera_motion(rt, orc, golden_dir, wave):
    """GGX scene with two instances; the camera moves between frames so the reprojection path is taken"""
    import os
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 80, 48
    c = rt.Context(0); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, W / H)
    o = orc.Oracle().load(sc, W / H)
    acc_o, st = np.zeros((H, W, 4), np.float32), None
    c.restir_reset(); c.clear(W, H)
    proj = rt.perspective_fov_rh(np.float32(np.pi / 3), W / H, 0.1, 1000.0)
    tot = np.zeros(3, np.uint64)
    for k, eye in enumerate([(-1.5, 1.5, 3.5), (-1.45, 1.5, 3.5), (-1.4, 1.52, 3.48)]):
        view = rt.lookat(eye, (0, 1, 0), (0, 1, 0))
        c.set_camera(view, proj); o.set_camera(view, proj)
        p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=100 + k)
        c.render_restir(p)
        acc_o, st, cnt = o.restir_frames(p, acc_o, st)
        s = c.stats(); assert (s.rays_primary, s.rays_extension, s.rays_shadow) == cnt
    ld, lg, ls = c.read_restir_last()
    assert np.array_equal(ld, st[3]) and np.array_equal(lg, st[4]) and np.array_equal(ls, st[5])
    assert rel_l2(c.read_accum()[..., :3], acc_o[..., :3]) <= REL_L2_TOL
    c.close()


@pytest.mark.parametrize("nshards,wave,blocks", [(2, 1, 0), (3, 1, 0), (2, 0, 0), (2, 1, 1), (3, 1, 1), (4, 1, 1), (4, 0, 1)])
def test_restir_on_shards_equals_the_unsharded_frames(rt, orc, golden_dir, nshards, wave, blocks):
    """SURVEY 8(f1) on pixel-tile shards: every shard runs passes 1 + 2 on its tiles dilated by the 20-px radius of the spatial pass, pass 3 on its own tiles, and
    after each frame the shards exchange the history (u3 / u5 / u7) of their own tiles (pack -> gather -> unpack; the gather is a torch.cat here, the
    shards being contexts on one GPU).  Three frames with a moving camera on garage.obj + monke.obj (GGX + Lambert, two instances): every shard's tiles of the
    accumulation buffer and the complete history equal the unsharded run's (which equals the oracle's: test_restir_garage_with_camera_motion).
    blocks = RTX_FLAG_BLOCK_TILES: every shard owns ONE rectangle of tiles (a 20-px rim of recomputed pixels instead of a rim around every tile)."""
    import torch
    from royaltracer_dx_amd import sharding
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H, TS = 160, 96, 32
    cams = [rt.lookat((-1.5 + 0.05 * k, 1.5, 3.5 - 0.04 * k), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)) for k in range(3)]
    proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=rt.FLAG_BLOCK_TILES if blocks else 0, tile_size=TS)
    ref = rt.Context(0); ref.upload(sc, W / H); ref.restir_reset(); ref.clear(W, H)
    for k, v in enumerate(cams):
        ref.set_camera(v, proj); ref.render_restir(rt.Params(frame_seed=70 + k, **base))
    ref_img, ref_last = ref.read_accum(), ref.read_restir_last()
    ranks = []
    for r in range(nshards):
        c = rt.Context(0); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, W / H); c.restir_reset(); c.clear(W, H); ranks.append(c)
    for k, v in enumerate(cams):
        slabs = []
        for r, c in enumerate(ranks):
            p = rt.Params(frame_seed=70 + k, shard_rank=r, shard_count=nshards, **base)
            c.set_camera(v, proj); c.render_restir(p)
            slab = torch.empty(c.restir_state_slab_bytes(p) // 4, dtype=torch.float32, device="cuda:0")
            c.restir_pack_state(p, slab.data_ptr()); slabs.append(slab)
        torch.cuda.synchronize()
        gathered = torch.cat(slabs)                                      # what ONE all_gather_into_tensor leaves on every rank
        torch.cuda.synchronize()                                         # the cat runs on torch's stream, the unpack on each context's own (non-blocking) stream: order them
        for r, c in enumerate(ranks):
            c.restir_unpack_state(rt.Params(frame_seed=70 + k, shard_rank=r, shard_count=nshards, **base), gathered.data_ptr())
    own = sharding.owner_map(W, H, TS, nshards, bool(blocks))
    for r, c in enumerate(ranks):
        img = c.read_accum()
        assert np.array_equal(bits(img[own == r]), bits(ref_img[own == r])), r
        assert not img[own != r].any()
        for a, b in zip(c.read_restir_last(), ref_last):
            assert np.array_equal(a, b), r
        with pytest.raises(rt.RtxError):
            c.render_restir(rt.Params(frame_seed=1, shard_rank=r, shard_count=nshards, **dict(base, spp=2)))    # the history must be exchanged after every frame
        c.close()
    ref.close()


@pytest.mark.parametrize("nshards,wave,halo", [(2, 1, 32), (4, 1, 32), (4, 0, 32), (3, 1, 24), (6, 1, 32)])
def test_restir_halo_exchange_equals_the_unsharded_frames(rt, golden_dir, nshards, wave, halo):
    """SURVEY 8(f1) as it names it: "halo exchange (20 px) if tiles are sharded".  In the block deal every rank sends each neighbour only the part of its rectangle within `halo`
    px of the neighbour's (rtx_restir_halo_plan / pack_halo / unpack_halo; the point-to-point exchange is a device-side slice copy here, the ranks being contexts on one GPU) instead
    of all-gathering 140 B per pixel of the whole image.  Three frames with a moving camera on garage.obj + monke.obj: every rank's tiles of the image equal the unsharded run's,
    its history equals the unsharded history everywhere inside its rectangle + halo, and no temporal read left that region (rtx_stats.restir_stale_history_reads == 0).  Then a
    camera CUT: the reprojection leaves the halo, the count says so, and the documented fallback — all-gather of the previous history, repeat the frame — is exact again."""
    import torch
    from royaltracer_dx_amd import sharding
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H, TS = 192, 128, 32
    cams = [rt.lookat((-1.5 + 0.05 * k, 1.5, 3.5 - 0.04 * k), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)) for k in range(3)]
    cut = rt.lookat((1.2, 1.1, 3.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0))
    proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=rt.FLAG_BLOCK_TILES, tile_size=TS)
    P = lambda k, r: rt.Params(frame_seed=70 + k, shard_rank=r, shard_count=nshards, **base)
    ref = rt.Context(0); ref.upload(sc, W / H); ref.restir_reset(); ref.clear(W, H)
    for k, v in enumerate(cams):
        ref.set_camera(v, proj); ref.render_restir(rt.Params(frame_seed=70 + k, **base))
    ref_img, ref_last = ref.read_accum(), ref.read_restir_last()
    ranks = []
    for r in range(nshards):
        c = rt.Context(0); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, W / H); c.restir_reset(); c.clear(W, H); ranks.append(c)
    plans = [rt.restir_halo_plan(P(0, r), halo) for r in range(nshards)]
    assert all(len(pl[0]) <= 8 for pl in plans)

    def exchange(k):
        send = []
        for r, c in enumerate(ranks):
            b = torch.zeros(max(plans[r][1], 4), dtype=torch.uint8, device="cuda:0"); c.restir_pack_halo(P(k, r), halo, b.data_ptr()); send.append(b)
        torch.cuda.synchronize()
        for r, c in enumerate(ranks):
            recv = torch.zeros(max(plans[r][2], 4), dtype=torch.uint8, device="cuda:0")
            for e in plans[r][0]:
                back = [q for q in plans[e.rank][0] if q.rank == r][0]          # what the peer sends me == what I receive from it
                assert back.send_bytes == e.recv_bytes
                recv[e.recv_offset:e.recv_offset + e.recv_bytes] = send[e.rank][back.send_offset:back.send_offset + back.send_bytes]
            torch.cuda.synchronize()
            c.restir_unpack_halo(P(k, r), halo, recv.data_ptr())
    for k, v in enumerate(cams):
        for r, c in enumerate(ranks):
            c.set_camera(v, proj); c.render_restir(P(k, r))
            assert c.stats().restir_stale_history_reads == 0, (k, r)
        exchange(k)
    L = sharding.layout(W, H, TS, nshards, True)
    own = sharding.owner_map(W, H, TS, nshards, True)
    yy, xx = np.mgrid[0:H, 0:W]
    slot = (((yy >> 2) * ((W + 3) >> 2) + (xx >> 2)) * 16 + (yy & 3) * 4 + (xx & 3))            # MapPixelID, Common_v6.hlsl:173-198
    for r, c in enumerate(ranks):
        img = c.read_accum()
        assert np.array_equal(bits(img[own == r]), bits(ref_img[own == r])), r
        x0, y0, x1, y1 = sharding.block_rect(L, W, H, r)
        valid = slot[max(y0 - halo, 0):min(H, y1 + halo), max(x0 - halo, 0):min(W, x1 + halo)].reshape(-1)
        for a, b in zip(c.read_restir_last(), ref_last):
            assert np.array_equal(a[valid], b[valid]), r
    # a camera cut: pixels reproject far outside the halo
    ref.set_camera(cut, proj); ref.render_restir(rt.Params(frame_seed=99, **base)); ref_img2 = ref.read_accum()
    slabs = []
    for r, c in enumerate(ranks):          # keep the pre-cut history for the fallback: every rank's own tiles (the all-gather's input)
        slab = torch.empty(c.restir_state_slab_bytes(P(3, r)) // 4, dtype=torch.float32, device="cuda:0"); c.restir_pack_state(P(3, r), slab.data_ptr()); slabs.append(slab)
    torch.cuda.synchronize(); gathered = torch.cat(slabs); torch.cuda.synchronize()
    before = [c.read_accum() for c in ranks]
    stale = 0
    for r, c in enumerate(ranks):
        c.set_camera(cut, proj); c.render_restir(P(29, r)); stale += c.stats().restir_stale_history_reads
    assert stale > 0, "a camera cut must be reported: the temporal pass read history this rank does not hold"
    # fallback: restore the image, all-gather the history the frame should have seen, repeat the frame (the cut camera is already set: set it again so that the
    # previous-view matrices are the cut frame's own predecessor as in the reference run — rtx_set_camera shifts them)
    for r, c in enumerate(ranks):
        acc = torch.from_numpy(before[r]).to("cuda:0"); c.bind_accum(acc.data_ptr(), acc.numel() * 4)
        c.restir_unpack_state(P(3, r), gathered.data_ptr())
        c.set_camera(cams[2], proj); c.set_camera(cut, proj)
        c.render_restir(P(29, r))
        assert c.stats().restir_stale_history_reads == 0
        img = acc.cpu().numpy()
        assert np.array_equal(bits(img[own == r]), bits(ref_img2[own == r])), r
        c.bind_accum(0, 0); c.close()
    ref.close()


def test_restir_wavefront_equals_literal_on_a_bvh_scene_at_scale(rt, orc):
    """The wavefront stages at a size where their machinery is exercised — many workgroups with several 256-pixel chunks per sub-queue, persistent traversal
    kernels that refill their lanes, a ray queue of several rays per pixel — on a 20 k-triangle atrium (compressed wide BVH): three frames with a moving camera,
    all six buffers, the image and the ray counts equal the literal thread-per-pixel form's AND the oracle's; RTX_OPT_RESTIR_CHUNKS 1 / 4 / 16 (sub-queue
    lengths) give the same bytes."""
    sc = rt.Scene.sponza_class(20000, 260)
    W, H = 384, 216
    proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    v0 = sc.view_proj(W / H)[0]
    cams = []
    for k in range(3):
        v = v0.copy(); v[12] += 0.02 * k; v[13] -= 0.01 * k; cams.append(v)       # small translations of the view
    o = orc.Oracle().load(sc, W / H)
    acc_o, st, total = np.zeros((H, W, 4), np.float32), None, np.zeros(3, np.uint64)
    for k, v in enumerate(cams):
        o.set_camera(v, proj)
        acc_o, st, cnt = o.restir_frames(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=30 + k), acc_o, st)
        total += np.array(cnt, np.uint64)
    for wave, chunks in ((1, 4), (1, 1), (1, 16), (0, 4)):
        c = rt.Context(0); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.set_option(rt.OPT_RESTIR_CHUNKS, chunks); c.upload(sc, W / H)
        c.restir_reset(); c.clear(W, H)
        rays = np.zeros(3, np.uint64)
        for k, v in enumerate(cams):
            c.set_camera(v, proj)
            c.render_restir(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=30 + k))
            s = c.stats(); rays += np.array([s.rays_primary, s.rays_extension, s.rays_shadow], np.uint64)
        assert np.array_equal(rays, total), (wave, chunks, rays, total)
        for name, a, b in zip(("cur_di", "cur_gi", "cur_sd", "last_di", "last_gi", "last_sd"), c.read_pass1_buffers() + c.read_restir_last(), st):
            assert np.array_equal(a, b), (wave, chunks, name, int((a != b).any(1).sum()))
        assert np.array_equal(bits(c.read_accum()), bits(acc_o)), (wave, chunks)
        c.close()
    assert total[2] > 4 * W * H and total[1] > 2 * W * H


def test_octant_sorted_fetch_changes_nothing(rt):
    """RTX_OPT_OCTANT_SORT (measured slower and off by default: docs/REJECTED.md): k_shade notes every survivor's direction octant, the closest-hit kernel of the next
    bounce counting-sorts its sub-queue's entries by it in a prologue and fetches through the permutation.  Hit records still land at the entries' own positions, so the frame is the
    same bits with the option on, off, and switched on a live context (sub-queue merging of the thin late bounces included).  Values 3 (key = the cell of the ray's origin on
    a 256-cell grid over the scene's box, from bounce 2), 2 (all keys zero) and 5 (hashed keys) are the variants the measurement in profiles/r04_octsort_ab.md compares."""
    import hashlib
    sc = rt.Scene.sponza_class(60000, 260)
    W, H = 640, 360
    p = rt.Params(width=W, height=H, spp=4, max_bounces=8, nee_samples=1, rr_start=3, sample_base=1, flags=1, frame_seed=3)

    def run(c):
        c.clear(W, H); c.render(p); st = c.stats()
        return (st.rays_primary, st.rays_extension, st.rays_shadow), hashlib.sha1(c.read_accum().tobytes()).hexdigest()
    a = rt.Context(0); a.upload(sc, W / H)
    ref = run(a)
    assert run(a) == ref                                     # (second frame: launch sizes predicted, thin launches merged)
    b = rt.Context(0); b.set_option(rt.OPT_OCTANT_SORT, 1); b.upload(sc, W / H)
    assert run(b) == ref and run(b) == ref
    b.set_option(rt.OPT_OCTANT_SORT, 0); assert run(b) == ref
    a.set_option(rt.OPT_OCTANT_SORT, 1); assert run(a) == ref
    for mode in (3, 2, 5):
        a.set_option(rt.OPT_OCTANT_SORT, mode); assert run(a) == ref, mode
    a.close(); b.close()


@pytest.mark.parametrize("spp", [1, 4, 6, 16, 24])
def test_sample_interleaved_queue_order_changes_nothing(rt, spp):
    """RTX_OPT_SAMPLE_INTERLEAVE (default on): k_raygen fills a chunk of 256 queue entries with 256 / S pixel slots x S consecutive samples (S = the largest power of two <= 16
    dividing the batch's sample count: 1, 4, 2, 16, 8 here) instead of 256 slots of one sample.  Which path sits where changes no path and no sum: same bits and ray counts as
    the one-sample chunks, whole frame and a 2-of-3 shard, and with a batch cap that splits the frame into batches of another sample count."""
    import hashlib
    sc = rt.Scene.sponza_class(30000, 260)
    W, H = 256, 144

    def run(c, **kw):
        p = rt.Params(width=W, height=H, spp=spp, max_bounces=6, nee_samples=1, rr_start=3, sample_base=2, flags=3, frame_seed=5, **kw)
        c.clear(W, H); c.render(p); st = c.stats()
        return (st.rays_primary, st.rays_extension, st.rays_shadow), hashlib.sha1(c.read_accum().tobytes()).hexdigest()
    c = rt.Context(0); c.upload(sc, W / H)
    on = run(c), run(c, shard_rank=1, shard_count=3, tile_size=16)
    c.set_option(rt.OPT_SAMPLE_INTERLEAVE, 0)
    off = run(c), run(c, shard_rank=1, shard_count=3, tile_size=16)
    assert on == off
    c.set_option(rt.OPT_SAMPLE_INTERLEAVE, 1); c.set_option(rt.OPT_PATHS_PER_BATCH, 5 * W * H)      # batches of 5 (and the rest) samples: S = 1 / 4 / ...
    assert run(c) == off[0]
    c.close()


def test_node_stride_128_changes_nothing_also_after_a_refit(rt, golden_dir):
    """RTX_OPT_NODE_STRIDE 128 (profiles/r04_node_stride_ab.md): the traversal fetches the nodes from a copy with one node per 128-B line; the copy is refreshed after a
    build and after every refit.  Same hit records, image and ray counts as the 80-B stride, before and after moving an instance.  (Auto, the default, makes the copy for
    trees above 16 MB and hands it to the path tracer's closest-hit launches of bounces >= 1: the street scene of test_full_size_baseline_configs_are_bit_identical.)"""
    import os
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 160, 96
    p = rt.Params(width=W, height=H, spp=4, max_bounces=5, nee_samples=1, flags=3, frame_seed=11)
    rays = random_rays(30000, 77, -4, 4)
    m = np.eye(4, dtype=np.float32); m[0, 0] = np.cos(1.9); m[0, 2] = -np.sin(1.9); m[2, 0] = np.sin(1.9); m[2, 2] = np.cos(1.9); m[3, 1] = 0.1
    out = []
    for stride in (80, 128):
        c = rt.Context(0); c.set_option(rt.OPT_NODE_STRIDE, stride); c.upload(sc, W / H)
        got = []
        for step in range(2):
            if step: c.set_instance_transform(1, m.reshape(16)); c.commit(); assert c.stats().bvh_refits == 1
            c.clear(W, H); c.render(p); st = c.stats()
            got.append((bits(c.trace_closest(rays)).tobytes(), c.trace_any(rays).tobytes(), c.read_accum().tobytes(), (st.rays_primary, st.rays_extension, st.rays_shadow)))
        assert got[0][2] != got[1][2]
        out.append(got); c.close()
    assert out[0] == out[1]


def test_staged_node_count_per_kind_of_launch_changes_nothing(rt):
    """RTX_OPT_LDS_NODES_CLOSEST: the path tracer's closest-hit launches may stage another number of BVH nodes in LDS than the shadow launches (DevScene goes by value per
    launch); switched on a live context, the frame is the same bits for 0 (as the shadow launches), 9, 40, 73 and the automatic choice."""
    import hashlib
    sc = rt.Scene.sponza_class(60000, 260)
    W, H = 320, 180
    p = rt.Params(width=W, height=H, spp=4, max_bounces=6, nee_samples=1, flags=1, frame_seed=4)
    c = rt.Context(0); c.upload(sc, W / H)
    out = []
    for n in (-1, 0, 9, 40, 73, -1):
        c.set_option(rt.OPT_LDS_NODES_CLOSEST, n)
        c.clear(W, H); c.render(p); st = c.stats()
        out.append(((st.rays_primary, st.rays_extension, st.rays_shadow), hashlib.sha1(c.read_accum().tobytes()).hexdigest()))
    assert all(o == out[0] for o in out)
    c.close()


def test_trace_counters_report_work_per_ray_and_change_nothing(rt):
    """RTX_OPT_TRACE_COUNTERS (bench.py: extra.*.work_per_ray): the persistent traversal kernels tally node steps and triangle tests; same image and ray counts with the
    counters on, and per closest-hit ray the tally lies within a few per cent of the one-ray-per-thread statistics kernel's (rtx_debug_trace_stats on the frame's own primary
    rays: the speculative schedule takes a few node steps more, because pending triangles have not yet shortened the ray)"""
    sc = rt.Scene.sponza_class(40000, 260)
    W, H = 160, 90
    c = rt.Context(0); c.upload(sc, W / H)
    p = rt.Params(width=W, height=H, spp=2, max_bounces=1, nee_samples=1, flags=1)
    c.clear(W, H); c.render(p); ref, s0 = c.read_accum(), c.stats()
    c.set_option(rt.OPT_TRACE_COUNTERS, 1)
    c.clear(W, H); c.render(p); s1 = c.stats()
    cn, ct, an, at = c.trace_counters()
    assert np.array_equal(bits(c.read_accum()), bits(ref))
    assert (s1.rays_primary, s1.rays_extension, s1.rays_shadow) == (s0.rays_primary, s0.rays_extension, s0.rays_shadow)
    assert s1.rays_extension == 0 and s1.rays_primary == W * H * 2                 # max_bounces 1: the closest-hit rays ARE the camera rays (two samples, no jitter: the same rays twice)
    st = c.trace_stats(c.primary_rays(p))
    steps_ref, tris_ref = float(st[:, 1].sum()) * 2, float(st[:, 2].sum()) * 2
    assert steps_ref * 0.999 <= cn <= steps_ref * 1.15, (cn, steps_ref)
    assert tris_ref * 0.999 <= ct <= tris_ref * 1.15, (ct, tris_ref)
    assert s1.rays_shadow > 0 and 2.0 < an / s1.rays_shadow < 40.0 and at > 0
    assert c.trace_counters() == (0, 0, 0, 0)                                      # reading resets
    c.close()


@pytest.mark.parametrize("lanes", [1, 2, 3, 4])
def test_restir_pipeline_lanes_equal_the_literal_form(rt, orc, golden_dir, lanes):
    """RTX_OPT_RESTIR_LANES: the pixel list of a ReSTIR frame as 1 .. 4 independent parts on as many streams (passes 1 + 2, join, pass 3).  The lanes only engage for lists
    of >= 65 536 pixels, so RTX_OPT_RESTIR_LANE_MIN is lowered here: three frames with a moving camera on garage.obj + monke.obj (a BVH scene: Morton-ordered list),
    unsharded and on two BLOCK_TILES shards with their halo lists — image, the three history buffers and the ray counts equal the LITERAL form's (one thread per pixel
    and pass, no lists, no streams), which equals the oracle's (test_restir_garage_with_camera_motion)."""
    import torch
    from royaltracer_dx_amd import sharding
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H, TS = 160, 96, 32
    cams = [rt.lookat((-1.5 + 0.05 * k, 1.5, 3.5 - 0.04 * k), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)) for k in range(3)]
    proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, tile_size=TS)
    ref = rt.Context(0); ref.set_option(rt.OPT_RESTIR_WAVEFRONT, 0); ref.upload(sc, W / H); ref.restir_reset(); ref.clear(W, H)
    ref_counts = []
    for k, v in enumerate(cams):
        ref.set_camera(v, proj); ref.render_restir(rt.Params(frame_seed=70 + k, flags=0, **base))
        s = ref.stats(); ref_counts.append((s.rays_primary, s.rays_extension, s.rays_shadow))
    ref_img, ref_last = ref.read_accum(), ref.read_restir_last()
    ref.close()

    def ctx():
        c = rt.Context(0); c.set_option(rt.OPT_RESTIR_LANES, lanes); c.set_option(rt.OPT_RESTIR_LANE_MIN, 256)
        c.set_option(rt.OPT_RESTIR_KEYS, lanes & 1)              # lanes 1 / 3: the selection stage of pass 3 on the compact neighbour records (the default); 2 / 4: on the full records
        c.upload(sc, W / H); c.restir_reset(); c.clear(W, H)
        return c
    c = ctx()
    for k, v in enumerate(cams):
        c.set_camera(v, proj); c.render_restir(rt.Params(frame_seed=70 + k, flags=0, **base))
        s = c.stats(); assert (s.rays_primary, s.rays_extension, s.rays_shadow) == ref_counts[k], k
    assert np.array_equal(bits(c.read_accum()), bits(ref_img))
    for a, b in zip(c.read_restir_last(), ref_last):
        assert np.array_equal(a, b)
    c.close()
    ranks = [ctx(), ctx()]
    for k, v in enumerate(cams):
        slabs = []
        for r, cr in enumerate(ranks):
            p = rt.Params(frame_seed=70 + k, flags=rt.FLAG_BLOCK_TILES, shard_rank=r, shard_count=2, **base)
            cr.set_camera(v, proj); cr.render_restir(p)
            slab = torch.empty(cr.restir_state_slab_bytes(p) // 4, dtype=torch.float32, device="cuda:0")
            cr.restir_pack_state(p, slab.data_ptr()); slabs.append(slab)
        torch.cuda.synchronize()
        gathered = torch.cat(slabs); torch.cuda.synchronize()
        for r, cr in enumerate(ranks):
            cr.restir_unpack_state(rt.Params(frame_seed=70 + k, flags=rt.FLAG_BLOCK_TILES, shard_rank=r, shard_count=2, **base), gathered.data_ptr())
    own = sharding.owner_map(W, H, TS, 2, True)
    for r, cr in enumerate(ranks):
        assert np.array_equal(bits(cr.read_accum()[own == r]), bits(ref_img[own == r])), r
        for a, b in zip(cr.read_restir_last(), ref_last):
            assert np.array_equal(a, b), r
        cr.close()


@pytest.mark.parametrize("split", [0, 10000], ids=["object_splits", "spatial_splits"])
@pytest.mark.parametrize("gpu_refit", [1, 0])
def test_animated_instance_refit_parity(rt, orc, golden_dir, gpu_refit, split):
    """rtx_set_instance_transform + rtx_commit_scene refits the BVH (reference: TLAS refit every frame); the images of
    the bounce-loop tracer and of the ReSTIR pipeline (which reprojects through prevObjectToWorld) stay identical to the
    oracle, which rebuilds its own BVH from scratch.  gpu_refit=1: k_refit_tris / k_refit_nodes re-derive the world triangles
    and re-quantise the resident wide nodes; gpu_refit=0: host refit + re-collapse + upload.  spatial_splits: the tree was built with RTX_OPT_BVH_SPLIT, so
    triangles are referenced from several leaves; a refit gives every reference the bounds of its whole (moved) triangle"""
    import os
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 80, 48
    c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, gpu_refit); c.set_option(rt.OPT_BVH_SPLIT, split); c.upload(sc, W / H); c.set_camera(*sc.view_proj(W / H))
    assert c.stats().triangles == sc.num_triangles and (c.stats().bvh_refs > sc.num_triangles) == (split > 0)
    o = orc.Oracle().load(sc, W / H); o.set_camera(*sc.view_proj(W / H))
    acc_o, st = np.zeros((H, W, 4), np.float32), None
    c.restir_reset(); c.clear(W, H)
    for k, ang in enumerate([1.57, 1.75, 2.0]):
        m = np.eye(4, dtype=np.float32); m[0, 0] = np.cos(ang); m[0, 2] = -np.sin(ang); m[2, 0] = np.sin(ang); m[2, 2] = np.cos(ang); m[3, 1] = 0.05 * k
        c.set_instance_transform(1, m.reshape(16)); c.commit()
        o.set_instance_transform(1, m.reshape(16))
        assert c.stats().bvh_refits == k + 1
        assert c.validate_bvh() == 0                      # coverage of the resident (refitted) tree, checked on the host
        rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H)), random_rays(20000, 40 + k, -4, 4)])
        g, b = c.trace_closest(rays), o.trace_closest(rays, 0)
        assert np.array_equal(bits(g)[:, 3], bits(b)[:, 3])
        hit = bits(b)[:, 3] != 0xFFFFFFFF
        assert np.array_equal(bits(g)[hit], bits(b)[hit])
        p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=50 + k)
        c.render_restir(p)
        acc_o, st, cnt = o.restir_frames(p, acc_o, st)
        s = c.stats(); assert (s.rays_primary, s.rays_extension, s.rays_shadow) == cnt
    ld, lg, ls = c.read_restir_last()
    assert np.array_equal(ld, st[3]) and np.array_equal(lg, st[4]) and np.array_equal(ls, st[5])
    assert np.array_equal(bits(c.read_accum()), bits(acc_o))
    pt = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1, flags=0)
    c.clear(W, H); c.render(pt)
    ref, _ = o.render(pt)
    assert rel_l2(c.read_accum()[..., :3], ref[..., :3]) <= REL_L2_TOL
    c.close()


class _EditedScene:
    """a Scene with another material table (what rtx_set_materials + rtx_commit_scene on a resident scene must equal)"""
    def __init__(self, base, materials):
        self.materials, self.meshes, self.instances, self._b = materials, base.meshes, base.instances, base

    def view_proj(self, aspect):
        return self._b.view_proj(aspect)


@pytest.mark.parametrize("gpu_refit", [1, 0])
@pytest.mark.parametrize("tris", [40000, 28])
def test_material_edit_after_commit(rt, orc, cornell, gpu_refit, tris):
    """rtx_set_materials on a scene that is already resident, then rtx_commit_scene: the material table, its count and the emissive
    list must all follow, on the GPU-refit commit path (non-tiny scene, RTX_OPT_GPU_REFIT=1), the host-refit path (=0) and the tiny-scene
    path alike — compared with an oracle loaded from scratch with the edited table (image bits, ray counts, light records)."""
    sc = rt.Scene.sponza_class(tris, 260) if tris > 64 else cornell
    W, H = 96, 54
    p = rt.Params(width=W, height=H, spp=2, max_bounces=4, nee_samples=1, flags=0, frame_seed=9)
    c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, gpu_refit); c.upload(sc, W / H)
    c.clear(W, H); c.render(p); before = c.read_accum(); n0 = c.stats().materials; l0 = c.stats().lights
    mats = np.array(sc.materials, dtype=np.float32, copy=True)
    used = np.unique(np.concatenate([m for _, _, m in sc.meshes]))
    dark = [int(i) for i in used if mats[i, 8:11].sum() == 0.0]
    assert len(dark) >= 2
    mats[dark[0], 0:3] = (0.9, 0.1, 0.2)                       # a new Kd
    mats[dark[1], 8:11] = (3.0, 2.0, 1.0)                      # a surface material becomes a light: the CDF grows, its hits now end paths
    mats = np.concatenate([mats, mats[:1]])                    # and the table grows by one (unused) entry
    c.set_materials(mats); c.commit()
    st = c.stats()
    assert st.materials == n0 + 1 and st.lights > l0
    c.clear(W, H); c.render(p); after = c.read_accum(); st = c.stats()
    o = orc.Oracle().load(_EditedScene(sc, mats), W / H)
    oa, oc = o.render(p)
    assert np.array_equal(bits(c.lights()), bits(o.lights()))
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc
    assert np.array_equal(bits(after), bits(oa)) and not np.array_equal(bits(after), bits(before))
    c.close()


@pytest.mark.parametrize("kind", ["cornell", "garage", "atrium", "garage_split"])
def test_scene_cache_renders_bit_identically(rt, orc, cornell, golden_dir, tmp_path, kind):
    """a scene loaded from the binary cache (prebuilt BVH, shading records, LUTs, light CDF: no build at commit) renders the
    same bits and traces the same rays as the scene committed from scratch — through the context-level pair rtx_save_scene_cache /
    rtx_load_scene_cache and through the host-level rtxh_scene_save / rtxh_scene_load; the loaded scene can still be edited (a
    transform-only commit refits on the GPU from object-space triangles re-derived at that point, a material edit re-derives the table).
    garage_split: the tree was built with spatial splits (RTX_OPT_BVH_SPLIT): the file then holds more leaf entries than triangles."""
    split = kind == "garage_split"
    sc = {"cornell": lambda: cornell, "atrium": lambda: rt.Scene.sponza_class(60000, 260),
          "garage": lambda: rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")}[kind.split("_")[0]]()
    W, H = 128, 72
    p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1, flags=0, frame_seed=3)
    a = rt.Context(0)
    if split:
        a.set_option(rt.OPT_BVH_SPLIT, 10000)
    a.upload(sc, W / H); a.clear(W, H); a.render(p); ref = a.read_accum(); sa = a.stats()
    assert (sa.bvh_refs > sa.triangles) == split
    a.save_scene_cache(tmp_path / "ctx.rtxscn")
    sc.save(tmp_path / "host.rtxscn")
    for how in (("ctx",) if split else ("ctx", "host")):            # (the host-level file is built with the process-wide builder defaults)
        b = rt.Context(0)
        if how == "ctx":
            b.load_scene_cache(tmp_path / "ctx.rtxscn"); b.set_camera(*sc.view_proj(W / H))
        else:
            b.upload(rt.Scene.load(tmp_path / "host.rtxscn"), W / H)
        b.clear(W, H); b.render(p); sb = b.stats()
        assert np.array_equal(bits(b.read_accum()), bits(ref)), (kind, how)
        assert (sb.rays_primary, sb.rays_extension, sb.rays_shadow, sb.triangles, sb.bvh_refs, sb.bvh_nodes, sb.lights, sb.materials) == \
               (sa.rays_primary, sa.rays_extension, sa.rays_shadow, sa.triangles, sa.bvh_refs, sa.bvh_nodes, sa.lights, sa.materials)
        assert np.array_equal(bits(b.lights()), bits(a.lights()))
        if kind != "cornell":
            assert b.validate_bvh() == 0
        # the loaded scene stays editable: move an instance in both contexts, commit, compare again
        M = np.eye(4, dtype=np.float32); M[3, 0] = 0.05; M[0, 0] = 1.1
        for c in (a, b):
            c.set_instance_transform(0, M.reshape(16)); c.commit(); c.clear(W, H); c.render(p)
        assert np.array_equal(bits(a.read_accum()), bits(b.read_accum())), (kind, how, "after a transform edit")
        I = np.eye(4, dtype=np.float32).reshape(16)
        a.set_instance_transform(0, I); a.commit()
        b.close()
    bad = tmp_path / "bad.rtxscn"; blob = bytearray(open(tmp_path / "ctx.rtxscn", "rb").read()); blob[len(blob) // 3] ^= 1; bad.write_bytes(bytes(blob))
    a.clear(W, H); a.render(p); again = a.read_accum()
    with pytest.raises(rt.RtxError):
        a.load_scene_cache(bad)                                 # refused ...
    a.clear(W, H); a.render(p)
    assert np.array_equal(bits(a.read_accum()), bits(again))   # ... and the resident scene is untouched
    a.close()


def test_scene_cache_loads_a_bistro_class_scene_faster_than_it_builds(rt, tmp_path):
    """3.8 M triangles: commit from the cache (read + checksum + upload) against the commit that builds the BVH, and the same image"""
    import time
    sc = rt.Scene.bistro_class()
    W, H = 160, 90
    p = rt.Params(width=W, height=H, spp=1, max_bounces=4, nee_samples=1, flags=0)
    a = rt.Context(0)
    t0 = time.perf_counter(); a.upload(sc, W / H); t_build = time.perf_counter() - t0
    a.clear(W, H); a.render(p); ref = a.read_accum()
    path = tmp_path / "bistro.rtxscn"
    t0 = time.perf_counter(); a.save_scene_cache(path); t_save = time.perf_counter() - t0
    a.close()
    b = rt.Context(0)
    t0 = time.perf_counter(); b.load_scene_cache(path); t_load = time.perf_counter() - t0
    b.set_camera(*sc.view_proj(W / H)); b.clear(W, H); b.render(p)
    print(f"bistro-class {sc.num_triangles} triangles: upload + commit with build {t_build:.2f} s, cache save {t_save:.2f} s ({os.path.getsize(path) / 1e6:.0f} MB), cache load + upload {t_load:.2f} s")
    assert np.array_equal(bits(b.read_accum()), bits(ref))
    assert t_load < 0.5 * t_build
    b.close()


def test_tile_size_contract_is_one_rule(rt, cornell):
    """tile_size: a power of two in [16, 1024] (0 = 64) — the same verdict from rtx_shard_slab_bytes, rtx_render, rtx_pack_tiles and the
    host-side layout; slab sizes are computed in 64 bits"""
    from royaltracer_dx_amd import sharding
    c = rt.Context(0); c.upload(cornell, 1.0)
    for ts in (8, 24, 48, 2048, 1000):
        p = rt.Params(width=64, height=64, spp=1, tile_size=ts)
        with pytest.raises(rt.RtxError):
            c.slab_bytes(p)
        with pytest.raises(rt.RtxError):
            c.render(p)
        with pytest.raises(ValueError):
            sharding.layout(64, 64, ts, 1)
    for ts in (0, 16, 64, 1024):
        p = rt.Params(width=200, height=120, spp=1, tile_size=ts, shard_rank=1, shard_count=3)
        assert c.slab_bytes(p) == sharding.layout(200, 120, ts, 3)["npl"] * 16
    with pytest.raises(rt.RtxError):
        c.slab_bytes(rt.Params(width=0xFFFFFFFF, height=0xFFFFFFFF, tile_size=16))      # 7.2e16 tile pixels: refused, not wrapped
    c.close()


def test_gpu_refit_large_scene_stays_conservative(rt, orc):
    """GPU refit of a 262 k-triangle tree under a large rigid motion + non-uniform scale: the refitted tree validates on the host,
    hit records and the image equal the oracle's (which rebuilds its own BVH), and a second refit back to identity reproduces
    the first image"""
    sc = rt.Scene.sponza_class(262144, 260)
    W, H = 96, 54
    p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=1, flags=1)
    c = rt.Context(0); c.upload(sc, W / H)
    o = orc.Oracle().load(sc, W / H)
    assert c.validate_bvh() == 0
    c.clear(W, H); c.render(p); first = c.read_accum()
    th = 0.7
    M = np.array([[1.3 * np.cos(th), 0, np.sin(th), 0], [0, 0.8, 0, 0], [-1.3 * np.sin(th), 0, np.cos(th), 0], [0.25, -0.1, 0.4, 1]], np.float32).reshape(16)   # column-major
    c.set_instance_transform(0, M); c.commit(); o.set_instance_transform(0, M)
    assert c.stats().bvh_refits == 1 and c.validate_bvh() == 0
    rays = np.concatenate([o.primary_rays(p), random_rays(30000, 41, -1.5, 1.5)])
    g, b = c.trace_closest(rays), o.trace_closest(rays, 1)
    assert np.array_equal(bits(g), bits(b))
    c.clear(W, H); c.render(p); ca, cnt = o.render(p)
    assert np.array_equal(bits(c.read_accum()), bits(ca))
    I = np.eye(4, dtype=np.float32).reshape(16)
    c.set_instance_transform(0, I); c.commit()
    assert c.stats().bvh_refits == 2 and c.validate_bvh() == 0
    c.clear(W, H); c.render(p)
    assert np.array_equal(bits(c.read_accum()), bits(first))
    c.close()


class _TwoScenes:
    """a large scene and a small one as ONE scene: the small one's meshes, materials and instances appended (duck-typed like rt.Scene for Context.upload / Oracle.load)"""
    def __init__(self, big, small, place):
        nm = len(big.materials)
        self.materials = np.concatenate([np.asarray(big.materials, np.float32), np.asarray(small.materials, np.float32)])
        self.meshes = list(big.meshes)
        base = sum(len(m) for _, _, m in big.meshes)
        for v, i, m in small.meshes:                             # Vertex.normal.w carries the mesh's base offset into the scene's materialIDs
            v = np.array(v, np.float32, copy=True).reshape(-1, 7); v[:, 6] = float(base)
            self.meshes.append((v, i, np.asarray(m, np.uint32) + np.uint32(nm))); base += len(m)
        self.instances = list(big.instances) + [(len(big.meshes) + mesh, place) for mesh, _ in small.instances]
        self._big = big
    def view_proj(self, aspect):
        return self._big.view_proj(aspect)


@pytest.mark.parametrize("partial", [1, 0])
def test_partial_refit_of_one_small_instance_in_a_large_scene(rt, orc, golden_dir, partial):
    """RTX_OPT_PARTIAL_REFIT (default): after the first (full) refit a transform-only commit re-derives the triangles of the MOVED instances only and re-quantises only the
    nodes above them (clean children keep the float boxes the previous refit left in node_aabb).  A 60 k-triangle atrium with monke.obj flying through it: after every move
    the resident tree validates on the host and the hit records of camera + random rays equal the oracle's (which rebuilds); then the atrium itself moves while monke
    stays, and finally both — the three dirty patterns.  partial=0: every refit touches the whole tree."""
    import os
    big = rt.Scene.sponza_class(60000, 260)
    small = rt.Scene.from_obj([os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    def at(x, y, z, ang, s=0.25):
        m = np.eye(4, dtype=np.float32); m[0, 0] = s * np.cos(ang); m[0, 2] = -s * np.sin(ang); m[2, 0] = s * np.sin(ang); m[2, 2] = s * np.cos(ang); m[1, 1] = s
        m[3, 0], m[3, 1], m[3, 2] = x, y, z
        return m.reshape(16)
    sc = _TwoScenes(big, small, at(0.0, 0.3, 0.0, 0.0))
    W, H = 96, 54
    c = rt.Context(0); c.set_option(rt.OPT_PARTIAL_REFIT, partial); c.upload(sc, W / H)
    o = orc.Oracle().load(sc, W / H)
    mk = len(sc.instances) - 1
    assert c.validate_bvh() == 0
    moves = [(mk, at(0.2, 0.35, 0.1, 0.4)), (mk, at(-0.3, 0.5, -0.2, 1.1)), (mk, at(0.6, 0.2, 0.3, 2.0, 0.4)),
             (0, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0.05, 0.0, -0.03, 1]], np.float32).reshape(16)), (mk, at(0.0, 0.3, 0.0, 0.0))]
    for k, (inst, M) in enumerate(moves):
        c.set_instance_transform(inst, M); o.set_instance_transform(inst, M)
        if k == 4:                                                  # the last commit moves both instances at once
            I = np.eye(4, dtype=np.float32).reshape(16); c.set_instance_transform(0, I); o.set_instance_transform(0, I)
        c.commit()
        assert c.stats().bvh_refits == k + 1 and c.validate_bvh() == 0, k
        rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H)), random_rays(20000, 90 + k, -1.5, 1.5)])
        g, b = c.trace_closest(rays), o.trace_closest(rays, 1)
        assert np.array_equal(bits(g), bits(b)), k
        assert np.array_equal(c.trace_any(rays), o.trace_any(rays, 1)), k
        assert np.array_equal(bits(c.lights()), bits(o.lights())), k       # the world-space half of the light records is refreshed from the 80-byte records, not re-scanned
    c.close()


def test_c4_sizes_4k_shards_and_stack_variants(rt, orc):
    """C4-shaped run (Sponza-class, 3840x2160, pixel-tile shards) at 1 spp: memory sizing of the queues at 8.3 M pixels,
    shard reassembly at full size, and every traversal-stack placement and wave schedule gives the same image"""
    sc = rt.Scene.sponza_class(262144, 260)
    W, H = 3840, 2160
    base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=1, flags=1)
    imgs = []
    # (stack, schedule): LDS column + speculative voted (default), private stack, while-while, voted, voted with other weights
    # the last entry also turns on the material-sorted k_shade variant
    # entries 8-12 run the separate trace / shade / shadow kernels (RTX_OPT_FUSED_BVH = 0), entry 10 with material-sorted shading,
    # entries 11-12 with work stealing between the sub-queues (RTX_OPT_WORK_STEALING)
    # entries 12-13: path state by path id, in place (RTX_OPT_COMPACT_STATE = 0; the others keep it by queue position, the default)
    for k, (stack, sched) in enumerate(((0, 6), (1, 6), (0, 0), (0, 2), (0, 5), (0, 7), (0, 3), (0, 6), (0, 6), (1, 2), (0, 6), (0, 6), (1, 2), (0, 6))):
        c = rt.Context(0); c.set_option(rt.OPT_STACK_PRIVATE, stack); c.set_option(rt.OPT_TRACE_SCHED, sched)
        c.set_option(rt.OPT_FUSED_BVH, 0 if k >= 8 else 1); c.set_option(rt.OPT_WORK_STEALING, 1 if k in (11, 12) else 0); c.set_option(rt.OPT_COMPACT_STATE, 0 if k >= 12 else 1)
        c.set_option(rt.OPT_OVERLAP_SHADOW, 0 if k in (8, 13) else 1)        # (default 1: shadow rays of bounce b beside the closest-hit rays of bounce b + 1)
        c.set_option(rt.OPT_SORT_MATERIALS, 1 if k in (7, 10) else 0); c.upload(sc, W / H)
        c.clear(W, H); c.render(rt.Params(**base)); imgs.append(c.read_accum())
        if k == 0:
            st = c.stats(); assert st.rays_primary == W * H
            c.clear(W, H)
            for r in range(8):
                c.render(rt.Params(shard_rank=r, shard_count=8, **base))
            assert np.array_equal(bits(c.read_accum()), bits(imgs[0]))
        c.close()
    for im in imgs[1:]:
        assert np.array_equal(bits(imgs[0]), bits(im))
    # a 64 x 36 crop-sized oracle check is done elsewhere; here: plausibility + determinism
    assert np.isfinite(imgs[0]).all() and (imgs[0][..., 3] == 1).all() and imgs[0][..., :3].mean() > 0.01


class RandomTinyScene:
    """<= 64 random triangles in the reference's data model: loose triangles, planar quads (some flush with the scene's bounding box, i.e.
    hull faces), boxes, one to three emissive polygons (sometimes hull faces themselves), Lambert and GGX materials"""
    def __init__(self, rt, seed, max_tris=64):
        rng = np.random.default_rng(seed)
        nm = int(rng.integers(2, 6))
        m = np.zeros((1 + nm, 32), np.float32)
        m[0, 0:4] = (1, 1, 1, 1); m[0, 12] = 1.0
        for i in range(1, 1 + nm):
            m[i, 0:3] = rng.uniform(0.05, 0.9, 3); m[i, 3] = 1.0
            m[i, 4:7] = rng.uniform(0.0, 0.9, 3) * rng.integers(0, 2); m[i, 7] = 1.0
            m[i, 12] = rng.choice([1.0, 0.5, 0.2, 0.03]); m[i, 13] = rng.choice([0.0, 0.0, 1.0])
            m[i, 16:32] = np.clip(rng.uniform(0.6, 1.0, 16), 0.05, 1.0)           # any plausible Ess table
        nl = int(rng.integers(1, 4))
        lights = list(range(1, 1 + min(nl, nm - 1)))
        for i in lights:
            m[i, 0:3] = 0.0; m[i, 8:11] = rng.uniform(0.5, 12.0, 3)
        self.materials = m
        tris, mats = [], []
        def quad(c, a, b, mat):
            p = [c - a - b, c + a - b, c + a + b, c - a + b]
            tris.extend([(p[0], p[1], p[2]), (p[0], p[2], p[3])]); mats.extend([mat, mat])
        R = 1.0
        nonl = [i for i in range(1, 1 + nm) if i not in lights]
        if rng.random() < 0.7:                                  # a room-like shell: some faces of the bounding box (hull faces)
            for ax in range(3):
                for sgn in (-1.0, 1.0):
                    if rng.random() < 0.6:
                        c = np.zeros(3); c[ax] = sgn * R
                        a = np.zeros(3); a[(ax + 1) % 3] = R
                        b = np.zeros(3); b[(ax + 2) % 3] = R
                        quad(c, a, b, int(rng.choice(nonl)))
        for li in lights:                                       # emissive quads: hanging inside, or flush with a box face
            ax = int(rng.integers(0, 3)); c = rng.uniform(-0.5, 0.5, 3); c[ax] = rng.choice([0.999 * R, R, rng.uniform(0.3, 0.9)])
            a = np.zeros(3); a[(ax + 1) % 3] = rng.uniform(0.1, 0.4)
            b = np.zeros(3); b[(ax + 2) % 3] = rng.uniform(0.1, 0.4)
            quad(c, a, b, li)
        target = 58 if max_tris <= 64 else int(rng.integers(65, max_tris))
        while len(tris) < target and (max_tris > 64 or rng.random() < 0.9):
            kind = rng.integers(0, 3)
            c = rng.uniform(-0.8, 0.8, 3)
            if kind == 0:                                       # loose triangle (slivers included)
                tris.append(tuple(c + rng.normal(scale=rng.choice([0.02, 0.2, 0.5]), size=3) for _ in range(3))); mats.append(int(rng.choice(nonl)))
            elif kind == 1:                                     # tilted quad
                a = rng.normal(size=3); a *= rng.uniform(0.05, 0.4) / np.linalg.norm(a)
                b = np.cross(a, rng.normal(size=3)); b *= rng.uniform(0.05, 0.4) / max(np.linalg.norm(b), 1e-9)
                quad(c, a, b, int(rng.choice(nonl)))
            elif len(tris) <= target - 12:                      # axis-aligned box
                h = rng.uniform(0.05, 0.3, 3); mat = int(rng.choice(nonl))
                for ax in range(3):
                    for sgn in (-1.0, 1.0):
                        cc = c.copy(); cc[ax] += sgn * h[ax]
                        a = np.zeros(3); a[(ax + 1) % 3] = h[(ax + 1) % 3]
                        b = np.zeros(3); b[(ax + 2) % 3] = h[(ax + 2) % 3]
                        quad(cc, a, b, mat)
        t = np.array(tris, np.float32).reshape(-1, 3, 3)[:max_tris]
        mats = np.array(mats[:len(t)], np.uint32)
        # some scenes: smooth vertex normals, with zero components here and there (Hit_v6.hlsl:40-46 tests all(n != 0) per component)
        nrm = np.zeros((len(t), 3, 3), np.float32)
        if seed % 4 == 1:
            fl = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]); fl /= np.maximum(np.linalg.norm(fl, axis=1, keepdims=True), 1e-20)
            nrm = fl[:, None, :] + rng.normal(scale=0.3, size=(len(t), 3, 3)).astype(np.float32)
            nrm[rng.random((len(t), 3)) < 0.15] = 0.0                                  # whole normal missing -> flat
            z = rng.random((len(t), 3, 3)) < 0.05; nrm[z] = 0.0                        # single zero component -> flat for that vertex too
        # some scenes: the triangles split over two meshes / several instances with affine transforms (mirrors, non-uniform scale);
        # the world-space geometry stays the one generated above (vertices are pulled back through the inverse transform)
        groups = [np.arange(len(t))]
        if seed % 5 == 2 and len(t) >= 6:
            cut = int(rng.integers(2, len(t) - 2)); groups = [np.arange(0, cut), np.arange(cut, len(t))]
        self.meshes, self.instances = [], []
        matid_base = 0                                            # Vertex.normal.w = base of the model inside the global materialIDs[] (ObjLoader.h:466)
        for gi, idx in enumerate(groups):
            M = np.eye(4)
            if len(groups) > 1:
                A = rng.normal(size=(3, 3)) * 0.4 + np.diag(rng.choice([-1.0, 1.0], 3) * rng.uniform(0.5, 1.5, 3))
                M[:3, :3] = A; M[:3, 3] = rng.uniform(-0.3, 0.3, 3)
            Mi = np.linalg.inv(M)
            tv = t[idx].reshape(-1, 3).astype(np.float64)
            local = (tv @ Mi[:3, :3].T + Mi[:3, 3]).astype(np.float32)
            v = np.zeros((len(idx) * 3, 7), np.float32); v[:, 0:3] = local; v[:, 3:6] = nrm[idx].reshape(-1, 3); v[:, 6] = matid_base
            matid_base += len(idx) * 3
            self.meshes.append((v, np.arange(len(idx) * 3, dtype=np.uint32), np.repeat(mats[idx], 3)))
            self.instances.append((gi, np.ascontiguousarray(M.T, np.float32).reshape(16)))       # column-major
        eye = rng.uniform(-0.9, 0.9, 3); eye[int(rng.integers(0, 3))] = rng.choice([-2.5, 2.5, 0.0])
        self._v = rt.lookat(tuple(eye), tuple(rng.uniform(-0.3, 0.3, 3)), (0.0, 1.0, 0.0) if abs(eye[1]) < 2 else (0.0, 0.0, 1.0))
        self._rt = rt

    def view_proj(self, aspect):
        return self._v, self._rt.perspective_fov_rh(np.radians(60.0), aspect, 0.1, 1000.0)


def test_random_tiny_scenes_fused_path_equals_oracle(rt, orc):
    """fuzz of the tiny-scene kernels (plane-form pre-test, merged quads, hull-face shortcut, nearest-first exact tests, shadow-ray
    compaction, persistent bounce loop) against the brute-force definition on random geometry, lights and cameras"""
    W, H = 64, 40
    bad = []
    general_too = os.environ.get("RTX_FUZZ_GENERAL", "0") == "1"
    for seed in range(int(os.environ.get("RTX_FUZZ_SCENES", "60"))):
        sc = RandomTinyScene(rt, 1000 + FUZZ_SEED + seed)
        flags = seed & 1
        nsh = 1 + (seed % 11) % 3                                   # every scene renders ONE shard of a 1 / 2 / 3-way tiling with 64 / 32 / 16-pixel tiles
        p = rt.Params(width=W, height=H, spp=3, max_bounces=5 + 2 * (seed % 7 == 3), nee_samples=[1, 1, 2, 4][seed % 4] if seed % 3 == 0 else 1, flags=flags | (2 if seed % 6 == 4 else 0),
                      frame_seed=seed, rr_start=3 if seed % 5 else 1, sample_base=1 + seed % 4, tile_size=[64, 32, 16][seed % 3], shard_rank=seed % nsh, shard_count=nsh)
        o = orc.Oracle().load(sc, W / H)
        oa, oc = o.render(p)
        rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H), 1), random_rays(4000, seed, -1.2, 1.2)])
        ob = o.trace_closest(rays, 0)
        for small in ((1, 0) if (seed % 3 == 0 or general_too) else (1,)):     # every third scene also through the general BVH kernels
            c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.upload(sc, W / H)
            assert c.stats().triangles <= 64
            c.clear(W, H); c.render(p); st = c.stats(); im = c.read_accum()
            ok = np.array_equal(bits(im), bits(oa)) and (st.rays_primary, st.rays_extension, st.rays_shadow) == oc \
                and np.array_equal(bits(c.trace_closest(rays)), bits(ob))
            c.close()
            if not ok:
                bad.append((seed, small))
    assert not bad, f"scenes that differ from the oracle: {bad}"


def test_random_midsize_scenes_general_path_and_refit_equal_oracle(rt, orc):
    """fuzz of the general BVH kernels on random scenes of 65 ... 3000 triangles: image, ray counts, closest-hit queries, then a
    transform-only commit of one instance (mirroring, non-uniform scale) through the GPU refit and the host refit; the oracle rebuilds.
    (600 scenes pass with RTX_FUZZ_SCENES=600.  Round 4: RTX_FUZZ_SEED=200000 RTX_FUZZ_SCENES=1500 over all five random tests reported ONE mismatch — seed 817, host-refit
    context — in the first of three identical runs and none in the other two nor in isolation (tools/repro_fuzz.py); the loop now records WHICH comparison differs.)"""
    W, H = 48, 32
    bad = []
    for seed in range(int(os.environ.get("RTX_FUZZ_SCENES", "12"))):
        sc = RandomTinyScene(rt, 9000 + FUZZ_SEED + seed, max_tris=[200, 800, 3000][seed % 3])
        p = rt.Params(width=W, height=H, spp=2, max_bounces=5, nee_samples=1 + seed % 2, flags=seed & 1, frame_seed=seed)
        o = orc.Oracle().load(sc, W / H)
        oa, oc = o.render(p)
        rays = np.concatenate([o.primary_rays(rt.Params(width=W, height=H), 1), random_rays(3000, seed, -1.2, 1.2)])
        ob = o.trace_closest(rays, 1)
        M = np.eye(4); M[:3, :3] = np.diag([1.1, 0.9, -1.05]) @ np.array([[np.cos(.3), 0, np.sin(.3)], [0, 1, 0], [-np.sin(.3), 0, np.cos(.3)]]); M[:3, 3] = (0.05, -0.02, 0.03)
        inst = len(sc.instances) - 1
        M2 = (M @ np.asarray(sc.instances[inst][1], np.float64).reshape(4, 4).T).T.astype(np.float32).reshape(16)
        o2 = orc.Oracle().load(sc, W / H); o2.set_instance_transform(inst, M2)
        oa2, oc2 = o2.render(p)
        for refit in (1, 0):
            c = rt.Context(0); c.set_option(rt.OPT_GPU_REFIT, refit); c.upload(sc, W / H)
            assert c.stats().triangles > 64
            c.clear(W, H); c.render(p); st = c.stats()
            why = []
            if not np.array_equal(bits(c.read_accum()), bits(oa)): why.append("image")
            if (st.rays_primary, st.rays_extension, st.rays_shadow) != oc: why.append(f"counts {(st.rays_primary, st.rays_extension, st.rays_shadow)} != {oc}")
            if not np.array_equal(bits(c.trace_closest(rays)), bits(ob)): why.append("closest")
            c.set_instance_transform(inst, M2); c.commit()
            c.clear(W, H); c.render(p); st2 = c.stats()
            if not np.array_equal(bits(c.read_accum()), bits(oa2)): why.append("image after the move")
            if (st2.rays_primary, st2.rays_extension, st2.rays_shadow) != oc2: why.append(f"counts after the move {(st2.rays_primary, st2.rays_extension, st2.rays_shadow)} != {oc2}")
            if c.validate_bvh() != 0: why.append("validate")
            c.close()
            if why:
                bad.append((seed, refit, why))
    assert not bad, f"scenes that differ from the oracle: {bad}"


def test_random_tiny_scenes_restir_frames_equal_oracle(rt, orc):
    """fuzz of the reference's own pipeline (pass 1 + temporal + spatial reuse, two frames) on random scenes, through both traversal paths:
    image, the three history buffers and the ray counts.  (The counts matter: paths of zero weight leave no trace in the buffers — a
    miscompiled loop-carried `outgoing` in pass 1 showed up ONLY as an extension-ray count off by one in 3 % of the scenes.)"""
    W, H = 48, 32
    bad = []
    for seed in range(int(os.environ.get("RTX_FUZZ_SCENES", "30"))):
        sc = RandomTinyScene(rt, 5000 + FUZZ_SEED + seed)
        o = orc.Oracle().load(sc, W / H)
        vp = sc.view_proj(W / H)
        p = rt.Params(width=W, height=H, spp=2, max_bounces=3, nee_samples=2 + seed % 3, flags=seed & 1, frame_seed=seed)
        o.set_camera(*vp); o.set_camera(*vp)
        oimg, st, cnt = o.restir_frames(p)
        for small, wave in ((1, 1), (0, 1), (1, 0), (0, 0)):                   # tiny-scene / BVH traversal x wavefront stages / one thread per pixel
            c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, W / H)
            c.set_camera(*vp); c.set_camera(*vp)
            c.restir_reset(); c.clear(W, H); c.render_restir(p)
            gimg = c.read_accum(); ld, lg, ls = c.read_restir_last(); s = c.stats()
            ok = (s.rays_primary, s.rays_extension, s.rays_shadow) == cnt and np.array_equal(ld, st[3]) and np.array_equal(lg, st[4]) \
                and np.array_equal(ls, st[5]) and np.array_equal(bits(gimg), bits(oimg))
            c.close()
            if not ok:
                bad.append((seed, small, wave))
    assert not bad, f"scenes that differ from the oracle: {bad}"


class _RawScene:
    """hand-built scene in the reference's data model: tris (n,3,3) world space, one material id per triangle"""
    def __init__(self, rt, tris, mat_of_tri, materials, eye=(0.2, 0.3, 2.5)):
        tris = np.asarray(tris, np.float32).reshape(-1, 3, 3)
        self.materials = np.asarray(materials, np.float32).reshape(-1, 32)
        if len(tris):
            v = np.zeros((len(tris) * 3, 7), np.float32); v[:, 0:3] = tris.reshape(-1, 3)
            self.meshes = [(v, np.arange(len(tris) * 3, dtype=np.uint32), np.repeat(np.asarray(mat_of_tri, np.uint32), 3))]
            self.instances = [(0, np.eye(4, dtype=np.float32).reshape(16))]
        else:
            self.meshes, self.instances = [], []
        self._v = rt.lookat(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); self._rt = rt

    def view_proj(self, aspect):
        return self._v, self._rt.perspective_fov_rh(np.radians(60.0), aspect, 0.1, 1000.0)


def test_degenerate_scenes_gpu_equals_oracle(rt, orc):
    """the edge cases of the scene model through the C-ABI: no geometry at all, geometry without lights, a lone emissive triangle, only
    zero-area triangles, a zero-area light — image and ray counts as the oracle's, on both kernel paths; a material id beyond the table is refused"""
    def mat(kd=(0.6, 0.5, 0.4), ke=(0, 0, 0)):
        m = np.zeros(32, np.float32); m[0:3] = kd; m[3] = 1; m[8:11] = ke; m[12] = 1; m[16:32] = 0.9
        return m
    quad = lambda y, s: [[(-s, y, -s), (-s, y, s), (s, y, s)], [(-s, y, -s), (s, y, s), (s, y, -s)]]
    floor, lamp = quad(-0.5, 1.5), quad(0.9, 0.3)
    point = [[(0.1, 0.2, 0.3)] * 3]                                    # zero-area triangle
    line = [[(0, 0, 0), (1, 1, 1), (2, 2, 2)]]                         # collinear vertices
    M = [mat(), mat(), mat(kd=(0, 0, 0), ke=(6, 5, 4))]
    scenes = {
        "empty": _RawScene(rt, [], [], M),
        "no lights": _RawScene(rt, floor, [1, 1], M),
        "lone light": _RawScene(rt, lamp[:1], [2], M),
        "only degenerate": _RawScene(rt, point + line + point, [1, 1, 2], M),
        "degenerate among real": _RawScene(rt, floor + lamp + point + line, [1, 1, 2, 2, 2, 1], M),          # incl. a zero-area LIGHT
    }
    # a material id beyond the table: the reference reads out of bounds (zeros); the C-ABI refuses the scene at commit instead
    bad = _RawScene(rt, floor + lamp, [1, 7, 2, 2], M)
    c = rt.Context(0)
    with pytest.raises(rt.RtxError, match="material id out of range"):
        c.upload(bad, 1.5)
    c.close()
    W, H = 48, 32
    for name, sc in scenes.items():
        o = orc.Oracle().load(sc, W / H)
        for flags in (1, 0):
            p = rt.Params(width=W, height=H, spp=3, max_bounces=4, nee_samples=2, flags=flags, frame_seed=9)
            oa, oc = o.render(p)
            for small in (1, 0):
                c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.upload(sc, W / H)
                c.clear(W, H); c.render(p); st = c.stats()
                assert np.array_equal(bits(c.read_accum()), bits(oa)), (name, flags, small)
                assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, (name, flags, small)
                c.close()
        assert oc[0] == W * H * 3


def test_full_size_headline_frame_is_bit_identical(rt, orc, cornell):
    """BASELINE.json configs[1] at its FULL size — Cornell Box, 1920 x 1080, 64 spp, 8 bounces: 132.7 M paths, 552 M rays — rendered by
    the fused tiny-scene kernels and compared with the oracle's frame bit for bit, ray counts included (the oracle needs ~10 s on 16
    threads).  Rare-event coverage the small images cannot give: ~2e8 NEE segments, every room corner, every grazing angle."""
    W, H = 1920, 1080
    p = rt.Params(width=W, height=H, spp=64, max_bounces=8, nee_samples=1, rr_start=3, flags=1, frame_seed=5)
    o = orc.Oracle().load(cornell, W / H); o.set_threads(min(16, os.cpu_count() or 1))
    oa, oc = o.render(p)
    c = rt.Context(0); c.upload(cornell, W / H)
    c.clear(W, H); c.render(p); st = c.stats(); im = c.read_accum()
    for taper in (0, 6):                                    # the default deal is tapered (RTX_OPT_TAPER): equal sub-queues and a steeper taper give the same frame
        c.set_option(rt.OPT_TAPER, taper); c.clear(W, H); c.render(p); s2 = c.stats()
        assert (s2.rays_primary, s2.rays_extension, s2.rays_shadow) == oc, taper
        assert np.array_equal(bits(c.read_accum()), bits(im)), taper
    c.close()
    assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc
    assert oc[0] == W * H * 64 and sum(oc) > 5.4e8
    d = (bits(im) != bits(oa)).any(-1)
    assert not d.any(), f"{int(d.sum())} of {W * H} pixels differ, first at {np.argwhere(d)[0].tolist()}"


@pytest.mark.parametrize("kind", ["garage", "atrium"])
def test_full_size_restir_frames_are_byte_identical(rt, orc, golden_dir, kind):
    """What bench.py's extra.restir_garage_1080p / restir_atrium_1080p time, checked at THEIR size: the reference's shipping frame (pass 1 + temporal + spatial reuse,
    nee 4, bounces 3: Common_v6.hlsl:8-12, Renderer.cpp:646-673) at the reference's hard-wired 1920 x 1080 (Main.cpp:25) on its start-up scene with its camera
    (garage.obj + monke.obj, Renderer.cpp:46-48,363) and on the Sponza-class atrium; three frames with a moving camera, so the temporal pass reprojects.  With the default
    options (wavefront stages, two pipeline lanes, the Morton-ordered pixel list, full-size chunking) and with ONE lane: the image, the three history buffers
    (u3 / u5 / u7) and the three ray counts equal the oracle's byte for byte."""
    import time
    W, H = 1920, 1080
    if kind == "garage":
        sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
        eyes, center = [(-1.5, 1.5, 3.5), (-1.46, 1.5, 3.5), (-1.42, 1.52, 3.47)], (0.0, 1.0, 0.0)                     # Renderer.cpp:46-48, then moving
    else:
        sc = rt.Scene.sponza_class()
        eyes, center = [(-1.8, 0.45, 0.0), (-1.77, 0.45, 0.01), (-1.74, 0.46, 0.02)], (0.5, 0.55, 0.0)
    proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    views = [rt.lookat(e, center, (0.0, 1.0, 0.0)) for e in eyes]
    o = orc.Oracle().load(sc, W / H); o.set_threads(_host_threads())
    acc_o, st, counts = np.zeros((H, W, 4), np.float32), None, []
    t0 = time.time()
    for k, v in enumerate(views):
        o.set_camera(v, proj)
        acc_o, st, cnt = o.restir_frames(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=300 + k), acc_o, st)
        counts.append(cnt)
    print(f"{kind}: oracle {time.time() - t0:.1f} s for 3 frames on {_host_threads()} threads, rays per frame {counts[-1]}")
    for lanes in (2, 1):
        c = rt.Context(0); c.set_option(rt.OPT_RESTIR_LANES, lanes); c.upload(sc, W / H)
        c.restir_reset(); c.clear(W, H)
        for k, v in enumerate(views):
            c.set_camera(v, proj)
            c.render_restir(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=300 + k))
            s = c.stats()
            assert (s.rays_primary, s.rays_extension, s.rays_shadow) == counts[k], (lanes, k)
        ld, lg, ls = c.read_restir_last()
        assert np.array_equal(ld, st[3]) and np.array_equal(lg, st[4]) and np.array_equal(ls, st[5]), lanes
        d = (bits(c.read_accum()) != bits(acc_o)).any(-1)
        assert not d.any(), f"lanes {lanes}: {int(d.sum())} of {W * H} pixels differ, first at {np.argwhere(d)[0].tolist()}"
        c.close()


def _host_threads():
    """threads for the oracle: the CPUs this process may run on (a GPU box hands a share of its cores to each GPU)"""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


@pytest.mark.parametrize("name,kind,W,H,spp,flags,shard", [
    ("C3", "sponza", 1920, 1080, 16, 1, (0, 1)),      # BASELINE.json configs[2]: Sponza-class 262 144 triangles, 1080p, 16 spp, 8 bounces
    ("C5", "bistro", 1920, 1080, 16, 4, (0, 1)),      # configs[4]: Bistro-class 3.8 M triangles, 1080p, 16 spp, dielectric (strategy 3, RTX_FLAG_TRANSMISSION) + GGX microfacet + NEE
    ("C4-shard-5-of-8", "sponza", 3840, 2160, 64, 1, (5, 8)),   # configs[3]: Sponza-class, 4K, 64 spp, 8 bounces: the tiles ONE of the 8 ranks renders
    # round 5: the HARD stand-ins — the same shells, budgets and cameras with the real assets' triangle-size distribution (host/Scenes.h) — at the same sizes
    ("C3-hard", "sponza_hard", 1920, 1080, 16, 1, (0, 1)),
    ("C5-hard", "bistro_hard", 1920, 1080, 16, 4, (0, 1)),
])
def test_full_size_baseline_configs_are_bit_identical(rt, orc, name, kind, W, H, spp, flags, shard):
    """BASELINE.json configs[2..4] at their FULL sizes through the general BVH path, against the oracle on all host cores: every pixel of
    the float accumulation buffer bit for bit, and the primary / extension / shadow ray counts.  C4 is the share of one rank (tile t ->
    rank t mod 8): seeds depend on (x, y, sample, frame_seed) only, so the 8 shards are independent and any one of them is as good a
    witness as the whole frame (the gather itself: test_pack_unpack_kernels_match_host_layout, test_bench_two_ranks_*); the oracle checks one half of that rank's
    tiles (rank 5 of 16), the GPU ties the other half to it."""
    import time
    sc = rt.Scene.sponza_class(hard=kind.endswith("_hard")) if kind.startswith("sponza") else rt.Scene.bistro_class(hard=kind.endswith("_hard"))
    assert abs(sc.num_triangles - (262144 if kind.startswith("sponza") else 3800000)) <= 0.01 * sc.num_triangles
    p = rt.Params(width=W, height=H, spp=spp, max_bounces=8, nee_samples=1, rr_start=3, flags=flags, frame_seed=5,
                  tile_size=64, shard_rank=shard[0], shard_count=shard[1])
    c = rt.Context(0); c.upload(sc, W / H)
    c.clear(W, H); c.render(p); st = c.stats(); im = c.read_accum()
    # the same frame again: now the launch sizes are PREDICTED from the first call's counters, and the thin launches (after Russian roulette) take 2 / 4 sub-queues per
    # workgroup (RTX_OPT_MERGE_RAYS); then with every launch merged as far as one round of workgroups allows: same image, same counts
    for merge_rays in (None, 1 << 20):
        if merge_rays is not None:
            c.set_option(rt.OPT_MERGE_RAYS, merge_rays)
        c.clear(W, H); c.render(p); s2 = c.stats()
        assert (s2.rays_primary, s2.rays_extension, s2.rays_shadow) == (st.rays_primary, st.rays_extension, st.rays_shadow), merge_rays
        assert np.array_equal(bits(c.read_accum()), bits(im)), merge_rays
    # RTX_OPT_TAPER: the first frame above ran with the default (sub-queue sizes tapered 8 | 4 | 2 | 1 over the dispatch order); equal sub-queues and a steeper taper must
    # give the same image and counts
    for taper in (0, 6):
        c.set_option(rt.OPT_TAPER, taper)
        c.clear(W, H); c.render(p); s2 = c.stats()
        assert (s2.rays_primary, s2.rays_extension, s2.rays_shadow) == (st.rays_primary, st.rays_extension, st.rays_shadow), ("taper", taper)
        assert np.array_equal(bits(c.read_accum()), bits(im)), ("taper", taper)
    from royaltracer_dx_amd import sharding
    own = sharding.owner_map(W, H, 64, shard[1]) == shard[0]
    assert not im[~own].any() and (im[own][:, 3] == spp).all()          # only this rank's tiles were touched, every sample landed
    po, sto, imo, halves = p, st, im, []
    if shard[1] > 1:
        # The oracle's share of the suite's time is this config (66 M paths on the host cores).  Tile t belongs to rank t mod 8, so rank r of 8 owns exactly the tiles of ranks
        # r and r + 8 of 16: the GPU renders those two halves as shards of 16 too, their union must be the rank-r-of-8 frame bit for bit (counts add up), and the ORACLE
        # checks one half — every pixel value is independent of how the image is cut (seeds depend on pixel, sample and frame only).
        for r16 in (shard[0], shard[0] + shard[1]):
            q = p.copy(shard_rank=r16, shard_count=2 * shard[1])
            c.clear(W, H); c.render(q); halves.append((q, c.stats(), c.read_accum()))
        assert np.array_equal(bits(halves[0][2] + halves[1][2]), bits(im))                       # disjoint tiles: one addend of every pixel is zero
        assert tuple(a + b for a, b in zip(*[(h[1].rays_primary, h[1].rays_extension, h[1].rays_shadow) for h in halves])) == (st.rays_primary, st.rays_extension, st.rays_shadow)
        po, sto, imo = halves[0]
    c.close()
    o = orc.Oracle().load(sc, W / H); o.set_threads(_host_threads())
    t0 = time.time(); oa, oc = o.render(po); dt = time.time() - t0
    print(f"{name}: {st.triangles} triangles, rays {oc} = {sum(oc) / 1e6:.1f} M, GPU {st.render_ms:.1f} ms, oracle {dt:.1f} s on {_host_threads()} threads")
    assert (sto.rays_primary, sto.rays_extension, sto.rays_shadow) == oc
    own_o = sharding.owner_map(W, H, 64, po.shard_count) == po.shard_rank
    assert oc[0] == int(own_o.sum()) * spp
    d = (bits(imo) != bits(oa)).any(-1)
    assert not d.any(), f"{name}: {int(d.sum())} of {W * H} pixels differ, first at {np.argwhere(d)[0].tolist()}"


def test_analytic_rectangle_light_scene_gpu_equals_oracle(rt, orc):
    """the floor-under-a-rectangular-light scene whose oracle image is pinned against the analytic irradiance
    (test_oracle_golden.py): the GPU must reproduce the oracle's image bit for bit, so the analytic pin carries over"""
    from test_oracle_golden import _FloorAndLight
    sc = _FloorAndLight(); W, H = 24, 16
    sc._v = rt.lookat((2.5, 1.6, 1.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); sc._p = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
    o = orc.Oracle().load(sc, W / H)
    for flags, mb in ((1, 2), (0, 3)):
        p = rt.Params(width=W, height=H, spp=64, max_bounces=mb, nee_samples=1, flags=flags)
        oa, oc = o.render(p)
        for small in (1, 0):                                  # fused tiny-scene kernels and the general BVH path
            c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, small); c.upload(sc, W / H)
            c.clear(W, H); c.render(p); st = c.stats()
            assert np.array_equal(bits(c.read_accum()), bits(oa)), (flags, mb, small)
            assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc
            c.close()


@pytest.mark.parametrize("knob", ["shade_dense", "occluder_cache", "both"])
def test_measured_and_rejected_knob_kernels_stay_bit_identical(rt, orc, golden_dir, cornell, knob):
    """RTX_OPT_SHADE_DENSE (k_shade_dense: hits compacted through an LDS ring before shading) and RTX_OPT_OCCLUDER_CACHE (any-hit rays first test the triangle that occluded
    the lane's previous ray) are kept as knobs after they measured slower: they must keep producing the oracle's image bit for bit and its ray counts —
    Cornell through the general BVH path (Lambert-only and GGX instantiations) and the reference's garage scene, plus one ReSTIR frame pair for the occluder cache"""
    garage = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    for sc, W, H, flags in ((cornell, 160, 90, 1), (cornell, 160, 90, 0), (garage, 192, 108, 0)):
        p = rt.Params(width=W, height=H, spp=3, max_bounces=6, nee_samples=2, rr_start=2, flags=flags, frame_seed=4)
        oa, oc = orc.Oracle().load(sc, W / H).render(p)
        c = rt.Context(0); c.set_option(rt.OPT_SMALL_SCENE, 0)
        c.set_option(rt.OPT_SHADE_DENSE, 1 if knob in ("shade_dense", "both") else 0)
        c.set_option(rt.OPT_OCCLUDER_CACHE, 1 if knob in ("occluder_cache", "both") else 0)
        c.upload(sc, W / H)
        for _ in range(2):                                                  # (the second frame: launch sizes predicted, lanes carry occluders over from the first)
            c.clear(W, H); c.render(p); st = c.stats()
            assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, (knob, flags)
            assert np.array_equal(bits(c.read_accum()), bits(oa)), (knob, flags)
        c.close()
    if knob != "shade_dense":
        W, H = 96, 54
        o = orc.Oracle().load(garage, W / H)
        vp = garage.view_proj(W / H); o.set_camera(*vp); o.set_camera(*vp)
        c = rt.Context(0); c.set_option(rt.OPT_OCCLUDER_CACHE, 1); c.upload(garage, W / H)
        c.set_camera(*vp); c.set_camera(*vp)
        c.restir_reset(); c.clear(W, H)
        acc_o, st = np.zeros((H, W, 4), np.float32), None
        for f in range(2):
            p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=20 + f)
            c.render_restir(p); s = c.stats()
            acc_o, st, cnt = o.restir_frames(p, acc_o, st)
            assert (s.rays_primary, s.rays_extension, s.rays_shadow) == cnt, f
            assert np.array_equal(bits(c.read_accum()), bits(acc_o)), f
        c.close()


def test_tapered_subqueues_are_result_neutral(rt, orc, golden_dir, cornell):
    """RTX_OPT_TAPER at a size the oracle renders in seconds: 640 x 360 x 4 spp with ONE sub-queue per CU (256 sub-queues of 14 chunks on average, so that the tapered deal is
    active: it needs >= 4 chunks per sub-queue), in one batch and in two (RTX_OPT_PATHS_PER_BATCH), on the reference's garage scene (general path) and on Cornell (fused
    tiny-scene kernels, which take the sub-queues longest first): equal sub-queues, the default taper and the steepest one give the oracle's image bit for bit"""
    garage = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 640, 360
    for sc, flags in ((garage, 0), (cornell, 1)):
        p = rt.Params(width=W, height=H, spp=4, max_bounces=6, nee_samples=1, rr_start=2, flags=flags, frame_seed=9)
        o = orc.Oracle().load(sc, W / H); o.set_threads(_host_threads())
        oa, oc = o.render(p)
        for batch_paths in (0, W * H * 2):
            c = rt.Context(0); c.set_option(rt.OPT_BLOCKS_PER_CU, 1)
            if batch_paths:
                c.set_option(rt.OPT_PATHS_PER_BATCH, batch_paths)
            c.upload(sc, W / H)
            for taper in (0, 1, 8):
                c.set_option(rt.OPT_TAPER, taper)
                c.clear(W, H); c.render(p); st = c.stats()
                assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, (flags, batch_paths, taper)
                assert np.array_equal(bits(c.read_accum()), bits(oa)), (flags, batch_paths, taper)
            c.close()


def test_merged_subqueues_of_thin_launches_are_result_neutral(rt, orc, golden_dir):
    """RTX_OPT_MERGE_RAYS: a workgroup of the persistent traversal kernels takes several consecutive sub-queues when the previous call's counters predict a thin launch.
    garage.obj + monke.obj at 640 x 360 x 8 spp with 64 sub-queues per CU (7 200 sub-queues: two per workgroup is the most one round of resident workgroups allows):
    frames rendered with the option off, at its default and at 2^20 (every launch merged) must equal the oracle's image bit for bit, ray counts included"""
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 640, 360
    p = rt.Params(width=W, height=H, spp=8, max_bounces=8, nee_samples=2, rr_start=2, flags=0, frame_seed=11)
    o = orc.Oracle().load(sc, W / H); o.set_threads(_host_threads())
    oa, oc = o.render(p)
    c = rt.Context(0); c.set_option(rt.OPT_BLOCKS_PER_CU, 64); c.upload(sc, W / H)
    for merge_rays in (0, 1024, 1 << 20, 1 << 20):              # (the first call of a context has no prediction: one sub-queue per workgroup whatever the option says)
        c.set_option(rt.OPT_MERGE_RAYS, merge_rays)
        c.clear(W, H); c.render(p); st = c.stats()
        assert (st.rays_primary, st.rays_extension, st.rays_shadow) == oc, merge_rays
        assert np.array_equal(bits(c.read_accum()), bits(oa)), merge_rays
    for bad in (-1, (1 << 20) + 1):
        with pytest.raises(rt.RtxError):
            c.set_option(rt.OPT_MERGE_RAYS, bad)
    c.close()


@pytest.mark.parametrize("flags", [1, 0])
def test_fused_dispatch_order_and_subqueue_count_are_result_neutral(rt, orc, cornell, flags):
    """the fused tiny-scene kernels take their sub-queues longest first (k_order_queues) and exist in a Lambert-only and a general
    instantiation: image and ray counts must not depend on the dispatch order or on the number of sub-queues, and both
    instantiations must equal the oracle"""
    W, H = 256, 144
    p = rt.Params(width=W, height=H, spp=4, max_bounces=6, nee_samples=1, flags=flags)
    ref = cnt0 = None
    for lpt, bpc in ((1, 0), (0, 0), (1, 1), (1, 7), (0, 64)):
        c = rt.Context(0); c.set_option(rt.OPT_LPT_ORDER, lpt); c.set_option(rt.OPT_BLOCKS_PER_CU, bpc); c.upload(cornell, W / H)
        c.clear(W, H); c.render(p); im = c.read_accum(); st = c.stats()
        cnt = (st.rays_primary, st.rays_extension, st.rays_shadow)
        c.close()
        if ref is None:
            ref, cnt0 = im, cnt
            o = orc.Oracle().load(cornell, W / H)
            oa, oc = o.render(p)
            assert np.array_equal(bits(im), bits(oa)) and cnt == oc
        assert np.array_equal(bits(im), bits(ref)) and cnt == cnt0, (lpt, bpc)


def test_bench_runs_a_real_asset_and_reports_bounded_rooflines(golden_dir, tmp_path):
    """(1) SURVEY 8(d) asset policy through the bench itself: with assets/sponza.obj present (tests/golden/garage.obj stands in, via $RTX_ASSETS) `--workload sponza_...` renders
    THAT model and says so in config.scene.  (2) VERDICT r02 3: no fraction of the JSON line exceeds 1, the frame figure prices SURVEY's 224 N_ext + 96 N_shadow + 32 N_px spp,
    and a counter profile recorded with other kernel sources is reported stale instead of being priced."""
    import json, shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copy(os.path.join(golden_dir, "garage.obj"), tmp_path / "sponza.obj"); shutil.copy(os.path.join(golden_dir, "garage.mtl"), tmp_path / "garage.mtl")
    env = dict(os.environ, RTX_ASSETS=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "sponza_1080p_16spp_8b", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    a = json.loads(r.stdout.strip().splitlines()[-1])
    assert a["config"]["scene"].startswith("asset:") and a["config"]["triangles"] == 1254 and a["value"] > 0

    def fractions(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from fractions(v, path + "/" + k)
        elif isinstance(o, (int, float)) and ("frac" in path.rsplit("/", 1)[-1]):
            yield path, o
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads(r.stdout.strip().splitlines()[-1])
    fr = list(fractions(b))
    assert fr and all(0.0 <= v <= 1.0 for _, v in fr), fr
    roof = b["roofline"]
    rays = b["config"]["rays_per_frame"]
    expect = 224.0 * rays["extension"] + 96.0 * rays["shadow"] + 32.0 * 1920 * 1080 * 64
    assert abs(roof["frame"]["alg_bytes"] - expect) <= 1e-6 * expect
    if roof.get("compute"):
        assert 0.0 < roof["compute"]["frac"] <= 1.0 and 0.0 < roof["compute"]["frac_x_lanes"] <= roof["compute"]["frac"]
    else:
        assert roof["traffic"] is None and (roof["traffic_source"] is None or "STALE" in roof["traffic_source"])


def test_bench_restir_extra_prices_its_dominant_class(rt):
    """bench.py's `extra.restir_*` entries (the reference's own frame at 1080p on its start-up scene): every kernel class is timed, the dominant one and the frame carry a
    fraction of the HBM roofline in (0, 1] with the byte model spelled out, and the ray counts are those of pass 1-3 (<= 8 + 2 + 10 rays per pixel)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    r = bench.time_restir(rt, 0, "garage", frames=2)
    assert set(r["kernel_ms_per_frame"]) >= {"raygen", "trace_closest", "shade", "trace_shadow"} and r["dominant_kernel"] in r["kernel_ms_per_frame"]
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["frame_frac"] <= 1.0 and "per pixel" in r["alg_bytes_model"]
    px = 1920 * 1080
    assert r["rays_per_frame"]["primary"] == px and 0 < r["rays_per_frame"]["extension"] <= 7 * px and 0 < r["rays_per_frame"]["shadow"] <= 20 * px
    assert 0.5 < r["ms_per_frame"] < 100.0


def test_bench_two_ranks_assemble_the_single_rank_frame():
    """bench.py's N > 1 flow end to end on ONE GPU: two processes (torch.distributed.run), each renders its pixel tiles on device 0, packs
    its slab, one gather (gloo, staged through the host: two ranks cannot share a GPU under RCCL), unpack; rank 0's JSON line must carry
    the same frame checksum and ray counts as the single-rank run.  The 8-GPU RCCL run differs only in the collective's backend."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--checksum"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29617",
                          os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0"] + common,
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    b = json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["config"]["parallelism"] == "pixel-tiles/2" and b["scaling"] == "strong"
    assert a["accum_sha1"] == b["accum_sha1"]
    assert a["config"]["rays_per_frame"] == b["config"]["rays_per_frame"]
    for k in ("metric", "value", "unit", "ms_per_step", "roofline"):
        assert k in b and b[k] is not None
    # the driver's plain form, no launcher: `python bench.py --gpus 2` starts its own ranks (before it touches the GPU) and relays rank 0's line
    env2 = {k: v for k, v in env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    self_launched = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0"] + common,
                                   capture_output=True, text=True, timeout=900, env=env2, cwd=root)
    assert self_launched.returncode == 0, self_launched.stderr[-3000:]
    c = json.loads([ln for ln in self_launched.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert c["n_gpus"] == 2 and c["ranks_seen"] == 2 and len(c["ms_per_step_by_rank"]) == 2
    assert c["accum_sha1"] == a["accum_sha1"] and c["config"]["rays_per_frame"] == a["config"]["rays_per_frame"]
    # a launcher / --gpus mismatch is an error, not a silently different run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=120,
                         env=dict(env2, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=root)
    assert bad.returncode != 0 and "does not match WORLD_SIZE" in bad.stderr


def test_native_multi_gpu_frame_of_the_cli(tmp_path):
    """SURVEY 8(e) in the C++ host layer: `rtx_render --gpus N` = ONE process, N contexts on N threads, tiles round-robin, pack -> all-gather -> unpack.
    On one GPU the ranks share device 0 and the collective is replaced by device copies (`--gather copy`: two ranks cannot share a device under
    RCCL); the slab layout, the pack / unpack kernels and the threading are the real ones.  1, 2 and 3 ranks must write byte-identical images."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "royaltracer-dx_amd", "rtx_render")
    blobs = []
    for n, scene in ((1, "cornell"), (2, "cornell"), (3, "cornell"), (1, "sponza"), (2, "sponza")):
        out = tmp_path / f"{scene}_{n}.exr"
        r = subprocess.run([exe, "--scene", scene, "--w", "320", "--h", "200", "--spp", "3", "--bounces", "5", "--gpus", str(n), "--devices", ",".join(["0"] * n),
                            "--gather", "copy", "--out", str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert f"on {n} GPUs" in r.stdout
        blobs.append((scene, n, out.read_bytes()))
    for scene, n, b in blobs:
        ref = [x for s, k, x in blobs if s == scene and k == 1][0]
        assert b == ref, (scene, n)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    lib = subprocess.run(["ldd", os.path.join(root, "royaltracer-dx_amd", "librtx_hip.so")], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "librccl" not in lib            # the collective library is linked into the executable only


def test_native_multi_gpu_restir_frames_of_the_cli(tmp_path):
    """`rtx_render --mode restir --gpus N` = MultiGpuFrame::RenderRestir: every rank runs the three passes on its tile rectangle (RTX_FLAG_BLOCK_TILES,
    passes 1 + 2 on the rectangle dilated by 20 px), then ONE exchange per frame — the all-gather of the history records (rtx_restir_pack_state / unpack_state) and of the
    framebuffer tiles.  Three frames with a moving camera (--orbit) on garage.obj + monke.obj and on the procedural atrium: 1, 2, 3 and 4 native ranks (device copies
    standing in for RCCL on the one GPU) and the single-context Renderer facade write byte-identical EXR images; the literal kernels write the same bytes."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "royaltracer-dx_amd", "rtx_render")
    gd = os.path.join(root, "tests", "golden")
    scenes = {"garage": ["--obj", os.path.join(gd, "garage.obj") + "," + os.path.join(gd, "monke.obj"), "--mtl", gd + "/"], "sponza": ["--scene", "sponza"]}
    for name, sargs in scenes.items():
        blobs = {}
        for tag, extra in (("facade", []), ("n1", ["--gpus", "1", "--devices", "0"]), ("n2", ["--gpus", "2", "--devices", "0,0"]), ("n3", ["--gpus", "3", "--devices", "0,0,0"]),
                           ("n4", ["--gpus", "4", "--devices", "0,0,0,0"]), ("n2_literal", ["--gpus", "2", "--devices", "0,0", "--literal"]),
                           # round 5: the history as border strips between neighbouring rectangles (rtx_restir_pack_halo: send / receive per peer) instead of the all-gather
                           ("n2_halo", ["--gpus", "2", "--devices", "0,0", "--halo", "40"]), ("n4_halo", ["--gpus", "4", "--devices", "0,0,0,0", "--halo", "40"]),
                           ("n3_halo_literal", ["--gpus", "3", "--devices", "0,0,0", "--halo", "40", "--literal"])):
            if name == "sponza" and tag in ("n3", "n4", "n3_halo_literal"):
                continue
            out = tmp_path / f"{name}_{tag}.exr"
            cmd = [exe] + sargs + ["--mode", "restir", "--w", "256", "--h", "144", "--frames", "3", "--orbit", "2", "--gather", "copy", "--out", str(out)] + extra
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (cmd, r.stderr[-2000:])
            assert r.stdout.count("frame ") == 3
            if "halo" in tag:
                assert r.stdout.count("border strips (halo)") == 3 and r.stdout.count("stale history reads 0") == 3, r.stdout[-1500:]
            blobs[tag] = out.read_bytes()
        for tag, b in blobs.items():
            assert b == blobs["facade"], (name, tag)


def test_spinning_instance_from_the_cpp_host(rt, orc, golden_dir, tmp_path):
    """a moving instance reachable from the C++ host, as the reference moves instance 1 and refits its TLAS every frame (Renderer.cpp:431-452, 594,
    2091-2121).  (a) Renderer::SetInstanceTransform + OnUpdate (a transform-only commit: the resident tree refits on the GPU) through the facade's C entry points: three
    ReSTIR frames on garage.obj + monke.obj with the monkey turning equal the oracle's frames bit for bit — the oracle rebuilds its tree from scratch, and the temporal pass
    reprojects through prevObjectToWorld.  (b) `rtx_render --spin`: the facade, one native rank and two native ranks (MultiGpuFrame::SetInstanceTransform: EVERY rank refits
    its replica) write byte-identical images, path tracer and ReSTIR frame."""
    import subprocess
    sc = rt.Scene.from_obj([os.path.join(golden_dir, "garage.obj"), os.path.join(golden_dir, "monke.obj")], golden_dir + "/")
    W, H = 96, 56
    mats = []
    for k in range(3):
        ang = np.float32(1.57 + 0.2 * k)
        m = np.eye(4, dtype=np.float32); m[0, 0] = np.cos(ang); m[0, 2] = -np.sin(ang); m[2, 0] = np.sin(ang); m[2, 2] = np.cos(ang); m[3, 1] = np.float32(0.03 * k)
        mats.append(m.reshape(16))
    o = orc.Oracle().load(sc, W / H)
    acc_o, st = np.zeros((H, W, 4), np.float32), None
    for k in range(3):
        o.set_camera(*sc.view_proj(W / H))
        o.set_instance_transform(1, mats[k])
        acc_o, st, _ = o.restir_frames(rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0, frame_seed=k + 1), acc_o, st)
    r = rt.Renderer(W, H, "spin", 0)
    r.set_scene(sc); r.set_mode(1); r.on_init()
    for k in range(3):
        r.set_instance_transform(1, mats[k])
        r.on_update(); r.on_render()
    acc = r.read_accum()
    assert np.array_equal(bits(acc), bits(acc_o))
    r.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "royaltracer-dx_amd", "rtx_render")
    sargs = ["--obj", os.path.join(golden_dir, "garage.obj") + "," + os.path.join(golden_dir, "monke.obj"), "--mtl", golden_dir + "/"]
    for mode, margs in (("restir", ["--mode", "restir"]), ("pt", ["--spp", "2", "--bounces", "4"])):
        blobs = {}
        for tag, extra in (("facade", []), ("n1", ["--gpus", "1", "--devices", "0"]), ("n2", ["--gpus", "2", "--devices", "0,0"])):
            out = tmp_path / f"spin_{mode}_{tag}.exr"
            cmd = [exe] + sargs + margs + ["--w", "192", "--h", "108", "--frames", "3", "--spin", "9", "--gather", "copy", "--out", str(out)] + extra
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, (cmd, p.stderr[-2000:])
            assert p.stdout.count("frame ") == 3 and (tag == "facade" or p.stdout.count("refit on") == 2)
            blobs[tag] = out.read_bytes()
        assert blobs["n1"] == blobs["facade"] and blobs["n2"] == blobs["facade"], mode


def test_rccl_collective_path_with_one_rank(tmp_path):
    """What one GPU allows of SURVEY 8(e)'s collective: the frame's RCCL path with a ONE-rank communicator.  (a) native: `rtx_render --gpus 1 --gather rccl --force-gather` =
    ncclCommInitAll over one device, pack -> ncclGroupStart / ncclAllGather / ncclGroupEnd on the context's stream -> unpack; the image must be byte-identical to the run
    without a collective, for the path tracer and for the ReSTIR frame (history + tiles).  (b) torch.distributed, backend nccl (= RCCL), world size 1: the all_gather_into_tensor of
    sharding.gather_slabs on a device slab returns the slab.  N > 1 ranks need N GPUs (two ranks cannot share a device under RCCL): the 8-rank gather itself stays unexecuted here."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "royaltracer-dx_amd", "rtx_render")
    for mode in (["--spp", "2", "--bounces", "4"], ["--mode", "restir", "--frames", "2", "--orbit", "1"]):
        blobs = []
        for extra in ([], ["--gather", "rccl", "--force-gather"]):
            out = tmp_path / f"o{len(blobs)}.exr"
            r = subprocess.run([exe, "--scene", "sponza", "--w", "192", "--h", "108", "--gpus", "1", "--devices", "0", "--out", str(out)] + mode + extra,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            blobs.append(out.read_bytes())
        assert blobs[0] == blobs[1], mode
    code = (
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g\n"
        "g.load_package()\n"
        "from royaltracer_dx_amd import sharding\n"
        "os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29641')\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(0)\n"
        "dist, rank, world = sharding.init_process_group('nccl', dev)\n"
        "slab = torch.arange(1 << 20, dtype=torch.float32, device=dev)\n"
        "out = sharding.gather_slabs(dist, slab)\n"
        "torch.cuda.synchronize()\n"
        "assert world == 1 and torch.equal(out, slab)\n"
        "print('backend', dist.get_backend(), 'ok')\n"
        "dist.destroy_process_group()\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "backend nccl ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_two_contexts_from_two_threads(rt, cornell):
    """SURVEY 8(b) threading contract: a context is not thread-safe, but different contexts may be driven from different threads.
    Two threads render different workloads concurrently on the same GPU (different scenes, options, streams); each result must be
    bit-identical to the same work done alone — no hidden process-wide state"""
    import threading
    sp = rt.Scene.sponza_class(40000, 260)
    jobs = [dict(scene=cornell, opts={}, p=rt.Params(width=192, height=108, spp=3, max_bounces=6, nee_samples=2, flags=1)),
            dict(scene=sp, opts={rt.OPT_TRACE_SCHED: 2, rt.OPT_REFILL_MIN: 20, rt.OPT_BLOCKS_PER_CU: 4}, p=rt.Params(width=160, height=90, spp=2, max_bounces=5, nee_samples=1, flags=0))]

    def run(job, out, reps):
        c = rt.Context(0)
        for k, v in job["opts"].items():
            c.set_option(k, v)
        c.upload(job["scene"], job["p"].width / job["p"].height)
        imgs = []
        for _ in range(reps):
            c.clear(job["p"].width, job["p"].height); c.render(job["p"]); imgs.append(c.read_accum())
        c.close()
        out.append(imgs)

    alone = []
    for j in jobs:
        o = []; run(j, o, 1); alone.append(o[0][0])
    outs = [[], []]
    th = [threading.Thread(target=run, args=(jobs[k], outs[k], 6)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(2):
        assert len(outs[k]) == 1 and len(outs[k][0]) == 6
        for im in outs[k][0]:
            assert np.array_equal(bits(im), bits(alone[k]))
