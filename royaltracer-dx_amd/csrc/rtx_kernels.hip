// rtx_kernels.hip — the wavefront path-tracing kernels for gfx950 (CDNA4, wave64).
//
// One sample batch is a set of paths with fixed slots ("pid"); per-path state lives in SoA float4 arrays in HBM.
// General scenes: each bounce runs  k_trace_closest -> k_shade -> k_trace_shadow[j]  over workgroup-private index
// sub-queues that k_shade re-compacts with a wave ballot + prefix sum (one LDS atomic per wave, no global atomics);
// the traversal kernels are persistent waves with dynamic ray fetch, the top of the BVH and the per-lane stack live
// in LDS.  Tiny scenes (<= 64 triangles, the Cornell Box): no BVH, a packed-FP32 plane/edge pre-test with
// scalar-loaded coefficients, and ONE fused kernel per bounce (k_bounce_small).  The reference's own passes
// (k_v6_pass1, k_restir_pass2/3) are thread-per-pixel kernels.  MFMA is unused on purpose: nothing here is a dense
// contraction.
//
// Parity-critical arithmetic (ray/triangle test, surface reconstruction, BSDF, light sampling, path
// throughput) follows rtx_math.hpp / rtx_bsdf.hpp with the library-wide -ffp-contract=off.  Ray/box
// tests are NOT parity-critical (closest hit is defined as the minimum over all triangles with a
// lowest-id tie break, any-hit as existence), they only have to be conservative.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rtx_kernels.hpp"

namespace rtx {

constexpr int kBlock = 256;
constexpr uint32_t kMaxNee = 16;

// ---------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Stream compaction into a WORKGROUP-PRIVATE sub-queue: every lane of the wave must call this (convergent).
// The counter lives in LDS (one ds_add per wave); there are no global atomics anywhere in the render loop —
// a single global counter saturates at ~88 returning atomics/us on MI355X and was the first bottleneck found
// (profiles/r01_cornell_c2_v1.md).
__device__ __forceinline__ uint32_t block_push(bool pred, uint32_t* lds_counter) {
    const unsigned long long mask = __ballot(pred);
    const uint32_t cnt = (uint32_t)__popcll(mask);
    if (cnt == 0) return 0xFFFFFFFFu;                    // wave-uniform
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(lds_counter, cnt);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return base + prefix;
}

// ---------------------------------------------------------------------------------------------
// pixel <-> local path-slot mapping (shard tiles, 8x8 pixel blocks inside a tile so that one wave
// covers a compact screen region)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slot_to_pixel(const DevFrame& f, uint32_t pl, uint32_t& x, uint32_t& y) {
    const uint32_t ts2 = 2u * f.tile_shift;                  // tile_size is a power of two
    const uint32_t k = pl >> ts2, r = pl & ((1u << ts2) - 1u);
    const uint32_t t = f.shard_rank + k * f.shard_count;
    if (t >= f.tiles_x * f.tiles_y) return false;
    const uint32_t ty = t / f.tiles_x, tx = t - ty * f.tiles_x;
    const uint32_t bshift = f.tile_shift - 3u;               // 8x8 pixel blocks per tile row = 2^bshift
    const uint32_t blk = r >> 6, ln = r & 63u;
    const uint32_t bx = blk & ((1u << bshift) - 1u), by = blk >> bshift;
    x = (tx << f.tile_shift) + bx * 8u + (ln & 7u);
    y = (ty << f.tile_shift) + by * 8u + (ln >> 3);
    return x < f.width && y < f.height;
}

// primary ray, RayGen_v6_pass1.hlsl:51-95
__device__ __forceinline__ void primary_ray(const CameraGPU& cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float jx, float jy, f3& o, f3& d) {
    const float dx = (((float)x + jx) / (float)W) * 2.0f - 1.0f;
    const float dy = (((float)y + jy) / (float)H) * 2.0f - 1.0f;
    const float* P = cam.projI; const float* Vi = cam.viewI;
    const float ndy = -dy;
    f3 tg = mk3(P[0] * dx + P[4] * ndy + P[8] + P[12], P[1] * dx + P[5] * ndy + P[9] + P[13], P[2] * dx + P[6] * ndy + P[10] + P[14]);
    d = normalize(xform_dir(Vi, tg));
    o = mk3(Vi[12], Vi[13], Vi[14]);
}

// ---------------------------------------------------------------------------------------------
// raygen: one thread per path slot of the batch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_raygen(DevFrame f, DevPaths p, const CameraGPU* __restrict__ cam_p, uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount) {
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_n;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t* myq = queue + (size_t)blockIdx.x * f.qcap;
    // chunk c = 256 consecutive path slots of ONE sample; chunks are dealt round-robin to workgroups so that
    // every workgroup's sub-queue holds a representative sample of the image (load balance across bounces)
    const uint32_t nchunks = f.chunks_per_sample * f.batch_spp;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint32_t sl = c / f.chunks_per_sample, cl = c - sl * f.chunks_per_sample;   // wave-uniform (SALU)
        const uint32_t pl = cl * kBlock + threadIdx.x;
        const uint32_t pid = sl * f.npl + pl;
        uint32_t x = 0, y = 0;
        const bool valid = slot_to_pixel(f, pl, x, y);
        if (valid) {
            uint32_t s0, s1; seed_init(x, y, f.sample_first + sl, f.frame_seed, s0, s1);
            float jx = 0.0f, jy = 0.0f;
            if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }   // RayGen.hlsl:84-85
            f3 o, d; primary_ray(cam, f.width, f.height, x, y, jx, jy, o, d);
            p.ray_o[pid] = {o.x, o.y, o.z, u2f(s1)};
            p.ray_d[pid] = {d.x, d.y, d.z, 1.0f};
            p.thr[pid] = {1.0f, 1.0f, 1.0f, u2f(s0)};
            p.rad[pid] = {0.0f, 0.0f, 0.0f, 0.0f};
        }
        const uint32_t slot = block_push(valid, &s_n);
        if (valid) myq[slot] = pid;
    }
    __syncthreads();
    if (threadIdx.x == 0) qcount[blockIdx.x] = s_n;
}

// ---------------------------------------------------------------------------------------------
// BVH traversal
// ---------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type: through a generic pointer hipcc emits flat_load /
// flat_store for the staged nodes and the traversal stack instead of ds_read_b128 / ds_write_b32 (found with
// SQ_INSTS_LDS vs SQ_INSTS_VMEM_RD in profiles/r01_pmc_v2.md).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2v splat2(float x) { f2v r = {x, x}; return r; }
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2u lds_u2;
struct TraceLds {
    const lds_v4f* nodes;   // LDS copy of nodes [0, lds_nodes)
    const lds_v4f* tris;    // LDS copy of tris  [0, lds_tris)
    lds_u2* stack;          // [depth][kBlock] sibling-group entries
};

// stage the top of the BVH and the first triangles into LDS (coalesced 16-B copies)
__device__ __forceinline__ TraceLds stage_lds(const DevScene& sc, F4* lds_generic) {
    TraceLds L;
    lds_v4f* ln = (lds_v4f*)lds_generic; lds_v4f* lt = ln + (size_t)sc.lds_nodes * 5;
    const v4f* gn = (const v4f*)sc.nodes; const v4f* gt = (const v4f*)(sc.nsmall ? sc.small_tris : sc.tris);
    for (uint32_t i = threadIdx.x; i < sc.lds_nodes * 5u; i += kBlock) ln[i] = gn[i];
    for (uint32_t i = threadIdx.x; i < sc.lds_tris * 3u; i += kBlock) lt[i] = gt[i];
    L.nodes = ln; L.tris = lt;
    L.stack = (lds_u2*)(lt + (size_t)sc.lds_tris * 3);
    return L;
}

// Moeller-Trumbore with the fixed operation order shared with the oracle (a11).  Exclusive (tmin, tmax).
// (A variant that checks the numerators conservatively before the IEEE division measured no faster: 32.4 vs 31.6 ms.)
__device__ __forceinline__ bool tri_test(f3 o, f3 d, v4f v0w, v4f e1w, v4f e2w, float tmin, float tmax, float& t, float& u, float& v) {
    const f3 v0 = mk3(v0w.x, v0w.y, v0w.z), e1 = mk3(e1w.x, e1w.y, e1w.z), e2 = mk3(e2w.x, e2w.y, e2w.z);
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const f3 s = o - v0;
    u = dot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 q = cross(s, e1);
    v = dot(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot(e2, q) * inv;
    return t > tmin && t < tmax;
}

// ---- compressed 8-wide node step ---------------------------------------------------------------------------------
// One traversal step fetches a Node8GPU (five 16-B loads, or five ds_read_b128 for the staged top of the tree) and tests its
// eight child boxes.  Planes are byte offsets on the node's power-of-two grid, so
//     t_plane = q * (2^e * idir) + (p - o) * idir        (cvt + fma per plane; near / far rows picked by the ray's octant)
// CONSERVATIVENESS (the closest hit must stay the minimum over ALL triangles): the builder rounds the boxes outward in exact
// arithmetic; 2^e * idir is exact; a = fl(fl(p - o) * idir) carries a relative error <= 2^-23, so the near planes use
// a - |a| 2^-22 and the far planes a + |a| 2^-22; what is left is relative to t and covered by kSlabLo / kSlabHi.
// Hit children are visited in increasing (slot ^ octant) order; the rest of a node's hit children stay together in
// ONE stack entry (base index + hit bits + internal mask), so the stack holds one entry per level.
struct Node8R { v4f h0; v4u h1, q0, q1, q2; };
struct Grp { uint32_t base, bits; };                 // node group: child_base, ordered internal hits (bits 0-7) | imask << 8
struct TriGrp { uint32_t base, bits, valid; };        // triangle group: tri_base, hit triangle bits, the node's trivalid
constexpr float kPlaneEps = 2.384185791015625e-07f;   // 2^-22
// relative widening of the slab interval.  It must cover the error of the TRIANGLE test's t, not only the slab arithmetic: a
// hit next to a vertex of a small triangle seen from far away has a Moeller-Trumbore t that is off by ~1e-5 relative (found by
// test_wide_bvh_equals_brute_force_on_hostile_soups: coincident duplicates lost their lowest-id tie at 2e-6), so 5e-5.
constexpr float kSlabLo = 0.99995f, kSlabHi = 1.00005f;

__device__ __forceinline__ Node8R load_node8(const DevScene& sc, const TraceLds& L, uint32_t idx) {
    Node8R N;
    if (idx < sc.lds_nodes) { const lds_v4f* n = L.nodes + idx * 5u; N.h0 = n[0]; N.h1 = (v4u)n[1]; N.q0 = (v4u)n[2]; N.q1 = (v4u)n[3]; N.q2 = (v4u)n[4]; }
    else { const v4f* n = (const v4f*)sc.nodes + (size_t)idx * 5u; N.h0 = n[0]; N.h1 = (v4u)n[1]; N.q0 = (v4u)n[2]; N.q1 = (v4u)n[3]; N.q2 = (v4u)n[4]; }
    return N;
}
__device__ __forceinline__ uint32_t ray_octant(f3 idir) { return (idir.x < 0.0f ? 1u : 0u) | (idir.y < 0.0f ? 2u : 0u) | (idir.z < 0.0f ? 4u : 0u); }
__device__ __forceinline__ float byte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK

// tests the 8 children; G = this node's internal hits in octant order, T = the triangles of its hit leaf children
__device__ __forceinline__ void node8_hits(const Node8R& N, f3 o, f3 idir, uint32_t oct, float tmin, float tbest, Grp& G, TriGrp& T) {
    const uint32_t w = f2u(N.h0.w);
    const float sx = u2f((w & 0xffu) << 23) * idir.x, sy = u2f((w & 0xff00u) << 15) * idir.y, sz = u2f((w & 0xff0000u) << 7) * idir.z;
    const float ax = (N.h0.x - o.x) * idir.x, ay = (N.h0.y - o.y) * idir.y, az = (N.h0.z - o.z) * idir.z;
    const float anx = __builtin_fmaf(-fabsf(ax), kPlaneEps, ax), afx = __builtin_fmaf(fabsf(ax), kPlaneEps, ax);
    const float any_ = __builtin_fmaf(-fabsf(ay), kPlaneEps, ay), afy = __builtin_fmaf(fabsf(ay), kPlaneEps, ay);
    const float anz = __builtin_fmaf(-fabsf(az), kPlaneEps, az), afz = __builtin_fmaf(fabsf(az), kPlaneEps, az);
    const bool nx = (oct & 1u) != 0u, ny = (oct & 2u) != 0u, nz = (oct & 4u) != 0u;
    // rows: q0 = (lox0, lox1, loy0, loy1)  q1 = (loz0, loz1, hix0, hix1)  q2 = (hiy0, hiy1, hiz0, hiz1)
    const uint32_t qnx[2] = {nx ? N.q1.z : N.q0.x, nx ? N.q1.w : N.q0.y}, qfx[2] = {nx ? N.q0.x : N.q1.z, nx ? N.q0.y : N.q1.w};
    const uint32_t qny[2] = {ny ? N.q2.x : N.q0.z, ny ? N.q2.y : N.q0.w}, qfy[2] = {ny ? N.q0.z : N.q2.x, ny ? N.q0.w : N.q2.y};
    const uint32_t qnz[2] = {nz ? N.q2.z : N.q1.x, nz ? N.q2.w : N.q1.y}, qfz[2] = {nz ? N.q1.x : N.q2.z, nz ? N.q1.y : N.q2.w};
    uint32_t hits = 0;
    const f2v vsx = splat2(sx), vsy = splat2(sy), vsz = splat2(sz);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {                     // two children per iteration on packed FP32 (v_pk_fma_f32 / v_pk_mul_f32)
        const int h = k >> 2, b = k & 3;
        const f2v bnx = {byte_f(qnx[h], b), byte_f(qnx[h], b + 1)}, bny = {byte_f(qny[h], b), byte_f(qny[h], b + 1)}, bnz = {byte_f(qnz[h], b), byte_f(qnz[h], b + 1)};
        const f2v bfx = {byte_f(qfx[h], b), byte_f(qfx[h], b + 1)}, bfy = {byte_f(qfy[h], b), byte_f(qfy[h], b + 1)}, bfz = {byte_f(qfz[h], b), byte_f(qfz[h], b + 1)};
        const f2v tnx = fma2(bnx, vsx, splat2(anx)), tny = fma2(bny, vsy, splat2(any_)), tnz = fma2(bnz, vsz, splat2(anz));
        const f2v tfx = fma2(bfx, vsx, splat2(afx)), tfy = fma2(bfy, vsy, splat2(afy)), tfz = fma2(bfz, vsz, splat2(afz));
        const f2v lo = {fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, tmin)), fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, tmin))};
        const f2v hi = {fminf(fminf(tfx.x, tfy.x), fminf(tfz.x, tbest)), fminf(fminf(tfx.y, tfy.y), fminf(tfz.y, tbest))};
        const f2v los = lo * kSlabLo, his = hi * kSlabHi;                 // lo >= tmin >= 0
        if (los.x <= his.x) hits |= 1u << k;
        if (los.y <= his.y) hits |= 2u << k;
    }
    const uint32_t imask = w >> 24;
    // internal hits, permuted so that bit j = slot (j ^ oct): lowest set bit = first child to visit
    uint32_t m = hits & imask;
    if (nx) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (ny) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (nz) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    G.base = N.h1.x; G.bits = m | (imask << 8);
    // leaf hits: spread each bit to its nibble and keep the triangles that exist
    uint32_t x = hits & ~imask;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    T.base = N.h1.y; T.valid = N.h1.z; T.bits = (x * 15u) & N.h1.z;
}

// traversal stack of sibling groups: per-lane column in LDS (conflict-free 8-byte accesses), or a private array (scratch)
struct StackLds { lds_u2* col; __device__ __forceinline__ void put(int i, Grp g) { v2u v = {g.base, g.bits}; col[i * kBlock] = v; }
                  __device__ __forceinline__ Grp get(int i) const { const v2u v = col[i * kBlock]; return Grp{v.x, v.y}; } };
constexpr int kPrivStack = 32;
struct StackPriv { Grp a[kPrivStack]; __device__ __forceinline__ void put(int i, Grp g) { a[i] = g; } __device__ __forceinline__ Grp get(int i) const { return a[i]; } };

// pick the first child of group G (which has internal hits), keep the remaining siblings on the stack, test the child's
// eight children: G / T become the child's groups
template <class STK>
__device__ __forceinline__ void descend8(const DevScene& sc, const TraceLds& L, f3 o, f3 idir, uint32_t oct, float tmin, float tbest,
                                         Grp& G, TriGrp& T, STK& stk, int& sp) {
    const uint32_t k = (uint32_t)__builtin_ctz(G.bits);
    const uint32_t rest = G.bits & (G.bits - 1u);
    if (rest & 0xffu) { stk.put(sp, Grp{G.base, rest}); sp++; }
    const uint32_t slot = k ^ oct;
    const uint32_t idx = G.base + (uint32_t)__builtin_popcount((G.bits >> 8) & ((1u << slot) - 1u));
    const Node8R N = load_node8(sc, L, idx);
    node8_hits(N, o, idir, oct, tmin, tbest, G, T);
}
// index of the triangle behind bit `bit` of a triangle group
__device__ __forceinline__ uint32_t tri_slot8(const TriGrp& T, uint32_t bit) { return T.base + (uint32_t)__builtin_popcount(T.valid & ((1u << bit) - 1u)); }

template <bool ANY>
__device__ __forceinline__ void traverse(const DevScene& sc, const TraceLds& L, f3 o, f3 d, float tmin, float tmax,
                                         float& bt, float& bu, float& bv, uint32_t& bprim) {
    // zero direction components -> huge finite reciprocal (keeps the slab test NaN-free and conservative)
    const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    const float dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    const float dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    const f3 idir = mk3(__builtin_amdgcn_rcpf(dxs), __builtin_amdgcn_rcpf(dys), __builtin_amdgcn_rcpf(dzs));
    const uint32_t oct = ray_octant(idir);
    bt = tmax; bu = 0.0f; bv = 0.0f; bprim = kMissPrim;
    StackLds stk; stk.col = L.stack + threadIdx.x;
    int sp = 0;
    Grp G{0u, (1u << oct) | (1u << 8)};                 // the root as slot 0 of a virtual parent
    TriGrp T{0u, 0u, 0u};
    while (true) {
        if (G.bits & 0xffu) descend8(sc, L, o, idir, oct, tmin, bt, G, T, stk, sp);
        while (T.bits) {
            const uint32_t bit = (uint32_t)__builtin_ctz(T.bits);
            T.bits &= T.bits - 1u;
            const uint32_t slot = tri_slot8(T, bit);
            v4f v0, e1, e2;
            if (slot < sc.lds_tris) { const lds_v4f* t = L.tris + slot * 3u; v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            else { const v4f* t = (const v4f*)(sc.tris + slot); v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            float t, u, w;
            if (tri_test(o, d, v0, e1, e2, tmin, tmax, t, u, w)) {
                if (ANY) { bprim = 0u; return; }
                const uint32_t gid = f2u(v0.w);
                if (t < bt || (t == bt && gid < bprim)) { bt = t; bu = u; bv = w; bprim = gid; }
            }
        }
        if (!(G.bits & 0xffu)) {
            if (sp == 0) break;
            sp--; G = stk.get(sp);
        }
    }
}

// closest-hit traversal that also counts node steps and triangle tests (rtx_debug_trace_stats: tree-quality measurements)
__device__ __forceinline__ void traverse_stats(const DevScene& sc, const TraceLds& L, f3 o, f3 d, float tmin, float tmax,
                                         float& bt, float& bu, float& bv, uint32_t& bprim) {
    constexpr bool ANY = false;
    uint32_t nsteps = 0, ntris = 0;
    // zero direction components -> huge finite reciprocal (keeps the slab test NaN-free and conservative)
    const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    const float dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    const float dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    const f3 idir = mk3(__builtin_amdgcn_rcpf(dxs), __builtin_amdgcn_rcpf(dys), __builtin_amdgcn_rcpf(dzs));
    const uint32_t oct = ray_octant(idir);
    bt = tmax; bu = 0.0f; bv = 0.0f; bprim = kMissPrim;
    StackLds stk; stk.col = L.stack + threadIdx.x;
    int sp = 0;
    Grp G{0u, (1u << oct) | (1u << 8)};                 // the root as slot 0 of a virtual parent
    TriGrp T{0u, 0u, 0u};
    while (true) {
        if (G.bits & 0xffu) { descend8(sc, L, o, idir, oct, tmin, bt, G, T, stk, sp); nsteps++; }
        while (T.bits) {
            const uint32_t bit = (uint32_t)__builtin_ctz(T.bits);
            T.bits &= T.bits - 1u; ntris++;
            const uint32_t slot = tri_slot8(T, bit);
            v4f v0, e1, e2;
            if (slot < sc.lds_tris) { const lds_v4f* t = L.tris + slot * 3u; v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            else { const v4f* t = (const v4f*)(sc.tris + slot); v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            float t, u, w;
            if (tri_test(o, d, v0, e1, e2, tmin, tmax, t, u, w)) {
                if (ANY) { bprim = 0u; return; }
                const uint32_t gid = f2u(v0.w);
                if (t < bt || (t == bt && gid < bprim)) { bt = t; bu = u; bv = w; bprim = gid; }
            }
        }
        if (!(G.bits & 0xffu)) {
            if (sp == 0) break;
            sp--; G = stk.get(sp);
        }
    }
    bu = (float)nsteps; bv = (float)ntris;
}

// Tiny-scene path (sc.nsmall != 0, all triangles staged in LDS): no BVH.  Phase 1 runs a CONSERVATIVE plane-form
// pre-test of every triangle in a wave-uniform loop — two triangles per iteration on packed-FP32 instructions,
// their coefficients wave-uniform (one s_load_dwordx16 pair per iteration, no LDS/VMEM traffic, no divergence) —
// and collects a per-lane candidate bit mask.  Phase 2 runs the exact Moeller-Trumbore test on the few
// candidates of each lane.  The result is the same minimum-over-all-triangles as the BVH path and the oracle's
// brute force: phase 1 only removes triangles that the exact test would reject (tolerances: the edge-plane distance
// delta and the t margin, built in rtx_scene_host.cpp).

template <bool ANY>
__device__ __forceinline__ void traverse_small(const DevScene& sc, const SmallRecPair* __restrict__ sp, const TraceLds& L, f3 o, f3 d, float tmin, float tmax,
                                               float& bt, float& bu, float& bv, uint32_t& bprim, uint32_t nrec, unsigned long long keep = ~0ull) {
    // keep (wave-uniform): bit r clear = no ray of this wave can touch record r (primary-ray packet culling); nrec = sc.nsmall, or sc.nsmall_occ for NEE shadow segments (both end points inside the scene's convex hull: the records
    // after the first nsmall_occ are faces OF that hull and cannot lie between them, rtx_scene_host.cpp)
    bt = tmax; bu = 0.0f; bv = 0.0f; bprim = kMissPrim;
    uint32_t cand_lo = 0u, cand_hi = 0u;
    const uint32_t npairs = (nrec + 1u) >> 1;
    const f2v dx = splat2(d.x), dy = splat2(d.y), dz = splat2(d.z), ox = splat2(o.x), oy = splat2(o.y), oz = splat2(o.z);
    const f2v cm = splat2(sc.small_cm), c5 = splat2(1e-5f), vtmin = splat2(tmin), vtmax = splat2(tmax), dl = splat2(sc.small_delta);
#pragma unroll 2
    for (uint32_t kp = 0; kp < npairs; kp++) {          // wave-uniform
        if (!((keep >> (2u * kp)) & 3ull)) continue;
        const f2v* __restrict__ R = (const f2v*)sp[kp].r;
        const f2v nd = fma2(R[2], dz, fma2(R[1], dy, R[0] * dx));
        const f2v no = R[3] - fma2(R[2], oz, fma2(R[1], oy, R[0] * ox));
        f2v ind; ind.x = __builtin_amdgcn_rcpf(nd.x); ind.y = __builtin_amdgcn_rcpf(nd.y);
        const f2v t = no * ind;
        const f2v px = fma2(t, dx, ox), py = fma2(t, dy, oy), pz = fma2(t, dz, oz);
        const f2v e0 = fma2(R[6], pz, fma2(R[5], py, fma2(R[4], px, R[7])));
        const f2v e1 = fma2(R[10], pz, fma2(R[9], py, fma2(R[8], px, R[11])));
        const f2v e2 = fma2(R[14], pz, fma2(R[13], py, fma2(R[12], px, R[15])));
        const f2v e3 = fma2(R[18], pz, fma2(R[17], py, fma2(R[16], px, R[19])));
        const f2v mt = fma2(cm, __builtin_elementwise_abs(ind), c5 * __builtin_elementwise_abs(t));
        // all slack values must be >= 0: t in [tmin - mt, tmax + mt] and P within delta of the inside of every edge
        const f2v a0 = (t + mt) - vtmin, a1 = (vtmax + mt) - t, b0 = e0 + dl, b1 = e1 + dl, b2 = e2 + dl, b3 = e3 + dl;
        const float m0 = fminf(fminf(fminf(a0.x, a1.x), fminf(b0.x, b1.x)), fminf(b2.x, b3.x));
        const float m1 = fminf(fminf(fminf(a0.y, a1.y), fminf(b0.y, b1.y)), fminf(b2.y, b3.y));
        const bool c0 = (m0 >= 0.0f) || (fabsf(nd.x) < 1e-3f);      // grazing rays always go to the exact test
        const bool c1 = (m1 >= 0.0f) || (fabsf(nd.y) < 1e-3f);
        const uint32_t bit = 1u << ((2u * kp) & 31u);
        const uint32_t add = (c0 ? bit : 0u) | (c1 ? (bit << 1) : 0u);
        if (kp < 16u) cand_lo |= add; else cand_hi |= add;
    }
    unsigned long long cand = ((unsigned long long)cand_hi << 32) | cand_lo;
    while (cand) {                                     // per-lane: exact test of the triangles of each candidate record
        const uint32_t k = (uint32_t)__builtin_ctzll(cand);
        cand &= cand - 1ull;
#pragma unroll
        for (uint32_t h = 0; h < 2u; h++) {
            const lds_v4f* tp = L.tris + (2u * k + h) * 3u;
            const v4f v0 = tp[0], e1 = tp[1], e2 = tp[2];
            float t, u, w;
            if (tri_test(o, d, v0, e1, e2, tmin, tmax, t, u, w)) {
                if (ANY) { bprim = 0u; return; }
                const uint32_t gid = f2u(v0.w);
                if (t < bt || (t == bt && gid < bprim)) { bt = t; bu = u; bv = w; bprim = gid; }
            }
        }
    }
}

template <bool ANY>
__device__ __forceinline__ void trace_ray(const DevScene& sc, const SmallRecPair* __restrict__ small, const TraceLds& L, f3 o, f3 d, float tmin, float tmax,
                                          float& bt, float& bu, float& bv, uint32_t& bprim) {
    if (sc.nsmall) traverse_small<ANY>(sc, small, L, o, d, tmin, tmax, bt, bu, bv, bprim, sc.nsmall);
    else traverse<ANY>(sc, L, o, d, tmin, tmax, bt, bu, bv, bprim);
}

// Tiny-scene bounce 0: generate the primary ray AND trace it; only paths that hit something are enqueued (their
// hit record goes to p.hit), so the bounce-0 shading kernel runs without the idle lanes of the camera rays that
// leave the scene (43 % of them on the Cornell view).  Missed paths only get their radiance slot zeroed.
// Packet culling for camera rays.  A wave's 64 primary rays share the origin and cover one 8x8 pixel block, so they lie inside
// the pyramid spanned by the block's four corner directions.  Lane r tests record r's polygon against the four side planes of
// that pyramid (widened by 1e-4 of |corner - origin| in L1 norm, ~a tenth of a pixel): a polygon with all corners outside one
// plane cannot be touched by any ray of the wave, and the wave skips its pre-test.  Conservative: only records that the exact
// test would reject for every ray of the block are dropped (Cornell at 1080p: ~3 of 17 records survive per block).
__device__ __forceinline__ unsigned long long packet_keep_mask(const DevScene& sc, const CameraGPU& cam, const DevFrame& f, uint32_t x0, uint32_t y0) {
    f3 o, c[4];
    for (int k = 0; k < 4; k++) {                                      // un-normalised corner directions (x0 + 8 (k & 1), y0 + 8 (k >> 1))
        const float dx = ((float)(x0 + 8u * (uint32_t)(k & 1)) / (float)f.width) * 2.0f - 1.0f;
        const float ndy = -(((float)(y0 + 8u * (uint32_t)(k >> 1)) / (float)f.height) * 2.0f - 1.0f);
        const float* P = cam.projI;
        const f3 tg = mk3(P[0] * dx + P[4] * ndy + P[8] + P[12], P[1] * dx + P[5] * ndy + P[9] + P[13], P[2] * dx + P[6] * ndy + P[10] + P[14]);
        c[k] = xform_dir(cam.viewI, tg);
    }
    o = mk3(cam.viewI[12], cam.viewI[13], cam.viewI[14]);
    const f3 mid = c[0] + c[3];                                        // inside direction (diagonal sum)
    const f3 n[4] = {cross(c[0], c[1]), cross(c[1], c[3]), cross(c[3], c[2]), cross(c[2], c[0])};
    const uint32_t r = lane_id();
    bool culled = false;
    if (r < sc.nsmall) {
        f3 v[4];
        for (int k = 0; k < 4; k++) { const F4 q = sc.small_poly[(size_t)r * 4 + k]; v[k] = mk3(q.x, q.y, q.z) - o; }
        for (int i = 0; i < 4; i++) {
            const float s = dot(n[i], mid) >= 0.0f ? 1.0f : -1.0f;     // orientation: the pyramid's inside has s * dot(n, .) >= 0
            const float nl1 = fabsf(n[i].x) + fabsf(n[i].y) + fabsf(n[i].z);
            bool all_out = true;
            for (int k = 0; k < 4; k++) {
                const float e = 1e-4f * nl1 * (fabsf(v[k].x) + fabsf(v[k].y) + fabsf(v[k].z));
                all_out = all_out && (s * dot(n[i], v[k]) < -e);
            }
            culled = culled || all_out;
        }
    }
    return __ballot(r < sc.nsmall && !culled) | (sc.nsmall & 1u ? (1ull << sc.nsmall) : 0ull);   // (the padding record of an odd count is never inside anyway)
}

__global__ __launch_bounds__(kBlock) void k_raygen_trace_small(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, DevPaths p, const CameraGPU* __restrict__ cam_p,
                                                               uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount, uint32_t* __restrict__ gencount) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_n[2];
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t* myq = queue + (size_t)blockIdx.x * f.qcap;
    const uint32_t nchunks = f.chunks_per_sample * f.batch_spp;
    uint32_t generated = 0;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint32_t sl = c / f.chunks_per_sample, cl = c - sl * f.chunks_per_sample;
        const uint32_t pl = cl * kBlock + threadIdx.x;
        const uint32_t pid = sl * f.npl + pl;
        uint32_t x = 0, y = 0, s0 = 0, s1 = 0;
        const bool valid = slot_to_pixel(f, pl, x, y);
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        if (valid) {
            seed_init(x, y, f.sample_first + sl, f.frame_seed, s0, s1);
            float jx = 0.0f, jy = 0.0f;
            if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }
            primary_ray(cam, f.width, f.height, x, y, jx, jy, o, d);
            generated++;
        }
        // pixel block of this wave: lane 0's pixel (slot_to_pixel lays 8x8 blocks out per wave)
        const uint32_t bx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(x & ~7u)), by0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(y & ~7u));
        const unsigned long long keep = packet_keep_mask(sc, cam, f, bx0, by0);
        float t, u, v; uint32_t prim;
        traverse_small<false>(sc, small, L, o, d, kTMinCam, valid ? kTMax : 0.0f, t, u, v, prim, sc.nsmall, keep);
        const bool hit = valid && prim != kMissPrim;
        if (valid) p.rad[pid] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (hit) {
            p.ray_o[pid] = {o.x, o.y, o.z, u2f(s1)};
            p.ray_d[pid] = {d.x, d.y, d.z, 1.0f};
            p.thr[pid] = {1.0f, 1.0f, 1.0f, u2f(s0)};
            p.hit[pid] = {t, u, v, u2f(prim)};
        }
        const uint32_t slot = block_push(hit, &s_n[0]);
        if (hit) myq[slot] = pid;
    }
    atomicAdd(&s_n[1], generated);
    __syncthreads();
    if (threadIdx.x == 0) { qcount[blockIdx.x] = s_n[0]; gencount[blockIdx.x] = s_n[1]; }
}

// ---------------------------------------------------------------------------------------------
// Persistent-wave BVH traversal with dynamic ray fetch (general scenes).  Lane utilisation of the plain
// one-ray-per-lane loop on a 262 k-triangle scene was 8/64 (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU,
// profiles/r01_pmc_sponza.md): traversal lengths have a heavy tail and internal / leaf phases diverge.  Here a
// wave keeps its lanes busy: finished lanes are re-filled from the workgroup's sub-queue with a wave ballot +
// mbcnt prefix sum and ONE LDS atomic per refill, and every outer iteration runs "all lanes walk internal nodes
// until each holds a leaf (or is done)" followed by "all lanes with a leaf test its triangles" (while-while).
// Exit: a wave leaves when the sub-queue is exhausted and no lane holds a ray — every wave reaches that.
// ---------------------------------------------------------------------------------------------
// Tuning knobs of the persistent traversal (per context: DevScene::refill_min, DevScene::trace_sched).
//   refill_min : refill when at least this many lanes are idle (default 12)
//   trace_sched: 0 = while-while; 1-4 = voted node / triangle steps (vote weights); 5-7 = voted + speculative (lanes with pending
//                triangles keep walking nodes; default 6).  C3 / C5 ms per frame on the 8-wide tree: while-while 67.7 / 56.2,
//                voted (2) 55.6 / 45.3, speculative (6) 53.1 / 44.5; lanes per VALU instruction 24.5 -> 40.8 -> 43.8
//                (profiles/r01_pmc_bvh.md).  With the 128-B 4-wide nodes the voted schedule was slower: the traversal was
//                texture-addresser bound then, not VALU bound.

struct RayLane {                                       // per-lane traversal state
    f3 o, d, idir; float tmin, tmax, bt, bu, bv; uint32_t bprim; uint32_t oct; Grp G; TriGrp T, T2; int sp; uint32_t item; bool has, done;   // T2: second pending triangle group (speculative schedule)
};
__device__ __forceinline__ void ray_begin(RayLane& R, f3 o, f3 d, float tmin, float tmax, uint32_t item) {
    R.o = o; R.d = d; R.tmin = tmin; R.tmax = tmax; R.item = item;
    const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    const float dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    const float dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    R.idir = mk3(__builtin_amdgcn_rcpf(dxs), __builtin_amdgcn_rcpf(dys), __builtin_amdgcn_rcpf(dzs));
    R.oct = ray_octant(R.idir);
    R.bt = tmax; R.bu = 0.0f; R.bv = 0.0f; R.bprim = kMissPrim; R.sp = 0; R.has = true; R.done = false;
    R.G = Grp{0u, (1u << R.oct) | (1u << 8)}; R.T = TriGrp{0u, 0u, 0u}; R.T2 = TriGrp{0u, 0u, 0u};
}
__device__ __forceinline__ void ray_idle(RayLane& R) {
    R.has = false; R.done = false; R.sp = 0; R.item = 0; R.o = mk3(0, 0, 0); R.d = mk3(0, 0, 1); R.idir = mk3(0, 0, 1); R.oct = 0;
    R.tmin = 0.0f; R.tmax = 0.0f; R.bt = 0.0f; R.bu = 0.0f; R.bv = 0.0f; R.bprim = kMissPrim; R.G = Grp{0u, 0u}; R.T = TriGrp{0u, 0u, 0u}; R.T2 = TriGrp{0u, 0u, 0u};
}
// after a node step or a finished triangle group: continue with the node's own internal hits, else pop, else done
template <class STK>
__device__ __forceinline__ void next_group(RayLane& R, STK& stk) {
    if (!(R.G.bits & 0xffu) && !R.T.bits) {
        if (R.sp == 0) R.done = true;
        else { R.sp--; R.G = stk.get(R.sp); }
    }
}
// all lanes with a node group walk down until they hold triangles to test or are done
template <bool ANY, class STK>
__device__ __forceinline__ void walk_internal(const DevScene& sc, const TraceLds& L, RayLane& R, STK& stk) {
    while (R.has && !R.done && !R.T.bits) {
        descend8(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, R.T, stk, R.sp);
        next_group(R, stk);
    }
}
template <bool ANY>
__device__ __forceinline__ void tri_step(const DevScene& sc, const TraceLds& L, RayLane& R) {
    const uint32_t bit = (uint32_t)__builtin_ctz(R.T.bits);
    R.T.bits &= R.T.bits - 1u;
    const uint32_t slot = tri_slot8(R.T, bit);
    v4f v0, e1, e2;
    if (slot < sc.lds_tris) { const lds_v4f* t = L.tris + slot * 3u; v0 = t[0]; e1 = t[1]; e2 = t[2]; }
    else { const v4f* t = (const v4f*)(sc.tris + slot); v0 = t[0]; e1 = t[1]; e2 = t[2]; }
    float t, u, w;
    if (tri_test(R.o, R.d, v0, e1, e2, R.tmin, R.tmax, t, u, w)) {
        if (ANY) { R.bprim = 0u; R.done = true; R.T.bits = 0u; }
        else {
            const uint32_t gid = f2u(v0.w);
            if (t < R.bt || (t == R.bt && gid < R.bprim)) { R.bt = t; R.bu = u; R.bv = w; R.bprim = gid; }
        }
    }
}
template <bool ANY, class STK>
__device__ __forceinline__ void process_leaf(const DevScene& sc, const TraceLds& L, RayLane& R, STK& stk) {
    while (R.has && !R.done && R.T.bits) tri_step<ANY>(sc, L, R);
    if (R.has && !R.done) next_group(R, stk);
}
// Voted schedule (trace_sched 1-4): instead of "walk until EVERY lane holds triangles, then test every lane's triangles"
// each iteration the wave votes for the step most of its busy lanes are waiting for: one node step, or one triangle test.
template <bool ANY, class STK>
__device__ __forceinline__ void voted_step(const DevScene& sc, const TraceLds& L, RayLane& R, STK& stk, uint32_t sched) {
    const bool busy = R.has && !R.done;
    const bool in_tri = busy && R.T.bits != 0u;
    const bool in_node = busy && !in_tri;
    const uint32_t ni = (uint32_t)__popcll(__ballot(in_node)), nl = (uint32_t)__popcll(__ballot(in_tri));
    const uint32_t wn = sched == 3u ? 2u : 1u, wl = sched == 2u ? 2u : sched == 4u ? 3u : 1u;    // experiment: weighted vote
    if (ni * wn >= nl * wl) {
        if (in_node) { descend8(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, R.T, stk, R.sp); next_group(R, stk); }
    } else if (in_tri) {
        tri_step<ANY>(sc, L, R);
        if (!R.done) next_group(R, stk);
    }
}
// Speculative voted schedule (trace_sched 5-7): a lane whose triangles are still waiting for a triangle step keeps walking
// nodes — the triangles of the next node go to a second pending group (T2) — so node steps run with most busy lanes instead of
// only those without pending triangles, and triangle steps run when many lanes have some.  Pending triangles are always tested
// before a ray finishes, and the order of tests does not change the result (minimum over all tested triangles / any hit); what
// speculation costs is culling: node steps taken before the pending triangles shrink the closest distance may visit boxes that
// would have been culled (shadow rays lose nothing: their interval is fixed).
template <bool ANY, class STK>
__device__ __forceinline__ void spec_step(const DevScene& sc, const TraceLds& L, RayLane& R, STK& stk, uint32_t sched) {
    const bool busy = R.has && !R.done;
    const bool has_tri = busy && R.T.bits != 0u;
    const bool can_node = busy && R.T2.bits == 0u && ((R.G.bits & 0xffu) != 0u || R.sp > 0);
    const uint32_t ni = (uint32_t)__popcll(__ballot(can_node)), nl = (uint32_t)__popcll(__ballot(has_tri));
    const uint32_t wn = sched == 7u ? 2u : 1u, wl = sched == 5u ? 1u : sched == 6u ? 2u : 1u;
    if (ni * wn >= nl * wl && ni) {
        if (can_node) {
            if (!(R.G.bits & 0xffu)) { R.sp--; R.G = stk.get(R.sp); }
            TriGrp Tn;
            descend8(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, Tn, stk, R.sp);
            if (Tn.bits) { if (R.T.bits) R.T2 = Tn; else R.T = Tn; }
        }
    } else if (has_tri) {
        tri_step<ANY>(sc, L, R);
        if (!R.T.bits) { R.T = R.T2; R.T2 = TriGrp{0u, 0u, 0u}; }
    }
    if (R.has && !R.done && !(R.G.bits & 0xffu) && R.sp == 0 && !R.T.bits) R.done = true;
}
// wave-level refill: returns false when the wave may exit (queue exhausted and nothing in flight)
template <class Fetch>
__device__ __forceinline__ bool refill(RayLane& R, uint32_t* s_head, uint32_t n, bool& drained, uint32_t refill_min, Fetch fetch) {
    const unsigned long long idle = __ballot(!R.has);
    const uint32_t nidle = (uint32_t)__popcll(idle);
    if (!drained && (nidle >= refill_min || nidle == 64u)) {            // wave-uniform
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(s_head, nidle);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n) drained = true;
        else {
            const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!R.has && idx < n) fetch(idx);
            if (base + nidle >= n) drained = true;
        }
    }
    return __ballot(R.has) != 0ull;
}

// closest hit for every path in this workgroup's sub-queue: reads ray_o/ray_d, writes hit
template <int STK>   // traversal stack: 0 = LDS column, 1 = private (scratch)
__global__ __launch_bounds__(kBlock) void k_trace_closest(DevScene sc, const SmallRecPair* __restrict__ small, DevPaths p, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount, uint32_t qcap, float tmin, uint32_t refill_min, uint32_t sched) {
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_head;
    const uint32_t n = qcount[blockIdx.x];
    if (n == 0) return;
    if (threadIdx.x == 0) s_head = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const uint32_t* myq = queue + (size_t)blockIdx.x * qcap;
    if (sc.nsmall) {                                       // tiny scene, un-fused kernels (test path)
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const uint32_t pid = myq[i];
            const F4 ro = p.ray_o[pid], rd = p.ray_d[pid];
            float t, u, v; uint32_t prim;
            traverse_small<false>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, kTMax, t, u, v, prim, sc.nsmall);
            p.hit[pid] = {t, u, v, u2f(prim)};
        }
        return;
    }
    typename std::conditional<STK == 1, StackPriv, StackLds>::type stk;
    if constexpr (STK != 1) stk.col = L.stack + threadIdx.x;
    RayLane R; ray_idle(R);
    bool drained = false;
    while (refill(R, &s_head, n, drained, refill_min, [&](uint32_t idx) {
               const uint32_t pid = myq[idx];
               const F4 ro = p.ray_o[pid], rd = p.ray_d[pid];
               ray_begin(R, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, kTMax, pid);
           })) {
        if (sched >= 5u) spec_step<false>(sc, L, R, stk, sched);
        else if (sched) voted_step<false>(sc, L, R, stk, sched);
        else { walk_internal<false>(sc, L, R, stk); process_leaf<false>(sc, L, R, stk); }
        if (R.has && R.done) { p.hit[R.item] = {R.bt, R.bu, R.bv, u2f(R.bprim)}; R.has = false; }
    }
}

// any-hit for NEE slot j: visible contributions are added to the path's radiance (a path appears at most once
// per slot, so the read-modify-write needs no atomic and the order of additions per path is fixed)
template <int STK>
__global__ __launch_bounds__(kBlock) void k_trace_shadow(DevScene sc, const SmallRecPair* __restrict__ small, DevPaths p, const F4* __restrict__ sh_o, const F4* __restrict__ sh_d,
                                                         const F4* __restrict__ sh_c, const uint32_t* __restrict__ shcount, uint32_t qcap, uint32_t refill_min, uint32_t sched) {
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_head;
    const uint32_t n = shcount[blockIdx.x];
    if (n == 0) return;
    if (threadIdx.x == 0) s_head = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const size_t qb = (size_t)blockIdx.x * qcap;
    auto finish = [&](uint32_t i, bool occluded) {
        if (!occluded) {
            const F4 c = sh_c[qb + i];
            const uint32_t pid = f2u(c.w);
            F4 r = p.rad[pid];
            r.x = r.x + c.x; r.y = r.y + c.y; r.z = r.z + c.z;
            p.rad[pid] = r;
        }
    };
    if (sc.nsmall) {
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const F4 so = sh_o[qb + i], sd = sh_d[qb + i];
            float t, u, v; uint32_t prim;
            traverse_small<true>(sc, small, L, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, sd.w, t, u, v, prim, sc.nsmall_occ);   // NEE segments only
            finish(i, prim != kMissPrim);
        }
        return;
    }
    typename std::conditional<STK == 1, StackPriv, StackLds>::type stk;
    if constexpr (STK != 1) stk.col = L.stack + threadIdx.x;
    RayLane R; ray_idle(R);
    bool drained = false;
    while (refill(R, &s_head, n, drained, refill_min, [&](uint32_t idx) {
               const F4 so = sh_o[qb + idx], sd = sh_d[qb + idx];
               ray_begin(R, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, sd.w, idx);
           })) {
        if (sched >= 5u) spec_step<true>(sc, L, R, stk, sched);
        else if (sched) voted_step<true>(sc, L, R, stk, sched);
        else { walk_internal<true>(sc, L, R, stk); process_leaf<true>(sc, L, R, stk); }
        if (R.has && R.done) { finish(R.item, R.bprim != kMissPrim); R.has = false; }
    }
}

// ---------------------------------------------------------------------------------------------
// surface reconstruction: ClosestHit, Hit_v6.hlsl:12-61, from the pre-gathered TriShade record
// ---------------------------------------------------------------------------------------------
struct Surf { f3 pos; f3 normal; uint32_t mat; uint32_t inst; float area; f3 flat; };
__device__ __forceinline__ Surf surface(const DevScene& sc, f3 o, f3 d, float t, float u, float v, uint32_t gid) {
    Surf s;
    const F4* rec = (const F4*)(sc.shade + gid);
    const F4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    s.mat = f2u(r0.x); s.inst = f2u(r0.y);
    const f3 flat = mk3(r0.z, r0.w, r1.x);
    const f3 n0 = mk3(r1.y, r1.z, r1.w), n1 = mk3(r2.x, r2.y, r2.z), n2 = mk3(r2.w, r3.x, r3.y);
    s.area = r3.z; s.flat = flat;
    s.pos = mk3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);                 // :15,60
    const float b0 = 1.0f - u - v;                                            // :18
    f3 smooth = mk3(0.0f, 0.0f, 0.0f);
    smooth = smooth + n0 * b0; smooth = smooth + n1 * u; smooth = smooth + n2 * v;   // :40-46
    const f3 n = (length(smooth) > 0.0001f) ? normalize(smooth) : flat;       // :49-54
    s.normal = normalize(xform_dir(sc.insts[s.inst].nrm, n));                 // :56
    return s;
}

// ---------------------------------------------------------------------------------------------
// shading building blocks.  Loop body of RayGen.hlsl:99-133 + Hit.hlsl:126-174,340-369 with the v6 leaf math,
// in the same statement order as oracle/rt_oracle.c:trace_path.  Shared by k_shade (separate trace / shade /
// shadow kernels: general BVH scenes) and k_bounce_small (one fused kernel per bounce: tiny scenes).
//
// Per-path state in HBM (48 B read + 48 B written per bounce):
//   ray_o = (origin.xyz, seed.y bits)   ray_d = (dir.xyz, pdf of the sampled direction)   thr = (throughput.xyz, seed.x bits)
// tmin is a function of the bounce index (camera rays 1e-4, pass1:94; later rays s_bias, Sampler_v6.hlsl:226),
// rad = (radiance.xyz, -) is only touched when something is added.
// ---------------------------------------------------------------------------------------------
struct PathState { uint32_t pid; f3 o, d; float prev_pdf; f3 thr; uint32_t s0, s1; };

__device__ __forceinline__ float bounce_tmin(uint32_t bounce) { return bounce == 0 ? kTMinCam : kSBias; }

__device__ __forceinline__ PathState load_path(const DevPaths& p, uint32_t pid) {
    PathState S; S.pid = pid;
    const F4 ro = p.ray_o[pid], rd = p.ray_d[pid], tv = p.thr[pid];
    S.o = mk3(ro.x, ro.y, ro.z); S.s1 = f2u(ro.w);
    S.d = mk3(rd.x, rd.y, rd.z); S.prev_pdf = rd.w;
    S.thr = mk3(tv.x, tv.y, tv.z); S.s0 = f2u(tv.w);
    return S;
}

// hit on an emissive surface: Hit.hlsl:126-174 with the v6 pdf conventions (Sampler_v6.hlsl:459-465)
__device__ __forceinline__ void add_emissive(const DevScene& sc, const DevPaths& p, const PathState& S, const Surf& sf, const MatGPU& m, uint32_t bounce, uint32_t nee) {
    const f3 Ke = mk3(m.Ke[0], m.Ke[1], m.Ke[2]);
    F4 radv = p.rad[S.pid];
    if (bounce == 0) { radv.x = radv.x + Ke.x; radv.y = radv.y + Ke.y; radv.z = radv.z + Ke.z; }   // Hit.hlsl:128-131
    else {
        float mi = 1.0f;
        if (nee) {                                            // Path_Sampler_v6.hlsl:241
            const f3 Lv = sf.pos - S.o;
            const float dist = length(Lv), dist2 = dist * dist;
            const float cos_t = fabsf(dot(sf.normal, -S.d));
            const float pdf_light = (((Ke.x + Ke.y + Ke.z) / 3.0f) / sc.total_weight) * dist2 / maxf_(cos_t, kEps);
            mi = S.prev_pdf / ((float)nee * pdf_light + S.prev_pdf);
        }
        const f3 e = mk3(Ke.x * S.thr.x * mi, Ke.y * S.thr.y * mi, Ke.z * S.thr.z * mi);   // Hit.hlsl:173
        if (finite3(e)) { radv.x = radv.x + e.x; radv.y = radv.y + e.y; radv.z = radv.z + e.z; }
    }
    p.rad[S.pid] = radv;
}

// one NEE sample: SampleLightNEE_GI, Sampler_v6.hlsl:508-647.  Returns true when a shadow ray is needed.
__device__ __forceinline__ bool nee_sample(const DevScene& sc, const MatGPU& m, uint32_t flags, uint32_t nee, PathState& S, f3 pos, f3 normal, f3 outgoing,
                                           F4& so, F4& sd, f3& con) {
    const float rv = tea_next(S.s0, S.s1);
    int left = 0, right = (int)sc.nlights - 1, sel = 0;
    while (left <= right) {                                   // :523-537
        const int mid = left + (right - left) / 2;
        if (rv < sc.lights[mid].cdf) { sel = mid; right = mid - 1; } else left = mid + 1;
    }
    const LightGPU& lt = sc.lights[sel];
    const f3 xv = mk3(lt.xv[0], lt.xv[1], lt.xv[2]), yv = mk3(lt.yv[0], lt.yv[1], lt.yv[2]), zv = mk3(lt.zv[0], lt.zv[1], lt.zv[2]);
    float xi1 = tea_next(S.s0, S.s1), xi2 = tea_next(S.s0, S.s1);
    if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
    const float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
    const f3 sp = mk3(u * xv.x + v * yv.x + w * zv.x, u * xv.y + v * yv.y + w * zv.y, u * xv.z + v * yv.z + w * zv.z);
    const f3 Lv = sp - pos;
    const float dist2 = dot(Lv, Lv);
    const float dist = sqrtf(maxf_(dist2, kEps));
    const f3 Ln = normalize(Lv);
    f3 nl = mk3(lt.nl[0], lt.nl[1], lt.nl[2]);
    if (dot(nl, -Ln) < 0.0f) nl = -nl;
    const float cos_x = dot(normal, Ln);
    const float cos_y = fabsf(dot(nl, -Ln));
    if (cos_x < kEps || cos_y < kEps) return false;           // :580-585
    const float pdf_light = lt.pdf_l * dist2 / cos_y;         // :629-630
    f3 F; float P, pd, ps; bsdf_mixture(m, flags, normal, Ln, outgoing, F, P, pd, ps);
    const float mi = pdf_light / ((float)nee * pdf_light + P);   // Path_Sampler_v6.hlsl:164
    const float g = cos_x / pdf_light * mi;
    con = mk3(lt.em[0] * (S.thr.x * F.x) * g, lt.em[1] * (S.thr.y * F.y) * g, lt.em[2] * (S.thr.z * F.z) * g);
    if (!finite3(con) || is_zero3(con)) return false;
    const f3 sorg = pos + normalize(normal) * kSBias;         // :616-621
    so = {sorg.x, sorg.y, sorg.z, 0.5f * kSBias};
    sd = {Ln.x, Ln.y, Ln.z, maxf_(kSBias, dist - kSBias * 5.0f)};
    return true;
}

// BSDF sampling + throughput + Russian roulette: Path_Sampler_v6.hlsl:205-229, Sampler_v6.hlsl:423-457,482-497,
// Hit.hlsl:366-369, RayGen.hlsl:118-130.  Returns true when the path continues (state updated in S, smp, P).
__device__ __forceinline__ bool bsdf_continue(const MatGPU& m, const DevFrame& f, uint32_t bounce, PathState& S, f3 normal, f3 outgoing, f3& smp, float& P) {
    const uint32_t st = select_strategy(m, outgoing, normal, f.flags, S.s0, S.s1);
    smp = sample_bsdf(m, st, outgoing, normal, S.s0, S.s1);
    f3 F; float pd, ps; bsdf_mixture(m, f.flags, normal, smp, outgoing, F, P, pd, ps);
    const float NdotL = dot(normal, smp);                     // unclamped, Sampler_v6.hlsl:455
    if (!(P > 0.0f)) return false;
    const float wgt = NdotL / P;                              // Hit.hlsl:366
    S.thr = mk3(S.thr.x * (F.x * wgt), S.thr.y * (F.y * wgt), S.thr.z * (F.z * wgt));
    if (!finite3(S.thr) || is_zero3(S.thr)) return false;
    if (bounce > f.rr_start) {                                // RayGen.hlsl:118-130
        const float mx = maxf_(S.thr.x, maxf_(S.thr.y, S.thr.z));
        const float q = minf_(maxf_(mx, 0.05f), 1.0f);
        const float r = tea_next(S.s0, S.s1);
        if (r > q) return false;
        const float iq = 1.0f / q;
        S.thr = S.thr * iq;
    }
    return true;
}

__device__ __forceinline__ void store_path(const DevPaths& p, const PathState& S, f3 pos, f3 smp, float P) {
    p.ray_o[S.pid] = {pos.x, pos.y, pos.z, u2f(S.s1)};       // un-offset origin, Sampler_v6.hlsl:224-227
    p.ray_d[S.pid] = {smp.x, smp.y, smp.z, P};                // pdf for the MIS at the next emissive hit, Hit.hlsl:369
    p.thr[S.pid] = {S.thr.x, S.thr.y, S.thr.z, u2f(S.s0)};
}

// shade: one thread per queued path (general path: hits come from k_trace_closest, shadow rays go to queues).
// SORT = material-sorted shading: the workgroup's sub-queue is consumed in chunks of kSortChunk entries; each chunk is
// counting-sorted in LDS by the material id of the hit (misses last), so that a wave shades ONE material and its
// branches (emissive / Lambert / GGX strategy, miss) are wave-uniform.  The sort never leaves LDS: the queue index and
// the hit record it reads are needed by the shading anyway.  Results do not depend on the order (per-path state only).
// MEASURED (MI355X, 1080p 16 spp 8 bounces, ms per 2 frames in k_shade): Bistro-class (30 % GGX) 18.1 unsorted vs 23.3
// sorted, Sponza-class 16.1 vs 32.1 — k_shade is HBM-bound, not divergence-bound, and the permutation turns its
// coalesced per-path state streams into gathers; k_trace_shadow gains 4-8 % from the more coherent shadow rays, the
// frame loses 1-10 %.  Hence RTX_OPT_SORT_MATERIALS defaults to 0.
constexpr uint32_t kSortChunk = 2048, kSortKeys = 64;
template <bool SORT>
__global__ __launch_bounds__(kBlock) void k_shade(DevScene sc, DevFrame f, DevPaths p, uint32_t bounce,
                                                  const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount,
                                                  uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_count,
                                                  uint32_t* __restrict__ shcounts /* [nee][gridDim.x] */) {
    __shared__ uint32_t s_cnt[1 + kMaxNee];                 // [0] next-queue length, [1 + j] shadow queue j length
    __shared__ uint32_t s_pid[SORT ? kSortChunk : 1], s_sorted[SORT ? kSortChunk : 1], s_hist[SORT ? kSortKeys : 1];
    __shared__ uint8_t s_key[SORT ? kSortChunk : 1];
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = qcount[blockIdx.x];
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const bool last = (bounce + 1u == f.max_bounces);
    const size_t qb = (size_t)blockIdx.x * f.qcap;
    const uint32_t* myq = queue + qb;
    uint32_t* mynext = next_queue + qb;
    const uint32_t chunk = SORT ? kSortChunk : n;
    for (uint32_t cb = 0; cb < n; cb += chunk) {
        const uint32_t cn = (n - cb < chunk) ? n - cb : chunk;
        if (SORT) {
            if (threadIdx.x < kSortKeys) s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cn; i += kBlock) {
                const uint32_t pid = myq[cb + i];
                const uint32_t prim = f2u(p.hit[pid].w);
                const uint32_t key = prim == kMissPrim ? kSortKeys - 1u : (sc.shade[prim].mat % (kSortKeys - 1u));
                s_pid[i] = pid; s_key[i] = (uint8_t)key;
                atomicAdd(&s_hist[key], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {                             // exclusive scan of the 64 bucket counts by one wave
                const uint32_t c = s_hist[threadIdx.x];
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += t; }
                s_hist[threadIdx.x] = incl - c;
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cn; i += kBlock) s_sorted[atomicAdd(&s_hist[s_key[i]], 1u)] = s_pid[i];
            __syncthreads();
        }
        for (uint32_t base = threadIdx.x & ~63u; base < cn; base += kBlock) {
            const uint32_t i = base + (threadIdx.x & 63u);
            PathState S; S.pid = 0; S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.thr = mk3(0, 0, 0); S.prev_pdf = 1.0f; S.s0 = S.s1 = 0;
            Surf sf; sf.mat = 0; sf.normal = mk3(0, 0, 1); sf.pos = mk3(0, 0, 0);
            bool shading = false;
            if (i < cn) {
                const uint32_t pid = SORT ? s_sorted[i] : myq[cb + i];
                const F4 h = p.hit[pid];
                const uint32_t prim = f2u(h.w);
                if (prim != kMissPrim) {                                          // miss: Miss.hlsl:3-11 -> black, terminate
                    S = load_path(p, pid);
                    sf = surface(sc, S.o, S.d, h.x, h.y, h.z, prim);
                    if (sf.mat < sc.nmat) {
                        const MatGPU& m = sc.mats[sf.mat];
                        if (m.Ke_len > 0.0f) add_emissive(sc, p, S, sf, m, bounce, nee);   // Hit.hlsl:126, Sampler_v6.hlsl:457
                        else shading = true;
                    }
                }
            }
            const f3 outgoing = -S.d, normal = sf.normal, pos = sf.pos;
            const MatGPU* mp = sc.mats + (shading ? sf.mat : 0u);
            for (uint32_t j = 0; j < nee; j++) {                                  // NEE: visibility deferred to k_trace_shadow
                bool push = false;
                F4 so = {0, 0, 0, 0}, sd = {0, 0, 0, 0}; f3 con = mk3(0, 0, 0);
                if (shading) push = nee_sample(sc, *mp, f.flags, nee, S, pos, normal, outgoing, so, sd, con);
                const size_t seg = (size_t)j * f.qcap * gridDim.x + qb;           // NEE slot j, this workgroup's sub-queue
                const uint32_t slot = block_push(push, &s_cnt[1 + j]);
                if (push) { p.sh_o[seg + slot] = so; p.sh_d[seg + slot] = sd; p.sh_c[seg + slot] = {con.x, con.y, con.z, u2f(S.pid)}; }
            }
            bool alive = false;
            f3 smp = mk3(0, 0, 1); float P = 0.0f;
            if (shading && !last) alive = bsdf_continue(*mp, f, bounce, S, normal, outgoing, smp, P);
            if (alive) store_path(p, S, pos, smp, P);
            const uint32_t slot = block_push(alive, &s_cnt[0]);
            if (alive) mynext[slot] = S.pid;
        }
        if (SORT) __syncthreads();                              // the next chunk overwrites the LDS buffers
    }
    __syncthreads();
    if (threadIdx.x == 0) next_count[blockIdx.x] = s_cnt[0];
    if (threadIdx.x >= 1 && threadIdx.x <= nee) shcounts[(size_t)(threadIdx.x - 1) * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}

// Fused bounce kernel for tiny scenes (sc.nsmall != 0): trace the extension ray, shade, trace the NEE shadow
// rays and add their contributions, sample the BSDF, compact — all in one pass over the workgroup's sub-queue.
// Nothing but the 48-B path state and the queue index moves through HBM; hit records and shadow-ray entries
// stay in registers.  Radiance additions happen in the oracle's order (emissive, then NEE slot 0, 1, ...).
template <int WAVES, bool HAVE_HIT>
__global__ __launch_bounds__(kBlock, WAVES) void k_bounce_small(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, DevPaths p, uint32_t bounce,
                                                         const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount,
                                                         uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_count,
                                                         uint32_t* __restrict__ shcounts /* [nee][gridDim.x]: shadow rays traced (statistics) */) {
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_cnt[1 + kMaxNee];
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    const uint32_t n = qcount[blockIdx.x];
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const bool last = (bounce + 1u == f.max_bounces);
    const float tmin = bounce_tmin(bounce);
    const size_t qb = (size_t)blockIdx.x * f.qcap;
    const uint32_t* myq = queue + qb;
    uint32_t* mynext = next_queue + qb;
    for (uint32_t base = threadIdx.x & ~63u; base < n; base += kBlock) {
        const uint32_t i = base + (threadIdx.x & 63u);
        const bool active = i < n;
        PathState S; S.pid = 0; S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.thr = mk3(0, 0, 0); S.prev_pdf = 1.0f; S.s0 = S.s1 = 0;
        if (active) S = load_path(p, myq[i]);
        float t = 0.0f, u = 0.0f, v = 0.0f; uint32_t prim = kMissPrim;
        if (HAVE_HIT) {                                   // bounce 0: the primary hit comes from k_raygen_trace_small
            if (active) { const F4 h = p.hit[S.pid]; t = h.x; u = h.y; v = h.z; prim = f2u(h.w); }
        } else traverse_small<false>(sc, small, L, S.o, S.d, tmin, active ? kTMax : 0.0f, t, u, v, prim, sc.nsmall);   // inactive lanes: empty interval
        Surf sf; sf.mat = 0; sf.normal = mk3(0, 0, 1); sf.pos = mk3(0, 0, 0);
        bool shading = false;
        if (active && prim != kMissPrim) {
            sf = surface(sc, S.o, S.d, t, u, v, prim);
            if (sf.mat < sc.nmat) {
                const MatGPU& m = sc.mats[sf.mat];
                if (m.Ke_len > 0.0f) add_emissive(sc, p, S, sf, m, bounce, nee);
                else shading = true;
            }
        }
        const f3 outgoing = -S.d, normal = sf.normal, pos = sf.pos;
        const MatGPU* mp = sc.mats + (shading ? sf.mat : 0u);
        bool loaded = false; F4 radv = {0, 0, 0, 0};
        for (uint32_t j = 0; j < nee; j++) {
            bool push = false;
            F4 so = {0, 0, 0, 0}, sd = {0, 0, 1, 0}; f3 con = mk3(0, 0, 0);
            if (shading) push = nee_sample(sc, *mp, f.flags, nee, S, pos, normal, outgoing, so, sd, con);
            const unsigned long long pm = __ballot(push);
            if (pm) {                                                          // wave-uniform
                float st_, su_, sv_; uint32_t sprim;
                traverse_small<true>(sc, small, L, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, push ? sd.w : 0.0f, st_, su_, sv_, sprim, sc.nsmall_occ);
                if (push && sprim == kMissPrim) {
                    if (!loaded) { radv = p.rad[S.pid]; loaded = true; }
                    radv.x = radv.x + con.x; radv.y = radv.y + con.y; radv.z = radv.z + con.z;
                }
                if (lane_id() == 0) atomicAdd(&s_cnt[1 + j], (uint32_t)__popcll(pm));
            }
        }
        if (loaded) p.rad[S.pid] = radv;
        bool alive = false;
        f3 smp = mk3(0, 0, 1); float P = 0.0f;
        if (shading && !last) alive = bsdf_continue(*mp, f, bounce, S, normal, outgoing, smp, P);
        if (alive) store_path(p, S, pos, smp, P);
        const uint32_t slot = block_push(alive, &s_cnt[0]);
        if (alive) mynext[slot] = S.pid;
    }
    __syncthreads();
    if (threadIdx.x == 0) next_count[blockIdx.x] = s_cnt[0];
    if (threadIdx.x >= 1 && threadIdx.x <= nee) shcounts[(size_t)(threadIdx.x - 1) * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// The v6 PASS-1 estimator, literally: RayGen_v6_pass1.hlsl:48-190 = primary hit, SampleRIS (Sampler_v6.hlsl:653-736),
// its visibility ray, SamplePathSimple (Path_Sampler_v6.hlsl:3-286), written as ONE kernel with a thread per pixel
// like the reference's raygen shader (this is the reference's own formulation; the wavefront kernels above are the
// product's estimator).  Quirks are kept (abs cosines and unshadowed NEE in the GI loop, two strategy draws per
// bounce, reservoir updates consuming random numbers, half-precision L2/E3/L1, `pdf_light = 1` initial value);
// the only deviations: a miss ends the estimator at that point, frame_seed / sample id are explicit.
// Statement order = oracle/rt_oracle.c:orc_render_v6_pass1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_round_dev(float x) {          // float -> binary16 (RNE) -> float
    const uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;
    if (a >= 0x477FF000u) return u2f(sign | 0x7F800000u);
    if (a < 0x33000001u) return u2f(sign);
    if (a < 0x38800000u) { const float r = rintf(u2f(a) * 16777216.0f); return u2f(sign | f2u(r * (1.0f / 16777216.0f))); }
    const uint32_t rem = a & 0x1FFFu; uint32_t base = a & ~0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}
__device__ __forceinline__ uint32_t half_bits_dev(float x) {
    const float r = half_round_dev(x);
    const uint32_t u = f2u(r), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return sign | 0x7C00u | ((a & 0x007FFFFFu) ? 0x200u : 0u);
    if (a == 0) return sign;
    const int e = (int)(a >> 23) - 127;
    if (e < -14) { const float q = u2f(a) * 16777216.0f; return sign | (uint32_t)q; }
    return sign | (uint32_t)((e + 15) << 10) | ((a >> 13) & 0x3FFu);
}
__device__ __forceinline__ f3 half3_dev(f3 a) { return mk3(half_round_dev(a.x), half_round_dev(a.y), half_round_dev(a.z)); }

struct Res { f3 x2; float w_sum; f3 n2; float W; f3 L2; uint32_t M; };
struct P1Ctx { const DevScene* sc; const SmallRecPair* small; const TraceLds* L; uint32_t flags; uint32_t cnt_ext, cnt_sh; };

__device__ __forceinline__ void lobes_dev(const MatGPU& m, uint32_t flags, f3 normal, f3 L, f3 out_eval, f3 out_pdf, f3& f0, f3& f1, float& q0, float& q1, float& pd, float& ps) {
    strategy_probs(m, out_eval, normal, flags, pd, ps);
    f0 = lambert_eval(m); q0 = lambert_pdf(normal, L);
    if (flags & 1u) { f1 = mk3(0.0f, 0.0f, 0.0f); q1 = 0.0f; }
    else { f1 = ggx_eval(m, normal, L, out_eval); q1 = ggx_pdf(m, normal, L, out_pdf); }
}
struct LSample { f3 sp, Ln, nl; float dist2, dist, pdf_l; f3 em; };
__device__ __forceinline__ LSample light_point_dev(const DevScene& sc, f3 origin, uint32_t& s0, uint32_t& s1) {
    LSample r;
    const float rv = tea_next(s0, s1);
    int left = 0, right = (int)sc.nlights - 1, sel = 0;
    while (left <= right) { const int mid = left + (right - left) / 2; if (rv < sc.lights[mid].cdf) { sel = mid; right = mid - 1; } else left = mid + 1; }
    const LightGPU& lt = sc.lights[sel];
    const f3 xv = mk3(lt.xv[0], lt.xv[1], lt.xv[2]), yv = mk3(lt.yv[0], lt.yv[1], lt.yv[2]), zv = mk3(lt.zv[0], lt.zv[1], lt.zv[2]);
    float xi1 = tea_next(s0, s1), xi2 = tea_next(s0, s1);
    if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
    const float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
    r.sp = mk3(u * xv.x + v * yv.x + w * zv.x, u * xv.y + v * yv.y + w * zv.y, u * xv.z + v * yv.z + w * zv.z);
    const f3 Lv = r.sp - origin;
    r.dist2 = dot(Lv, Lv); r.dist = sqrtf(maxf_(r.dist2, kEps)); r.Ln = normalize(Lv);
    r.nl = mk3(lt.nl[0], lt.nl[1], lt.nl[2]);
    if (dot(r.nl, -r.Ln) < 0.0f) r.nl = -r.nl;
    r.pdf_l = lt.pdf_l;      // already max(EPS, weight / max(area, EPS)); the callers' max(EPS, .) is idempotent
    r.em = mk3(lt.em[0], lt.em[1], lt.em[2]);
    return r;
}
__device__ __forceinline__ bool res_update_dev(Res& r, float wi, f3 x, f3 n, f3 L, uint32_t& s0, uint32_t& s1) {
    r.w_sum += wi;
    if (tea_next(s0, s1) < wi / r.w_sum) { r.x2 = x; r.n2 = n; r.L2 = half3_dev(L); return true; }
    return false;
}
__device__ __forceinline__ f3 reconnect_di_dev(const MatGPU& m, uint32_t flags, f3 x1, f3 n1, f3 x2, f3 n2, f3 L, f3 outgoing) {
    const f3 dir = x2 - x1;
    const float dist = length(dir);
    const float cos1 = maxf_(0.0f, dot(n1, normalize(dir)));
    if (dot(n2, normalize(-dir)) < 0.0f) n2 = -n2;
    const float cos2 = maxf_(0.0f, dot(n2, normalize(-dir)));
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(m, flags, n1, normalize(dir), normalize(outgoing), normalize(outgoing), f0, f1, q0, q1, pd, ps);
    const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
    const float d2 = dist * dist;
    return mk3(F.x * L.x * cos1 * cos2 / d2, F.y * L.y * cos1 * cos2 / d2, F.z * L.z * cos1 * cos2 / d2);
}
__device__ __forceinline__ bool p1_any(P1Ctx& C, f3 o, f3 d, float tmin, float tmax) {
    float t, u, v; uint32_t prim;
    trace_ray<true>(*C.sc, C.small, *C.L, o, d, tmin, tmax, t, u, v, prim);
    C.cnt_sh++;
    return prim != kMissPrim;
}
__device__ __forceinline__ bool p1_hit(P1Ctx& C, f3 o, f3 d, float tmin, Surf& sf) {
    float t, u, v; uint32_t prim;
    trace_ray<false>(*C.sc, C.small, *C.L, o, d, tmin, kTMax, t, u, v, prim);
    if (prim == kMissPrim) return false;
    sf = surface(*C.sc, o, d, t, u, v, prim);
    return sf.mat < C.sc->nmat;
}

__device__ void sample_ris_dev(P1Ctx& C, uint32_t M1, uint32_t M2, f3 outgoing, Res& rs, const Surf& pay, uint32_t& s0, uint32_t& s1) {
    const DevScene& sc = *C.sc; const uint32_t flags = C.flags;
    const MatGPU& m = sc.mats[pay.mat];
    const uint32_t strategy = select_strategy(m, outgoing, pay.normal, flags, s0, s1);
    const f3 origin = pay.pos, normal = pay.normal;
    for (uint32_t i = 0; i < M1 && sc.nlights; i++) {
        const LSample ls = light_point_dev(sc, origin, s0, s1);
        const float cos_x = dot(normal, ls.Ln), cos_y = dot(ls.nl, -ls.Ln);
        const float G = maxf_(cos_y * cos_x / ls.dist2, kEps);
        f3 f0, f1; float q0, q1, pd, ps;
        lobes_dev(m, flags, normal, ls.Ln, normalize(outgoing), normalize(outgoing), f0, f1, q0, q1, pd, ps);
        const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
        const float P = safe_mul(pd, q0 * cos_y / ls.dist2) + safe_mul(ps, q1 * cos_y / ls.dist2);
        const float p_hat = length(mk3(ls.em.x * F.x * G * 1.0f, ls.em.y * F.y * G * 1.0f, ls.em.z * F.z * G * 1.0f));
        const float pdf_light = maxf_(kEps, ls.pdf_l);
        const float mi = pdf_light / ((float)M1 * pdf_light + (float)M2 * P);
        const float wi = mi * p_hat / pdf_light;
        if (p_hat > 0.0f) res_update_dev(rs, wi, ls.sp, ls.nl, ls.em, s0, s1);
    }
    for (uint32_t j = 0; j < M2; j++) {
        float pdf_light = 0.0f, pdf_bsdf = 0.0f, p_hat = 0.0f;
        f3 em = mk3(0, 0, 0), x2 = mk3(0, 0, 0), n2 = mk3(0, 0, 0);
        const f3 smp = sample_bsdf(m, strategy, outgoing, normal, s0, s1);
        Surf h2;
        C.cnt_ext++;
        if (p1_hit(C, origin, smp, kSBias, h2)) {
            const MatGPU& mk = sc.mats[h2.mat];
            const float Ke = mk.KeFull[0] + mk.KeFull[1] + mk.KeFull[2];
            em = mk3(mk.KeFull[0], mk.KeFull[1], mk.KeFull[2]); x2 = h2.pos; n2 = h2.normal;
            if (Ke > kEps && sc.nlights) {
                const float dist = length(h2.pos - origin), dist2 = dist * dist;
                const float cos_t = dot(h2.normal, -smp);
                pdf_light = (Ke / 3.0f) / sc.total_weight;
                f3 f0, f1; float q0, q1, pd, ps;
                lobes_dev(m, flags, normal, smp, normalize(outgoing), outgoing, f0, f1, q0, q1, pd, ps);
                const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
                pdf_bsdf = safe_mul(pd, q0 * cos_t / dist2) + safe_mul(ps, q1 * cos_t / dist2);
                const float ndot = dot(normal, smp);
                p_hat = length(mk3(F.x * em.x * ndot * cos_t / dist2, F.y * em.y * ndot * cos_t / dist2, F.z * em.z * ndot * cos_t / dist2));
            }
        }
        const float mi = pdf_bsdf / ((float)M1 * pdf_light + (float)M2 * pdf_bsdf);
        const float wi = mi * p_hat / pdf_bsdf;
        if (p_hat > 0.0f) res_update_dev(rs, wi, x2, n2, em, s0, s1);
    }
    rs.M = 1;
}

__device__ f3 sample_path_simple_dev(P1Ctx& C, const DevFrame& f, Res& rs, f3 init_point, f3 init_normal, f3 init_outgoing, uint32_t init_mat, uint32_t& s0, uint32_t& s1) {
    const DevScene& sc = *C.sc; const uint32_t flags = C.flags;
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    f3 acc_f = mk3(1, 1, 1), acc_f_rec = mk3(1, 1, 1), acc_L = mk3(0, 0, 0);
    float acc_pdf = 1.0f;
    f3 x1s = mk3(0, 0, 0), x2s = mk3(0, 0, 0);
    f3 origin = init_point, normal = init_normal, outgoing = normalize(init_outgoing);
    uint32_t mat = init_mat;
    {
        const uint32_t st = select_strategy(sc.mats[mat], outgoing, normal, flags, s0, s1);
        const f3 smp = sample_bsdf(sc.mats[mat], st, outgoing, normal, s0, s1);
        Surf h;
        C.cnt_ext++;
        if (!p1_hit(C, origin, smp, kSBias, h)) return mk3(0, 0, 0);
        if (sc.mats[h.mat].KeFullLen > 0.0f) return mk3(0, 0, 0);
        const f3 incoming = normalize(-smp);
        f3 f0, f1; float q0, q1, pd, ps;
        lobes_dev(sc.mats[mat], flags, normal, -incoming, outgoing, outgoing, f0, f1, q0, q1, pd, ps);
        const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
        const float P = safe_mul(pd, q0) + safe_mul(ps, q1);
        const float NdotL = dot(normal, smp);
        acc_pdf *= P;
        acc_f = mk3(acc_f.x * (F.x * NdotL), acc_f.y * (F.y * NdotL), acc_f.z * (F.z * NdotL));
        outgoing = incoming; mat = h.mat; normal = h.normal; origin = h.pos;
    }
    const f3 xn = origin, nn = normalize(normal);
    for (uint32_t i = 0; i < f.max_bounces; i++) {
        (void)select_strategy(sc.mats[mat], outgoing, normal, flags, s0, s1);
        for (uint32_t j = 0; j < nee; j++) {
            const LSample ls = light_point_dev(sc, origin, s0, s1);
            float cos_x = fabsf(dot(normal, ls.Ln)); if (cos_x < kEps) cos_x = 0.0f;
            float cos_y = fabsf(dot(ls.nl, -ls.Ln)); if (cos_y < kEps) cos_y = 0.0f;
            f3 f0, f1; float q0, q1, pd, ps;
            lobes_dev(sc.mats[mat], flags, normal, ls.Ln, normalize(outgoing), normalize(outgoing), f0, f1, q0, q1, pd, ps);
            const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
            const float pdf_bsdf = safe_mul(pd, q0) + safe_mul(ps, q1);
            float pdf_light = 1.0f;
            if (cos_y > 0.0f) pdf_light = maxf_(kEps, ls.pdf_l) * ls.dist2 / cos_y;
            const float a_pdf = acc_pdf * pdf_light;
            const f3 thr = mk3(F.x * cos_x * 1.0f, F.y * cos_x * 1.0f, F.z * cos_x * 1.0f);
            const f3 a_l = mk3(acc_f.x * thr.x, acc_f.y * thr.y, acc_f.z * thr.z);
            const f3 contribution = a_pdf > 0.0f ? mk3(ls.em.x * a_l.x / a_pdf, ls.em.y * a_l.y / a_pdf, ls.em.z * a_l.z / a_pdf) : mk3(0, 0, 0);
            const float mi = pdf_light / ((float)nee * pdf_light + pdf_bsdf);
            const f3 E_rec = mk3(acc_f_rec.x * mi * ls.em.x * thr.x, acc_f_rec.y * mi * ls.em.y * thr.y, acc_f_rec.z * mi * ls.em.z * thr.z);
            const f3 E_path = contribution * mi;
            float wi = length(E_path);
            acc_L = acc_L + E_path;
            if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
            if (res_update_dev(rs, wi, xn, normalize(nn), E_rec, s0, s1)) { x1s = origin + normalize(normal) * kSBias; x2s = ls.sp; }
        }
        const uint32_t st = select_strategy(sc.mats[mat], outgoing, normal, flags, s0, s1);
        const f3 smp = sample_bsdf(sc.mats[mat], st, outgoing, normal, s0, s1);
        Surf h;
        C.cnt_ext++;
        if (!p1_hit(C, origin, smp, kSBias, h)) break;
        f3 f0, f1; float q0, q1, pd, ps;
        lobes_dev(sc.mats[mat], flags, normal, smp, normalize(outgoing), outgoing, f0, f1, q0, q1, pd, ps);
        const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
        const float pdf_bsdf = safe_mul(pd, q0) + safe_mul(ps, q1);
        const float NdotL = dot(normal, smp);
        const MatGPU& mk = sc.mats[h.mat];
        const f3 thr = mk3(F.x * NdotL, F.y * NdotL, F.z * NdotL);
        acc_pdf *= pdf_bsdf;
        acc_f = mk3(acc_f.x * thr.x, acc_f.y * thr.y, acc_f.z * thr.z);
        acc_f_rec = mk3(acc_f_rec.x * thr.x, acc_f_rec.y * thr.y, acc_f_rec.z * thr.z);
        if (mk.Ke_len > 0.0f) {
            const float dist = length(h.pos - origin), dist2 = dist * dist;
            const float cos_t = dot(h.normal, -smp);
            const float pdf_light = sc.nlights ? (((mk.Ke[0] + mk.Ke[1] + mk.Ke[2]) / 3.0f) / sc.total_weight) * dist2 / cos_t : 0.0f;
            const f3 contribution = mk3(mk.Ke[0] * acc_f.x / acc_pdf, mk.Ke[1] * acc_f.y / acc_pdf, mk.Ke[2] * acc_f.z / acc_pdf);
            if (length(contribution) > 0.0f) {
                const float mi = pdf_bsdf / ((float)nee * pdf_light + pdf_bsdf);
                const f3 E_rec = mk3(acc_f_rec.x * mi * mk.Ke[0], acc_f_rec.y * mi * mk.Ke[1], acc_f_rec.z * mi * mk.Ke[2]);
                const f3 E_path = contribution * mi;
                float wi = length(E_path);
                acc_L = acc_L + E_path;
                if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
                res_update_dev(rs, wi, xn, normalize(nn), E_rec, s0, s1);
                break;
            }
        }
        origin = h.pos; mat = h.mat; outgoing = -smp; normal = h.normal;
    }
    if (nee > 0 && length(x2s - x1s) > kEps) {
        const f3 dv = x2s - x1s;
        if (p1_any(C, x1s, normalize(dv), 0.5f * kSBias, maxf_(kSBias, length(dv) - kSBias * 5.0f))) rs.w_sum *= 0.0f;
        else rs.w_sum *= 1.0f;
    }
    return acc_L;
}

__device__ __forceinline__ uint32_t map_pixel_id(uint32_t w, uint32_t x, uint32_t y) {    // Common_v6.hlsl:173-198
    const uint32_t tcx = (w + 3u) >> 2;
    return ((y >> 2) * tcx + (x >> 2)) * 16u + (y & 3u) * 4u + (x & 3u);
}
__device__ __forceinline__ void store_res(uint32_t* dst, const Res& r) {                  // 40 bytes = 10 dwords
    dst[0] = f2u(r.x2.x); dst[1] = f2u(r.x2.y); dst[2] = f2u(r.x2.z); dst[3] = f2u(r.w_sum);
    dst[4] = f2u(r.n2.x); dst[5] = f2u(r.n2.y); dst[6] = f2u(r.n2.z); dst[7] = f2u(r.W);
    dst[8] = half_bits_dev(r.L2.x) | (half_bits_dev(r.L2.y) << 16); dst[9] = half_bits_dev(r.L2.z) | ((r.M & 0xFFFFu) << 16);
}

__global__ __launch_bounds__(kBlock) void k_v6_pass1(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, uint32_t sample_id,
                                                     F4* __restrict__ accum, uint32_t* __restrict__ res_di, uint32_t* __restrict__ res_gi, uint32_t* __restrict__ sdata,
                                                     unsigned long long* __restrict__ counters /* primary, extension, shadow */) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_prim = 0, n_ext = 0, n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        uint32_t s0, s1; seed_init(x, y, sample_id, f.frame_seed, s0, s1);
        f3 origin, dir; primary_ray(cam, f.width, f.height, x, y, 0.0f, 0.0f, origin, dir);      // jitter = 0, pass1:80-82
        Res rdi; rdi.x2 = mk3(0, 0, 0); rdi.w_sum = 0.0f; rdi.n2 = mk3(0, 0, 0); rdi.W = 0.0f; rdi.L2 = mk3(0, 0, 0); rdi.M = 0;
        Res rgi = rdi;
        f3 x1 = mk3(0, 0, 0), n1 = mk3(0, 0, 0), ov = mk3(0, 0, 0), debug = mk3(0, 0, 0), L1 = mk3(0, 0, 0), out = mk3(0, 0, 0);
        uint32_t mID = kMissMat, objID = 0;
        P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
        Surf pay;
        n_prim++;
        if (p1_hit(C, origin, dir, kTMinCam, pay)) {
            mID = pay.mat; objID = pay.inst;
            const MatGPU& m = sc.mats[mID];
            L1 = mk3(m.Ke[0], m.Ke[1], m.Ke[2]);
            if (!(m.KeFullLen > 0.0f)) {                                                         // performSampling, pass1:102-106
                const f3 outgoing = -dir;
                sample_ris_dev(C, sc.nlights ? f.nee_samples : 0u, 1u, outgoing, rdi, pay, s0, s1);
                x1 = pay.pos; n1 = normalize(pay.normal); ov = outgoing;
                const float f_g = length(reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov));
                const f3 dv = rdi.x2 - x1;
                const float vis = p1_any(C, x1 + normalize(n1) * kSBias, normalize(dv), 0.0f, maxf_(length(dv) - 10.0f * kSBias, 2.0f * kSBias)) ? 0.0f : 1.0f;
                const float p_hat = f_g * vis;
                rdi.W = p_hat > kEps ? rdi.w_sum / p_hat : 0.0f;
                debug = sample_path_simple_dev(C, f, rgi, pay.pos, pay.normal, outgoing, mID, s0, s1);
                const f3 rc = reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov);
                debug = debug + rc * rdi.W;
                {
                    const f3 dg = rgi.x2 - x1;
                    const float cos1 = fabsf(dot(n1, normalize(dg)));
                    f3 f0, f1; float q0, q1, pd, ps;
                    lobes_dev(m, f.flags, n1, normalize(dg), normalize(ov), normalize(ov), f0, f1, q0, q1, pd, ps);
                    const f3 Fx = safe_mul(pd, f0) + safe_mul(ps, f1);
                    f3 fr = mk3(Fx.x * cos1 * rgi.L2.x, Fx.y * cos1 * rgi.L2.y, Fx.z * cos1 * rgi.L2.z);
                    if (!finite3(fr)) fr = mk3(0, 0, 0);
                    const float fc = length(fr);
                    rgi.W = fc > kEps ? rgi.w_sum / fc : 0.0f;
                    rgi.M = 1;
                }
                out = debug;
            } else out = L1;
        }
        n_ext += C.cnt_ext; n_sh += C.cnt_sh;
        const size_t slot = map_pixel_id(f.width, x, y);
        store_res(res_di + slot * 10, rdi);
        store_res(res_gi + slot * 10, rgi);
        uint32_t* d = sdata + slot * 15;                                                        // 60 bytes: Reservoir_v6.hlsl:2-11
        d[0] = f2u(x1.x); d[1] = f2u(x1.y); d[2] = f2u(x1.z);
        d[3] = (mID & 0xFFFFu) | (half_bits_dev(L1.x) << 16); d[4] = half_bits_dev(L1.y) | (half_bits_dev(L1.z) << 16);
        d[5] = f2u(n1.x); d[6] = f2u(n1.y); d[7] = f2u(n1.z); d[8] = f2u(ov.x); d[9] = f2u(ov.y); d[10] = f2u(ov.z);
        d[11] = objID; d[12] = f2u(debug.x); d[13] = f2u(debug.y); d[14] = f2u(debug.z);
        if (finite3(out)) { F4 a = accum[(size_t)y * f.width + x]; a.x = a.x + out.x; a.y = a.y + out.y; a.z = a.z + out.z; a.w = a.w + 1.0f; accum[(size_t)y * f.width + x] = a; }
    }
    atomicAdd(&counters[0], (unsigned long long)n_prim); atomicAdd(&counters[1], (unsigned long long)n_ext); atomicAdd(&counters[2], (unsigned long long)n_sh);
}

// ---------------------------------------------------------------------------------------------
// ReSTIR temporal reuse (pass 2, RayGen_v6_pass2.hlsl:46-204) and spatial reuse + final shade (pass 3,
// RayGen_v6_pass3.hlsl:46-441) with the pairwise MIS of MIS_v6.hlsl / MIS_GI_v6.hlsl, on the reference's packed
// buffers; one thread per pixel like the reference's raygen shaders.  Statement order = oracle/rt_oracle.c.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_to_float_dev(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) { const float f = (float)m * (1.0f / 16777216.0f); return u2f(f2u(f) | sign); }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
struct SData { f3 x1; uint32_t mID; f3 L1; f3 n1; f3 o; uint32_t objID; };
__device__ __forceinline__ Res load_res_dev(const uint32_t* p) {
    Res r;
    r.x2 = mk3(u2f(p[0]), u2f(p[1]), u2f(p[2])); r.w_sum = u2f(p[3]); r.n2 = mk3(u2f(p[4]), u2f(p[5]), u2f(p[6])); r.W = u2f(p[7]);
    r.L2 = mk3(half_to_float_dev(p[8] & 0xFFFFu), half_to_float_dev(p[8] >> 16), half_to_float_dev(p[9] & 0xFFFFu)); r.M = p[9] >> 16;
    return r;
}
__device__ __forceinline__ Res zero_res() { Res r; r.x2 = mk3(0, 0, 0); r.w_sum = 0.0f; r.n2 = mk3(0, 0, 0); r.W = 0.0f; r.L2 = mk3(0, 0, 0); r.M = 0; return r; }
__device__ __forceinline__ SData load_sd_dev(const uint32_t* d) {
    SData s;
    s.x1 = mk3(u2f(d[0]), u2f(d[1]), u2f(d[2])); s.mID = d[3] & 0xFFFFu;
    s.L1 = mk3(half_to_float_dev(d[3] >> 16), half_to_float_dev(d[4] & 0xFFFFu), half_to_float_dev(d[4] >> 16));
    s.n1 = mk3(u2f(d[5]), u2f(d[6]), u2f(d[7])); s.o = mk3(u2f(d[8]), u2f(d[9]), u2f(d[10])); s.objID = d[11];
    return s;
}
__device__ __forceinline__ SData zero_sd() { SData s; s.x1 = mk3(0, 0, 0); s.mID = 0; s.L1 = mk3(0, 0, 0); s.n1 = mk3(0, 0, 0); s.o = mk3(0, 0, 0); s.objID = 0; return s; }
__device__ __forceinline__ float minf_u(float cap, uint32_t m) { return (float)(m < (uint32_t)cap ? m : (uint32_t)cap); }

__device__ __forceinline__ float get_p_hat_dev(P1Ctx& C, const MatGPU& m, f3 x1, f3 n1, f3 x2, f3 n2, f3 L2, f3 o, bool vis) {
    const float f_g = length(reconnect_di_dev(m, C.flags, x1, n1, x2, n2, L2, o));
    float v = 1.0f;
    if (vis) { const f3 dv = x2 - x1; v = p1_any(C, x1 + normalize(n1) * kSBias, normalize(dv), 0.0f, maxf_(length(dv) - 10.0f * kSBias, 2.0f * kSBias)) ? 0.0f : 1.0f; }
    return f_g * v;
}
__device__ __forceinline__ f3 get_p_hat_gi_dev(P1Ctx& C, const MatGPU& m, f3 x1, f3 n1, f3 x2, f3 L, f3 o, bool vis) {
    const f3 dir = x2 - x1;
    const float cos1 = fabsf(dot(n1, normalize(dir)));
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(m, C.flags, n1, normalize(dir), normalize(o), normalize(o), f0, f1, q0, q1, pd, ps);
    const f3 Fx = safe_mul(pd, f0) + safe_mul(ps, f1);
    f3 fr = mk3(Fx.x * cos1 * L.x, Fx.y * cos1 * L.y, Fx.z * cos1 * L.z);
    if (!finite3(fr)) fr = mk3(0, 0, 0);
    float v = 1.0f;
    if (vis) v = p1_any(C, x1 + normalize(n1) * kSBias, normalize(dir), 0.0f, maxf_(length(dir) - 10.0f * kSBias, 2.0f * kSBias)) ? 0.0f : 1.0f;
    return fr * v;
}
__device__ __forceinline__ float get_w_dev(float w_sum, float p_hat) { return p_hat > kEps ? w_sum / p_hat : 0.0f; }
__device__ __forceinline__ float jacobian_dev(const SData& r, const SData& q, f3 x2q, f3 n2q) {
    const f3 vq = x2q - q.x1, vr = x2q - r.x1;
    const float cq = fabsf(dot(normalize(-vq), normalize(n2q))), cr = fabsf(dot(normalize(-vr), normalize(n2q)));
    return (cq / cr) * (dot(vr, vr) / dot(vq, vq));
}
__device__ __forceinline__ bool valid_res_dev(const Res& r) { return length(r.n2) > 0.0f && length(r.L2) > 0.0f && r.w_sum > 0.0f && r.M > 0; }
__device__ __forceinline__ bool valid_res_gi_dev(const Res& r) { return r.w_sum > 0.0f && r.M > 0; }
__device__ __forceinline__ bool reject_distance_dev(f3 x1, f3 x2, f3 cam, float thr) {
    const float d1 = length(x1 - cam), d2 = length(x2 - cam);
    return fabsf(d1 - d2) / maxf_(d1, d2) > thr;
}
__device__ __forceinline__ bool reject_jacobian_dev(float J, float thr) { return J > thr || J < 1.0f / thr || is_nan(J) || is_inf(J); }
__device__ __forceinline__ f3 mul44_dev(const float* m, f3 p, float w, float& ow) {
    ow = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * w;
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * w, m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * w, m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * w);
}
__device__ __forceinline__ void random_pixel_dev(uint32_t radius, uint32_t w, uint32_t h, uint32_t x, uint32_t y, uint32_t& s0, uint32_t& s1, int& ox, int& oy) {
    int nx, ny;
    do {
        const float u = tea_next(s0, s1);
        const float r = (float)radius * u;
        const float ang = tea_next(s0, s1) * 6.2831853f;
        float sn, cs; sincos_(ang, sn, cs);
        nx = (int)x + (int)(cs * r); ny = (int)y + (int)(sn * r);
        while (nx < 0 || nx >= (int)w) { if (nx < 0) nx = -nx; else nx = 2 * (int)w - nx - 2; }
        while (ny < 0 || ny >= (int)h) { if (ny < 0) ny = -ny; else ny = 2 * (int)h - ny - 2; }
    } while (nx == (int)x && ny == (int)y);
    ox = nx; oy = ny;
}

struct RestirBufs { uint32_t *cur_di, *cur_gi, *cur_sd, *last_di, *last_gi, *last_sd; };

__global__ __launch_bounds__(kBlock) void k_restir_pass2(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, RestirBufs B,
                                                         unsigned long long* __restrict__ counters) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        const size_t slot = map_pixel_id(f.width, x, y);
        Res rc = load_res_dev(B.cur_di + slot * 10), gc = load_res_dev(B.cur_gi + slot * 10);
        const SData sd = load_sd_dev(B.cur_sd + slot * 15);
        if (!(sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f) || sd.mID == 0xFFFEu || sd.mID >= sc.nmat) continue;
        P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
        const f3 camo = mk3(cam.viewI[12], cam.viewI[13], cam.viewI[14]);
        uint32_t s0, s1; seed_init(x, y, 2u, f.frame_seed, s0, s1);
        int px, py;
        {   // GetBestReprojectedPixel_d, Sampler_v6.hlsl:738-785
            float w0, w1, w2, w3;
            const InstGPU& in = sc.insts[sd.objID < sc.ninst ? sd.objID : 0u];
            const f3 lp = mul44_dev(in.o2w_inv, sd.x1, 1.0f, w0);
            const f3 pw = mul44_dev(in.prev_o2w, lp, w0, w1);
            const f3 vp = mul44_dev(cam.prev_view, pw, w1, w2);
            const f3 cp = mul44_dev(cam.prev_proj, vp, w2, w3);
            if (w3 <= 0.0f) { px = -1; py = -1; }
            else { const float ux = (cp.x / w3) * 0.5f + 0.5f; float uy = (cp.y / w3) * 0.5f + 0.5f; uy = 1.0f - uy; px = (int)rintf(ux * (float)f.width); py = (int)rintf(uy * (float)f.height); }
        }
        const bool inside = px >= 0 && py >= 0 && px < (int)f.width && py < (int)f.height;
        const size_t ts = inside ? map_pixel_id(f.width, (uint32_t)px, (uint32_t)py) : 0;
        const Res rl = inside ? load_res_dev(B.last_di + ts * 10) : zero_res(), gl = inside ? load_res_dev(B.last_gi + ts * 10) : zero_res();
        const SData sl = inside ? load_sd_dev(B.last_sd + ts * 15) : zero_sd();
        const bool base_ok = (px != -1 && py != -1) && length(sl.L1) == 0.0f && !reject_distance_dev(sd.x1, sl.x1, camo, 0.1f) && sl.mID == sd.mID;
        const bool acc_di = base_ok && valid_res_dev(rl) && (rl.x2.x != 0.0f && rl.x2.y != 0.0f && rl.x2.z != 0.0f);
        const bool acc_gi = base_ok && !(gl.w_sum > 5.0f) && valid_res_gi_dev(gl);
        const MatGPU& m = sc.mats[sd.mID];
        if (acc_di) {
            const float mc = minf_u(16.0f, rc.M), ml = minf_u(16.0f, rl.M), M_sum = mc + ml;
            float mi_c = mc / M_sum;
            { const float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
            float mi_t;
            { const float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
            if (length(rl.n2) == 0.0f) { mi_c = 1.0f; mi_t = 0.0f; }
            const float w_c = mi_c * get_p_hat_dev(C, m, sd.x1, sd.n1, rc.x2, rc.n2, rc.L2, sd.o, false) * rc.W;
            const float w_t = mi_t * get_p_hat_dev(C, m, sd.x1, sd.n1, rl.x2, rl.n2, rl.L2, sd.o, true) * rl.W;
            rc.M = (uint32_t)mc; rc.w_sum = w_c;
            rc.w_sum += w_t; rc.M = (rc.M + (uint32_t)ml) & 0xFFFFu;
            if (tea_next(s0, s1) < w_t / rc.w_sum) { rc.x2 = rl.x2; rc.n2 = rl.n2; rc.L2 = rl.L2; }
            const float p_hat = get_p_hat_dev(C, m, sd.x1, sd.n1, rc.x2, rc.n2, rc.L2, sd.o, false);
            rc.W = get_w_dev(rc.w_sum, p_hat);
        }
        if (acc_gi) {
            const float mc = minf_u(16.0f, gc.M), ml = minf_u(16.0f, gl.M), M_sum = mc + ml;
            float mi_c = mc / M_sum;
            { const float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
            float mi_t;
            { const float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
            const float w_c = mi_c * length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, gc.x2, gc.L2, sd.o, false)) * gc.W;
            const float w_t = mi_t * length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, gl.x2, gl.L2, sd.o, true)) * gl.W;
            gc.M = (uint32_t)mc; gc.w_sum = w_c;
            gc.w_sum += w_t; gc.M = (gc.M + (uint32_t)ml) & 0xFFFFu;
            if (tea_next(s0, s1) < w_t / gc.w_sum) { gc.x2 = gl.x2; gc.n2 = gl.n2; gc.L2 = gl.L2; }
            gc.W = get_w_dev(gc.w_sum, length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, gc.x2, gc.L2, sd.o, false)));
        }
        store_res(B.cur_di + slot * 10, rc); store_res(B.cur_gi + slot * 10, gc);
        n_sh += C.cnt_sh;
    }
    atomicAdd(&counters[2], (unsigned long long)n_sh);
}

__global__ __launch_bounds__(kBlock) void k_restir_pass3(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, RestirBufs B,
                                                         F4* __restrict__ accum, unsigned long long* __restrict__ counters) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t W = f.width, H = f.height;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        const size_t slot = map_pixel_id(W, x, y);
        const SData sd = load_sd_dev(B.cur_sd + slot * 15);
        f3 out = mk3(0, 0, 0);
        if (!(sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f)) out = sd.L1;                      // pass3:457-462
        else if (!(sd.mID == 0xFFFEu || sd.mID >= sc.nmat)) {
            P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
            const f3 camo = mk3(cam.viewI[12], cam.viewI[13], cam.viewI[14]);
            uint32_t s0, s1; seed_init(x, y, 3u, f.frame_seed, s0, s1);
            const MatGPU& m = sc.mats[sd.mID];
            Res rcur = load_res_dev(B.cur_di + slot * 10), gcur = load_res_dev(B.cur_gi + slot * 10);
            size_t cand_di[3], cand_gi[3]; int n_di = 0, n_gi = 0;
            float M_sum_DI = minf_u(128.0f, rcur.M), M_sum_GI = minf_u(128.0f, gcur.M);
            for (int a = 0; a < 9 && n_di < 3; a++) {
                int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
                const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
                const SData sn = load_sd_dev(B.cur_sd + pr * 15); const Res rn = load_res_dev(B.cur_di + pr * 10);
                const bool ok = !(dot(sd.n1, sn.n1) < 0.9f) && !reject_distance_dev(sd.x1, sn.x1, camo, 0.1f) && valid_res_dev(rn) && length(sn.L1) == 0.0f && sn.mID == sd.mID;
                if (ok) { cand_di[n_di++] = pr; M_sum_DI += minf_u(128.0f, rn.M); }
            }
            for (int a = 0; a < 9 && n_gi < 3; a++) {
                int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
                const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
                const SData sn = load_sd_dev(B.cur_sd + pr * 15); const Res gn = load_res_dev(B.cur_gi + pr * 10);
                const bool ok = m.Pr > 0.3f && !reject_distance_dev(sd.x1, sn.x1, camo, 0.1f) && !(dot(normalize(gn.x2 - sd.x1), sd.n1) < 0.0f) &&
                                !(gn.w_sum > 5.0f) && valid_res_gi_dev(gn) && !reject_jacobian_dev(jacobian_dev(sn, sd, gn.x2, gn.n2), 5.0f) &&
                                length(sn.L1) == 0.0f && sn.mID == sd.mID;
                if (ok) { cand_gi[n_gi++] = pr; M_sum_GI += minf_u(128.0f, gn.M); }
            }
            const Res can = rcur, can_gi = gcur;
            const float cMmin = minf_u(128.0f, can.M), cMmax = M_sum_DI - cMmin;
            const float p_c = get_p_hat_dev(C, m, sd.x1, sd.n1, can.x2, can.n2, can.L2, sd.o, false);
            const float c_m_num = cMmin * p_c; float mi_c = cMmin / M_sum_DI;
            for (int j = 0; j < n_di; j++) {
                const SData sn = load_sd_dev(B.cur_sd + cand_di[j] * 15); const Res rn = load_res_dev(B.cur_di + cand_di[j] * 10);
                const float nM = minf_u(128.0f, rn.M);
                const float p_from = get_p_hat_dev(C, m, sn.x1, sn.n1, can.x2, can.n2, can.L2, sn.o, true);
                const float m_den = c_m_num + (cMmax * p_from);
                if (m_den > 0.0f) mi_c += (nM / M_sum_DI) * (c_m_num / m_den);
            }
            const float w_c = mi_c * get_p_hat_dev(C, m, sd.x1, sd.n1, can.x2, can.n2, can.L2, sd.o, false) * can.W;
            const float gMmin = minf_u(128.0f, can_gi.M), gMmax = M_sum_GI - gMmin;
            const float pg_c = length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, can_gi.x2, can_gi.L2, sd.o, false));
            const float g_m_num = gMmin * pg_c; float mi_c_gi = gMmin / M_sum_GI;
            for (int j = 0; j < n_gi; j++) {
                const SData sn = load_sd_dev(B.cur_sd + cand_gi[j] * 15); const Res gn = load_res_dev(B.cur_gi + cand_gi[j] * 10);
                const float nM = minf_u(128.0f, gn.M);
                const float j_gi = jacobian_dev(sd, sn, can_gi.x2, can_gi.n2);
                const float p_from = length(get_p_hat_gi_dev(C, m, sn.x1, sn.n1, can_gi.x2, can_gi.L2, sn.o, true)) * j_gi;
                const float m_den = g_m_num + (gMmax * p_from);
                if (m_den > 0.0f) mi_c_gi += (nM / M_sum_GI) * (g_m_num / m_den);
            }
            mi_c_gi = minf_(maxf_(mi_c_gi, 0.0f), 1.0f);
            const float w_c_gi = mi_c_gi * length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, can_gi.x2, can_gi.L2, sd.o, false)) * can_gi.W;
            rcur.M = (uint32_t)cMmin; rcur.w_sum = w_c;
            gcur.M = (uint32_t)gMmin; gcur.w_sum = w_c_gi;
            for (int v = 0; v < n_di; v++) {
                const SData sn = load_sd_dev(B.cur_sd + cand_di[v] * 15); const Res rn = load_res_dev(B.cur_di + cand_di[v] * 10);
                const float pc2 = get_p_hat_dev(C, m, sd.x1, sd.n1, can.x2, can.n2, can.L2, sd.o, false);
                const float p_from = get_p_hat_dev(C, m, sn.x1, sn.n1, can.x2, can.n2, can.L2, sn.o, false);
                const float m_num = (M_sum_DI - cMmin) * p_from, m_den = m_num + (cMmin * pc2);
                const float mi_s = m_den > 0.0f ? (minf_u(128.0f, rn.M) / M_sum_DI) * (m_num / m_den) : 0.0f;
                const float w_s = mi_s * get_p_hat_dev(C, m, sd.x1, sd.n1, rn.x2, rn.n2, rn.L2, sd.o, false) * rn.W;
                rcur.w_sum += w_s; rcur.M = (rcur.M + (uint32_t)minf_u(128.0f, rn.M)) & 0xFFFFu;
                if (tea_next(s0, s1) < w_s / rcur.w_sum) { rcur.x2 = rn.x2; rcur.n2 = rn.n2; rcur.L2 = rn.L2; }
            }
            for (int v = 0; v < n_gi; v++) {
                const SData sn = load_sd_dev(B.cur_sd + cand_gi[v] * 15); const Res gn = load_res_dev(B.cur_gi + cand_gi[v] * 10);
                const float pc2 = length(get_p_hat_gi_dev(C, m, sd.x1, sd.n1, can_gi.x2, can_gi.L2, sd.o, false));
                const float jj = jacobian_dev(sd, sn, can_gi.x2, can_gi.n2);
                const float p_from = length(get_p_hat_gi_dev(C, m, sn.x1, sn.n1, can_gi.x2, can_gi.L2, sn.o, false)) * jj;
                const float m_num = (M_sum_GI - gMmin) * p_from, m_den = m_num + (gMmin * pc2);
                const float mi_s = m_den > 0.0f ? minf_(maxf_((minf_u(128.0f, gn.M) / M_sum_GI) * (m_num / m_den), 0.0f), 1.0f) : 0.0f;
                const float j_gi = jacobian_dev(sn, sd, gn.x2, gn.n2);
                const f3 f_gi = get_p_hat_gi_dev(C, m, sd.x1, sd.n1, gn.x2, gn.L2, sd.o, true);
                const float w_s = mi_s * length(f_gi) * gn.W * j_gi;
                if (j_gi != 0.0f) {
                    gcur.w_sum += w_s; gcur.M = (gcur.M + (uint32_t)minf_u(128.0f, gn.M)) & 0xFFFFu;
                    if (tea_next(s0, s1) < w_s / gcur.w_sum) { gcur.x2 = gn.x2; gcur.n2 = gn.n2; gcur.L2 = gn.L2; }
                }
            }
            const float p_hat = get_p_hat_dev(C, m, sd.x1, sd.n1, rcur.x2, rcur.n2, rcur.L2, sd.o, true);
            rcur.W = get_w_dev(rcur.w_sum, p_hat);
            f3 acc = reconnect_di_dev(m, f.flags, sd.x1, sd.n1, rcur.x2, rcur.n2, rcur.L2, sd.o) * rcur.W;
            const f3 f_fin = get_p_hat_gi_dev(C, m, sd.x1, sd.n1, gcur.x2, gcur.L2, sd.o, false);
            gcur.W = get_w_dev(gcur.w_sum, length(f_fin));
            acc = acc + f_fin * gcur.W;
            store_res(B.last_di + slot * 10, rcur); store_res(B.last_gi + slot * 10, gcur);
            for (int k = 0; k < 15; k++) B.last_sd[slot * 15 + k] = B.cur_sd[slot * 15 + k];
            out = acc;
            n_sh += C.cnt_sh;
        }
        if (finite3(out)) { F4 a = accum[(size_t)y * W + x]; a.x = a.x + out.x; a.y = a.y + out.y; a.z = a.z + out.z; a.w = a.w + 1.0f; accum[(size_t)y * W + x] = a; }
    }
    atomicAdd(&counters[2], (unsigned long long)n_sh);
}

// ---------------------------------------------------------------------------------------------
// accumulate: gPermanentData running sum + count, RayGen_v6_pass3.hlsl:383-405.  Fixed order: the
// batch's samples are added in sample order, batches run in order on the stream.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accumulate(DevFrame f, DevPaths p, F4* __restrict__ accum) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        F4 a = accum[(size_t)y * f.width + x];
        for (uint32_t s = 0; s < f.batch_spp; s++) {
            const F4 r = p.rad[(size_t)s * f.npl + pl];
            const f3 rv = mk3(r.x, r.y, r.z);
            if (finite3(rv)) { a.x = a.x + rv.x; a.y = a.y + rv.y; a.z = a.z + rv.z; a.w = a.w + 1.0f; }
        }
        accum[(size_t)y * f.width + x] = a;
    }
}

// sRGB8 output: RayGen_v6_pass3.hlsl:405,428-441 + Common_v6.hlsl:353-376
__global__ __launch_bounds__(kBlock) void k_srgb8(const F4* __restrict__ accum, uint32_t npix, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    const F4 a = accum[i];
    const float cnt = maxf_(a.w, 1.0f);
    float c[3] = {a.x / cnt, a.y / cnt, a.z / cnt};
    if (is_nan(c[0]) || is_nan(c[1]) || is_nan(c[2])) { c[0] = 1.0f; c[1] = 0.0f; c[2] = 1.0f; }
    if (is_inf(c[0]) || is_inf(c[1]) || is_inf(c[2])) { c[0] = 0.0f; c[1] = 1.0f; c[2] = 1.0f; }
    uint32_t px = 0xFF000000u;
    for (int k = 0; k < 3; k++) {
        float v = c[k] <= 0.0031308f ? 12.92f * c[k] : 1.055f * pow_(c[k], 1.0f / 2.4f) - 0.055f;
        v = saturate(v);
        px |= ((uint32_t)(int)(v * 255.0f + 0.5f)) << (8 * k);
    }
    out[i] = px;
}

// tile slabs for the multi-GPU gather
__global__ __launch_bounds__(kBlock) void k_pack_tiles(DevFrame f, const F4* __restrict__ accum, F4* __restrict__ slab) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        F4 v = {0, 0, 0, 0};
        if (slot_to_pixel(f, pl, x, y)) v = accum[(size_t)y * f.width + x];
        slab[pl] = v;
    }
}
__global__ __launch_bounds__(kBlock) void k_unpack_tiles(DevFrame f, uint32_t nshards, const F4* __restrict__ slabs, F4* __restrict__ accum) {
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t total = f.npl * nshards;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        DevFrame g = f; g.shard_rank = i / f.npl; g.shard_count = nshards;
        uint32_t x, y;
        if (slot_to_pixel(g, i - g.shard_rank * f.npl, x, y)) accum[(size_t)y * f.width + x] = slabs[i];
    }
}

// ---------------------------------------------------------------------------------------------
// kernel-level debug entry points (parity tests): same device functions as the render loop
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dbg_trace(DevScene sc, const SmallRecPair* __restrict__ small, const F4* __restrict__ rays, uint32_t n, int any, F4* __restrict__ hits) {
    extern __shared__ F4 lds[];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const F4 ro = rays[2 * i], rd = rays[2 * i + 1];
        float t, u, v; uint32_t prim;
        if (any == 2) traverse_stats(sc, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        else if (any) trace_ray<true>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        else trace_ray<false>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        hits[i] = {t, u, v, u2f(prim)};
    }
}
__global__ __launch_bounds__(kBlock) void k_dbg_surface(DevScene sc, const F4* __restrict__ rays, const F4* __restrict__ hits, uint32_t n, F4* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const F4 h = hits[i];
    F4 z = {0, 0, 0, 0};
    out[4 * i] = z; out[4 * i + 1] = z; out[4 * i + 2] = z; out[4 * i + 3] = z;
    if (f2u(h.w) == kMissPrim) { out[4 * i].w = u2f(kMissMat); return; }
    const F4 ro = rays[2 * i], rd = rays[2 * i + 1];
    const Surf s = surface(sc, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), h.x, h.y, h.z, f2u(h.w));
    out[4 * i] = {s.pos.x, s.pos.y, s.pos.z, u2f(s.mat)};
    out[4 * i + 1] = {s.normal.x, s.normal.y, s.normal.z, s.area};
    out[4 * i + 2] = {u2f(s.inst), s.flat.x, s.flat.y, s.flat.z};
}
__global__ __launch_bounds__(kBlock) void k_dbg_bsdf_eval(DevScene sc, uint32_t mat, uint32_t flags, const float* __restrict__ in9, uint32_t n, float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* q = in9 + (size_t)i * 9; float* o = out8 + (size_t)i * 8;
    f3 F; float P, pd, ps;
    bsdf_mixture(sc.mats[mat], flags, mk3(q[0], q[1], q[2]), mk3(q[6], q[7], q[8]), mk3(q[3], q[4], q[5]), F, P, pd, ps);
    o[0] = F.x; o[1] = F.y; o[2] = F.z; o[3] = P; o[4] = pd; o[5] = ps; o[6] = 0.0f; o[7] = 0.0f;
}
__global__ __launch_bounds__(kBlock) void k_dbg_bsdf_sample(DevScene sc, uint32_t mat, uint32_t flags, const float* __restrict__ in8, uint32_t n, float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* q = in8 + (size_t)i * 8; float* o = out8 + (size_t)i * 8;
    uint32_t s0 = f2u(q[6]), s1 = f2u(q[7]);
    const f3 nrm = mk3(q[0], q[1], q[2]), wo = mk3(q[3], q[4], q[5]);
    const uint32_t st = select_strategy(sc.mats[mat], wo, nrm, flags, s0, s1);
    const f3 wi = sample_bsdf(sc.mats[mat], st, wo, nrm, s0, s1);
    o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = u2f(st); o[4] = u2f(s0); o[5] = u2f(s1); o[6] = 0.0f; o[7] = 0.0f;
}
__global__ void k_dbg_tea(uint32_t s0, uint32_t s1, uint32_t n, float* __restrict__ out, uint32_t* __restrict__ seed_out) {
    if (threadIdx.x || blockIdx.x) return;
    for (uint32_t i = 0; i < n; i++) out[i] = tea_next(s0, s1);
    seed_out[0] = s0; seed_out[1] = s1;
}
__global__ __launch_bounds__(kBlock) void k_dbg_primary(DevFrame f, const CameraGPU* __restrict__ cam, uint32_t sample_id, F4* __restrict__ rays) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= f.width * f.height) return;
    const uint32_t x = i % f.width, y = i / f.width;
    uint32_t s0, s1; seed_init(x, y, sample_id, f.frame_seed, s0, s1);
    float jx = 0.0f, jy = 0.0f;
    if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }
    f3 o, d; primary_ray(*cam, f.width, f.height, x, y, jx, jy, o, d);
    rays[2 * i] = {o.x, o.y, o.z, kTMinCam};
    rays[2 * i + 1] = {d.x, d.y, d.z, kTMax};
}


// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// GPU refit of the wide BVH after a transform-only commit (the reference refits its TLAS every frame: Renderer.cpp:594,
// TopLevelASGenerator.cpp:149-250).  Topology, slot assignment and triangle order stay; k_refit_tris re-derives the world-space
// triangles from the object-space vertices with the host's operation order (xform_point: bit-identical TriGPU records, so the
// triangle tests still match the oracle's), k_refit_nodes re-derives and re-quantises the child boxes level by level, deepest
// first.  Quantisation is conservative by construction: lo - p is rounded DOWN before floor(), hi - p UP before ceil(), and
// 2^e is chosen with 255 * 2^e >= extent, so the decoded planes bracket the float boxes exactly as the host's double-checked
// build does.
// ---------------------------------------------------------------------------------------------
// directed-rounding stand-ins (HIP has no __fsub_rd / __fsub_ru here): the neighbours of the round-to-nearest result bracket the
// exact difference (|exact - fl| <= half a spacing), at the price of at most one extra spacing of slack
__device__ __forceinline__ float next_below(float x) { uint32_t b = f2u(x); if (x > 0.0f) b--; else if (x < 0.0f) b++; else b = 0x80000001u; return u2f(b); }
__device__ __forceinline__ float next_above(float x) { uint32_t b = f2u(x); if (x > 0.0f) b++; else if (x < 0.0f) b--; else b = 0x00000001u; return u2f(b); }
__device__ __forceinline__ float sub_down(float a, float b) { return next_below(a - b); }
__device__ __forceinline__ float sub_up(float a, float b) { return next_above(a - b); }

__global__ __launch_bounds__(kBlock) void k_refit_tris(TriGPU* __restrict__ tris, uint32_t ntris, const TriShade* __restrict__ shade, const InstGPU* __restrict__ insts,
                                                       const F4* __restrict__ objtris, uint32_t* __restrict__ scale_bits) {
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    float amax = 0.0f;
    if (s < ntris) {
        const uint32_t g = f2u(tris[s].v0.w);
        const float* M = insts[shade[g].inst].o2w;
        const F4 a = objtris[(size_t)g * 3], b = objtris[(size_t)g * 3 + 1], c = objtris[(size_t)g * 3 + 2];
        const f3 w0 = xform_point(M, mk3(a.x, a.y, a.z)), w1 = xform_point(M, mk3(b.x, b.y, b.z)), w2 = xform_point(M, mk3(c.x, c.y, c.z));
        const f3 e1 = w1 - w0, e2 = w2 - w0;
        tris[s].v0 = {w0.x, w0.y, w0.z, u2f(g)};
        tris[s].e1 = {e1.x, e1.y, e1.z, 0.0f};
        tris[s].e2 = {e2.x, e2.y, e2.z, 0.0f};
        amax = fmaxf(fmaxf(fmaxf(fabsf(w0.x), fabsf(w0.y)), fmaxf(fabsf(w0.z), fabsf(w1.x))), fmaxf(fmaxf(fabsf(w1.y), fabsf(w1.z)), fmaxf(fmaxf(fabsf(w2.x), fabsf(w2.y)), fabsf(w2.z))));
    }
    atomicMax(&s_max, f2u(amax));                      // non-negative floats order like their bit patterns
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(scale_bits, s_max);
}

__global__ __launch_bounds__(kBlock) void k_refit_nodes(Node8GPU* __restrict__ nodes, uint32_t first, uint32_t count, const TriGPU* __restrict__ tris,
                                                        F4* __restrict__ node_aabb /* 2 per node: min, max */, const uint32_t* __restrict__ scale_bits) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t n = first + i;
    Node8GPU N = nodes[n];
    const float pad = 2e-6f * u2f(*scale_bits);          // the host build's bvh_pad (rtx_scene_host.cpp)
    const uint32_t imask = N.e_imask >> 24;
    float cmn[8][3], cmx[8][3];
    float bmn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t rank = 0, tri_at = N.tri_base, used = 0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) {
        const uint32_t nib = (N.trivalid >> (4 * sl)) & 0xfu;
        for (int a = 0; a < 3; a++) { cmn[sl][a] = 0.0f; cmx[sl][a] = 0.0f; }
        if ((imask >> sl) & 1u) {
            const F4 mn = node_aabb[2 * (size_t)(N.child_base + rank)], mx = node_aabb[2 * (size_t)(N.child_base + rank) + 1];
            rank++;
            cmn[sl][0] = mn.x; cmn[sl][1] = mn.y; cmn[sl][2] = mn.z; cmx[sl][0] = mx.x; cmx[sl][1] = mx.y; cmx[sl][2] = mx.z;
        } else if (nib) {
            float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            const uint32_t cnt = (uint32_t)__builtin_popcount(nib);
            for (uint32_t k = 0; k < cnt; k++, tri_at++) {
                const TriGPU T = tris[tri_at];
                const float v[3][3] = {{T.v0.x, T.v0.y, T.v0.z}, {T.v0.x + T.e1.x, T.v0.y + T.e1.y, T.v0.z + T.e1.z}, {T.v0.x + T.e2.x, T.v0.y + T.e2.y, T.v0.z + T.e2.z}};
                for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], fminf(v[0][a], fminf(v[1][a], v[2][a]))); mx[a] = fmaxf(mx[a], fmaxf(v[0][a], fmaxf(v[1][a], v[2][a]))); }
            }
            for (int a = 0; a < 3; a++) { cmn[sl][a] = mn[a] - pad; cmx[sl][a] = mx[a] + pad; }
        } else continue;
        used |= 1u << sl;
        for (int a = 0; a < 3; a++) { bmn[a] = fminf(bmn[a], cmn[sl][a]); bmx[a] = fmaxf(bmx[a], cmx[sl][a]); }
    }
    if (!used) { for (int a = 0; a < 3; a++) { bmn[a] = 0.0f; bmx[a] = 0.0f; } }
    node_aabb[2 * (size_t)n] = {bmn[0], bmn[1], bmn[2], 0.0f}; node_aabb[2 * (size_t)n + 1] = {bmx[0], bmx[1], bmx[2], 0.0f};
    // byte grid: p = box minimum, smallest power of two with 255 steps covering the (upward-rounded) extent
    uint32_t eb[3]; float inv_step[3];
    for (int a = 0; a < 3; a++) {
        const float ext = sub_up(bmx[a], bmn[a]);
        int e = -120;
        if (ext > 0.0f) {
            int k; const float m = frexpf(ext, &k);            // ext = m * 2^k, m in [0.5, 1)
            e = m <= 0.99609375f ? k - 8 : k - 7;               // 255 * 2^(k-8) = 0.99609375 * 2^k
            if (e < -120) e = -120;
            if (e > 120) e = 120;                               // (cannot cover; such coordinates are rejected at commit)
        }
        eb[a] = (uint32_t)(e + 127); inv_step[a] = u2f((uint32_t)(127 - e) << 23);
    }
    N.px = bmn[0]; N.py = bmn[1]; N.pz = bmn[2];
    N.e_imask = eb[0] | eb[1] << 8 | eb[2] << 16 | imask << 24;
    for (int r = 0; r < 12; r++) N.q[r] = 0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) {
        if (!((used >> sl) & 1u)) continue;
        for (int a = 0; a < 3; a++) {
            float qlo = floorf(sub_down(cmn[sl][a], bmn[a]) * inv_step[a]), qhi = ceilf(sub_up(cmx[sl][a], bmn[a]) * inv_step[a]);
            qlo = fminf(255.0f, fmaxf(0.0f, qlo)); qhi = fminf(255.0f, fmaxf(0.0f, qhi));
            N.q[2 * a + (sl >> 2)] |= (uint32_t)qlo << (8 * (sl & 3));
            N.q[2 * (3 + a) + (sl >> 2)] |= (uint32_t)qhi << (8 * (sl & 3));
        }
    }
    nodes[n] = N;
}

// ---------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t items, uint32_t max_blocks) {
    uint32_t b = (items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return b < max_blocks ? b : max_blocks;
}
size_t trace_lds_bytes(const DevScene& sc) {      // the LDS column stack is always reserved: debug / pass-1 kernels use it
    return (size_t)sc.lds_nodes * 80 + (size_t)sc.lds_tris * 48 + (size_t)sc.stack_depth * kBlock * 8;
}
size_t trace_lds_bytes_queue(const DevScene& sc) {   // queue kernels with a private stack need no LDS stack
    const size_t stack = sc.stack_private == 1 ? 0 : (size_t)sc.stack_depth * kBlock * 8;
    return (size_t)sc.lds_nodes * 80 + (size_t)sc.lds_tris * 48 + stack;
}

void launch_raygen(hipStream_t st, const DevFrame& f, const DevPaths& p, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount) {
    hipLaunchKernelGGL(k_raygen, dim3(f.nblocks), dim3(kBlock), 0, st, f, p, cam, queue, qcount);
}
void launch_raygen_trace_small(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount, uint32_t* gencount) {
    hipLaunchKernelGGL(k_raygen_trace_small, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, p, cam, queue, qcount, gencount);
}
void launch_trace_closest(hipStream_t st, const DevFrame& f, const DevScene& sc, const DevPaths& p, uint32_t bounce, const uint32_t* queue, const uint32_t* qcount) {
    const float tmin = bounce == 0 ? kTMinCam : kSBias;
    if (sc.stack_private == 1) hipLaunchKernelGGL(k_trace_closest<1>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes_queue(sc), st, sc, sc.small, p, queue, qcount, f.qcap, tmin, sc.refill_min, sc.trace_sched);
    else hipLaunchKernelGGL(k_trace_closest<0>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, p, queue, qcount, f.qcap, tmin, sc.refill_min, sc.trace_sched);
}
void launch_bounce_small(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce, bool have_hit,
                         const uint32_t* queue, const uint32_t* qcount, uint32_t* next_queue, uint32_t* next_count, uint32_t* shcounts) {
    // 4 waves/SIMD (114 VGPRs, no spills); forcing 5 or 6 spills to scratch and measured 6 % / 16 % slower
    if (have_hit) hipLaunchKernelGGL((k_bounce_small<4, true>), dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
    else hipLaunchKernelGGL((k_bounce_small<4, false>), dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
}
void launch_trace_shadow(hipStream_t st, const DevFrame& f, const DevScene& sc, const DevPaths& p, uint32_t j, const uint32_t* shcount) {
    const size_t seg = (size_t)j * f.qcap * f.nblocks;
    if (sc.stack_private == 1) hipLaunchKernelGGL(k_trace_shadow<1>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes_queue(sc), st, sc, sc.small, p, p.sh_o + seg, p.sh_d + seg, p.sh_c + seg, shcount, f.qcap, sc.refill_min, sc.trace_sched);
    else hipLaunchKernelGGL(k_trace_shadow<0>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, p, p.sh_o + seg, p.sh_d + seg, p.sh_c + seg, shcount, f.qcap, sc.refill_min, sc.trace_sched);
}
void launch_shade(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce,
                  const uint32_t* queue, const uint32_t* qcount, uint32_t* next_queue, uint32_t* next_count, uint32_t* shcounts) {
    // material-sorted variant: measured slower (see k_shade), the permutation un-coalesces the per-path state streams
    if (sc.sort_materials) hipLaunchKernelGGL(k_shade<true>, dim3(f.nblocks), dim3(kBlock), 0, st, sc, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
    else hipLaunchKernelGGL(k_shade<false>, dim3(f.nblocks), dim3(kBlock), 0, st, sc, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
}
void launch_v6_pass1(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t sample_id,
                     F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata, unsigned long long* counters) {
    hipLaunchKernelGGL(k_v6_pass1, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, sample_id, accum, res_di, res_gi, sdata, counters);
}
void launch_restir_pass2(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t* const bufs[6], unsigned long long* counters) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_pass2, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, B, counters);
}
void launch_restir_pass3(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t* const bufs[6], F4* accum, unsigned long long* counters) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_pass3, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, B, accum, counters);
}
void launch_accumulate(hipStream_t st, uint32_t max_blocks, const DevFrame& f, const DevPaths& p, F4* accum) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), 0, st, f, p, accum);
}
void launch_srgb8(hipStream_t st, const F4* accum, uint32_t npix, uint32_t* out) {
    hipLaunchKernelGGL(k_srgb8, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, st, accum, npix, out);
}
void launch_pack_tiles(hipStream_t st, uint32_t max_blocks, const DevFrame& f, const F4* accum, F4* slab) {
    hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), 0, st, f, accum, slab);
}
void launch_unpack_tiles(hipStream_t st, uint32_t max_blocks, const DevFrame& f, uint32_t nshards, const F4* slabs, F4* accum) {
    hipLaunchKernelGGL(k_unpack_tiles, dim3(grid_for(f.npl * nshards, max_blocks)), dim3(kBlock), 0, st, f, nshards, slabs, accum);
}
void launch_refit(hipStream_t st, Node8GPU* nodes, const uint32_t* level_start, uint32_t nlevels, TriGPU* tris, uint32_t ntris, const TriShade* shade,
                  const InstGPU* insts, const F4* objtris, F4* node_aabb, uint32_t* scale_bits) {
    if (ntris) hipLaunchKernelGGL(k_refit_tris, dim3((ntris + kBlock - 1) / kBlock), dim3(kBlock), 0, st, tris, ntris, shade, insts, objtris, scale_bits);
    for (uint32_t l = nlevels; l-- > 0;) {                       // deepest level first: children are refitted before their parents
        const uint32_t first = level_start[l], count = level_start[l + 1] - first;
        if (count) hipLaunchKernelGGL(k_refit_nodes, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nodes, first, count, tris, node_aabb, scale_bits);
    }
}
void launch_dbg_trace(hipStream_t st, const DevScene& sc, const F4* rays, uint32_t n, int any, F4* hits) {
    hipLaunchKernelGGL(k_dbg_trace, dim3(grid_for(n, 2048)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, rays, n, any, hits);
}
void launch_dbg_surface(hipStream_t st, const DevScene& sc, const F4* rays, const F4* hits, uint32_t n, F4* out) {
    hipLaunchKernelGGL(k_dbg_surface, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, rays, hits, n, out);
}
void launch_dbg_bsdf_eval(hipStream_t st, const DevScene& sc, uint32_t mat, uint32_t flags, const float* in9, uint32_t n, float* out8) {
    hipLaunchKernelGGL(k_dbg_bsdf_eval, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, mat, flags, in9, n, out8);
}
void launch_dbg_bsdf_sample(hipStream_t st, const DevScene& sc, uint32_t mat, uint32_t flags, const float* in8, uint32_t n, float* out8) {
    hipLaunchKernelGGL(k_dbg_bsdf_sample, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, mat, flags, in8, n, out8);
}
void launch_dbg_tea(hipStream_t st, uint32_t s0, uint32_t s1, uint32_t n, float* out, uint32_t* seed_out) {
    hipLaunchKernelGGL(k_dbg_tea, dim3(1), dim3(64), 0, st, s0, s1, n, out, seed_out);
}
void launch_dbg_primary(hipStream_t st, const DevFrame& f, const CameraGPU* cam, uint32_t sample_id, F4* rays) {
    hipLaunchKernelGGL(k_dbg_primary, dim3((f.width * f.height + kBlock - 1) / kBlock), dim3(kBlock), 0, st, f, cam, sample_id, rays);
}

}  // namespace rtx
