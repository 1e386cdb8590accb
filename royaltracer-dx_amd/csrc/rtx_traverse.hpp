// rtx_traverse.hpp — ray traversal on the device: LDS staging, the exact triangle test, the compressed 8-wide BVH (node step, simple and
// persistent-wave traversal with the while-while / voted / speculative schedules), the tiny-scene pre-test path, packet culling
#pragma once
#include "rtx_dev_common.hpp"

namespace rtx {

// Section profiler (tooling only: `make PROFILE=1`, tools/section_profile.py): s_memtime deltas per code section, summed per wave
// and flushed to g_sec[] by lane 0.  With several waves per SIMD the deltas include the other waves' issue slots, so the SHARES
// are meaningful, not the absolute cycle counts.  Compiled out of the product library.
#ifdef RTX_PROFILE_SECTIONS
struct Prof {
    unsigned long long t, acc[12], lanes[12], calls[12];
    __device__ __forceinline__ void begin() { t = __builtin_readcyclecounter(); for (int i = 0; i < 12; i++) { acc[i] = 0; lanes[i] = 0; calls[i] = 0; } }
    __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += n - t; t = n; }
    // how many lanes are executing at this point (EXEC), once per call: lanes[i] / calls[i] = average active lanes of the code that follows
    __device__ __forceinline__ void count(int i) { lanes[i] += (unsigned long long)__builtin_popcountll(__builtin_amdgcn_read_exec()); calls[i] += 1; }
};
#define RTX_PROF_MARK(pf, i) do { if (pf) (pf)->mark(i); } while (0)
#define RTX_PROF_COUNT(pf, i) do { if (pf) (pf)->count(i); } while (0)
#else
struct Prof {};
#define RTX_PROF_MARK(pf, i) do { } while (0)
#define RTX_PROF_COUNT(pf, i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// BVH traversal
// ---------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type: through a generic pointer hipcc emits flat_load /
// flat_store for the staged nodes and the traversal stack instead of ds_read_b128 / ds_write_b32 (found with
// SQ_INSTS_LDS vs SQ_INSTS_VMEM_RD in a round-1 counter run).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2v splat2(float x) { f2v r = {x, x}; return r; }
// acc = (acc << 1) | c as ONE VALU instruction: acc + acc + carry-in, the carry-in being c's lane mask (the compare writes it to an SGPR pair)
// (m is built from ballots of single compares OR-ed on the scalar unit: the ballot of an i1 expression goes through v_cndmask + v_cmp_ne)
__device__ __forceinline__ uint32_t shift_in(uint32_t acc, unsigned long long m) {
    uint32_t r;
    asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(acc), "s"(m) : "vcc");
    return r;
}
// cm |ind| + 1e-5 |t| in two instructions (|x| as a source modifier; left to the compiler it becomes four v_and + two packed ops)
__device__ __forceinline__ float margin_t(float cm, float ind, float t) {
    float a, r;
    asm("v_mul_f32_e64 %0, |%1|, %2" : "=v"(a) : "v"(t), "v"(1e-5f));
    asm("v_fma_f32 %0, |%1|, %2, %3" : "=v"(r) : "v"(ind), "s"(cm), "v"(a));
    return r;
}
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2u lds_u2;
typedef __attribute__((address_space(3))) uint16_t lds_u16;
struct TraceLds {
    const lds_v4f* nodes;   // LDS copy of nodes [0, lds_nodes)
    const lds_v4f* tris;    // LDS copy of tris  [0, lds_tris)
    const lds_v4f* planes;  // tiny scenes: plane (n, n.p0) of every pre-test record
    lds_u32* stack;         // [depth][kBlock] sibling-group entries: the group's base index ...
    lds_u16* stack_bits;    // [depth][kBlock] ... and its 16 bits (ordered internal hits | internal mask << 8): 6 B per entry and lane (kStackEntryBytes)
    unsigned long long* ovf; uint32_t cap, ovf_stride;      // this lane's column of the overflow area (DevScene::stack_ovf), entries in LDS, stride between overflow entries
};
__host__ __device__ inline uint32_t small_planes_count(uint32_t nsmall) { return (nsmall + 1u) & ~1u; }

// stage the top of the BVH and the first triangles into LDS (coalesced 16-B copies)
__device__ __forceinline__ TraceLds stage_lds(const DevScene& sc, F4* lds_generic) {
    TraceLds L;
    lds_v4f* ln = (lds_v4f*)lds_generic; lds_v4f* lt = ln + (size_t)sc.lds_nodes * 5;
    const v4f* gn = (const v4f*)sc.nodes; const v4f* gt = (const v4f*)(sc.nsmall ? sc.small_tris : sc.tris);
    for (uint32_t i = threadIdx.x; i < sc.lds_nodes * 5u; i += kBlock) ln[i] = gn[i];
    for (uint32_t i = threadIdx.x; i < sc.lds_tris * 3u; i += kBlock) lt[i] = gt[i];
    lds_v4f* lp = lt + (size_t)sc.lds_tris * 3;
    const uint32_t npl = sc.nsmall ? small_planes_count(sc.nsmall) : 0u;
    for (uint32_t i = threadIdx.x; i < npl; i += kBlock) {                    // rows 0-3 of SmallRecPair: the record's plane
        const float* r = &sc.small[i >> 1].r[0][i & 1u];
        const v4f pl = {r[0], r[2], r[4], r[6]};
        lp[i] = pl;
    }
    L.nodes = ln; L.tris = lt; L.planes = lp;
    L.stack = (lds_u32*)(lp + npl);
    L.stack_bits = (lds_u16*)(L.stack + (size_t)sc.stack_depth * kBlock);
    L.cap = sc.stack_depth; L.ovf_stride = sc.stack_ovf_stride;
    L.ovf = sc.stack_ovf ? sc.stack_ovf + ((blockIdx.x * kBlock + threadIdx.x) & (sc.stack_ovf_stride - 1u)) : nullptr;
    return L;
}

// Moeller-Trumbore with the fixed operation order shared with the oracle (a11).  Exclusive (tmin, tmax).  e1w.w = the triangle's determinant floor
// (rtx_math.hpp: tri_det_floor): |det| at or below it means the ray lies in the triangle's plane up to rounding, and the quotients would be 0 / 0.
// (A variant that checks the numerators conservatively before the IEEE division measured no faster: 32.4 vs 31.6 ms.)
__device__ __forceinline__ bool tri_test(f3 o, f3 d, v4f v0w, v4f e1w, v4f e2w, float tmin, float tmax, float& t, float& u, float& v) {
    const f3 v0 = mk3(v0w.x, v0w.y, v0w.z), e1 = mk3(e1w.x, e1w.y, e1w.z), e2 = mk3(e2w.x, e2w.y, e2w.z);
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (!(fabsf(det) > e1w.w)) return false;
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

// The same test without early exits (same operations in the same order, so t, u, v are the same bits whenever it accepts): for the wave
// schedules, where a step tests ONE triangle per lane and the exits only nest the control flow — the compiler kept the lane's best hit in a
// second register set across the nest and copied it on every exit edge (5 x 4 v_mov per triangle step).
__device__ __forceinline__ bool tri_test_flat(f3 o, f3 d, v4f v0w, v4f e1w, v4f e2w, float tmin, float tmax, float& t, float& u, float& v) {
    const f3 v0 = mk3(v0w.x, v0w.y, v0w.z), e1 = mk3(e1w.x, e1w.y, e1w.z), e2 = mk3(e2w.x, e2w.y, e2w.z);
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = 1.0f / det;
    const f3 s = o - v0;
    u = dot(s, p) * inv;
    const f3 q = cross(s, e1);
    v = dot(d, q) * inv;
    t = dot(e2, q) * inv;
    return (fabsf(det) > e1w.w) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > tmin) & (t < tmax);
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
constexpr float kSlabK = 1.00010002f;                 // >= kSlabHi / kSlabLo (1.000100005...): the same widening as one factor on the far side

__device__ __forceinline__ Node8R load_node8(const DevScene& sc, const TraceLds& L, uint32_t idx) {
    Node8R N;
    if (idx < sc.lds_nodes) { const lds_v4f* n = L.nodes + idx * 5u; N.h0 = n[0]; N.h1 = (v4u)n[1]; N.q0 = (v4u)n[2]; N.q1 = (v4u)n[3]; N.q2 = (v4u)n[4]; }
    else { const v4f* n = (const v4f*)sc.nodes_f + (size_t)idx * sc.node_v4; N.h0 = n[0]; N.h1 = (v4u)n[1]; N.q0 = (v4u)n[2]; N.q1 = (v4u)n[3]; N.q2 = (v4u)n[4]; }
    return N;
}
__device__ __forceinline__ uint32_t ray_octant(f3 idir) { return (idir.x < 0.0f ? 1u : 0u) | (idir.y < 0.0f ? 2u : 0u) | (idir.z < 0.0f ? 4u : 0u); }
__device__ __forceinline__ float byte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK

// tests the 8 children; G = this node's internal hits in octant order, T = the triangles of its hit leaf children
// ORDERED = false (any-hit rays): the visiting order does not matter, the octant permutation of the hit bits is skipped
// nx / ny / nz: the ray's direction is negative on that axis (bits 0-2 of its octant).  The persistent schedules hand them in as lanes of three WAVE MASKS kept in SGPR pairs
// (OctMasks below) — derived from the octant inside the step they cost 3 v_and + 3 v_cmp per node step, 3 % of its instructions, for values that only change on a refill.
// aord (any-hit rays, wave-uniform; DevScene::any_order): 0 = slot order, 1 = nearest octant first, 2 = FARTHEST first (the octant permutation of the mirrored direction) — the
// order of an any-hit ray changes no answer, only how soon an occluder turns up; which one pays is a property of the scene and is probed at commit (probe_anyhit_order)
template <bool ORDERED>
__device__ __forceinline__ void node8_hits(const Node8R& N, f3 o, f3 idir, bool nx, bool ny, bool nz, float tmin, float tbest, Grp& G, TriGrp& T, uint32_t aord = 0u) {
    const uint32_t w = f2u(N.h0.w);
    const float sx = u2f((w & 0xffu) << 23) * idir.x, sy = u2f((w & 0xff00u) << 15) * idir.y, sz = u2f((w & 0xff0000u) << 7) * idir.z;
    const float ax = (N.h0.x - o.x) * idir.x, ay = (N.h0.y - o.y) * idir.y, az = (N.h0.z - o.z) * idir.z;
    const float anx = __builtin_fmaf(-fabsf(ax), kPlaneEps, ax), afx = __builtin_fmaf(fabsf(ax), kPlaneEps, ax);
    const float any_ = __builtin_fmaf(-fabsf(ay), kPlaneEps, ay), afy = __builtin_fmaf(fabsf(ay), kPlaneEps, ay);
    const float anz = __builtin_fmaf(-fabsf(az), kPlaneEps, az), afz = __builtin_fmaf(fabsf(az), kPlaneEps, az);
    // rows: q0 = (lox0, lox1, loy0, loy1)  q1 = (loz0, loz1, hix0, hix1)  q2 = (hiy0, hiy1, hiz0, hiz1)
    const uint32_t qnx[2] = {nx ? N.q1.z : N.q0.x, nx ? N.q1.w : N.q0.y}, qfx[2] = {nx ? N.q0.x : N.q1.z, nx ? N.q0.y : N.q1.w};
    const uint32_t qny[2] = {ny ? N.q2.x : N.q0.z, ny ? N.q2.y : N.q0.w}, qfy[2] = {ny ? N.q0.z : N.q2.x, ny ? N.q0.w : N.q2.y};
    const uint32_t qnz[2] = {nz ? N.q2.z : N.q1.x, nz ? N.q2.w : N.q1.y}, qfz[2] = {nz ? N.q1.x : N.q2.z, nz ? N.q1.y : N.q2.w};
    // The verdict of child k is  lo <= hi kSlabK  <=>  hi kSlabK - lo >= 0  <=>  the SIGN BIT of d = fma(hi, kSlabK, -lo) is clear (d = -0 needs hi = -0 and
    // lo = +0, i.e. tmin = 0 and a box that ends exactly at the ray origin: every ray of the renderer has tmin > 0, and a leaf box is padded by 2e-6 scale, so
    // nothing that can hold a hit is lost).  So the eight bits are collected with ONE v_alignbit_b32 per child — (acc << 1) | sign(d) — instead of a compare
    // into an SGPR pair + v_addc_co + the hazard wait state between dependent carry chains: 12.4 instead of 23 issue cycles per pair of children.
    uint32_t nacc = 0;                                   // bit k = child k MISSED
    const f2v vsx = splat2(sx), vsy = splat2(sy), vsz = splat2(sz);
#pragma unroll
    for (int k = 6; k >= 0; k -= 2) {                    // two children per iteration on packed FP32 (v_pk_fma_f32), last pair first so that child 0 ends up in bit 0
        const int h = k >> 2, b = k & 3;
        const f2v bnx = {byte_f(qnx[h], b), byte_f(qnx[h], b + 1)}, bny = {byte_f(qny[h], b), byte_f(qny[h], b + 1)}, bnz = {byte_f(qnz[h], b), byte_f(qnz[h], b + 1)};
        const f2v bfx = {byte_f(qfx[h], b), byte_f(qfx[h], b + 1)}, bfy = {byte_f(qfy[h], b), byte_f(qfy[h], b + 1)}, bfz = {byte_f(qfz[h], b), byte_f(qfz[h], b + 1)};
        const f2v tnx = fma2(bnx, vsx, splat2(anx)), tny = fma2(bny, vsy, splat2(any_)), tnz = fma2(bnz, vsz, splat2(anz));
        const f2v tfx = fma2(bfx, vsx, splat2(afx)), tfy = fma2(bfy, vsy, splat2(afy)), tfz = fma2(bfz, vsz, splat2(afz));
        const f2v lo = {fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, tmin)), fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, tmin))};
        const f2v hi = {fminf(fminf(tfx.x, tfy.x), fminf(tfz.x, tbest)), fminf(fminf(tfx.y, tfy.y), fminf(tfz.y, tbest))};
        const f2v dd = fma2(hi, splat2(kSlabK), -lo);                     // lo kSlabLo <= hi kSlabHi as lo <= hi (kSlabHi / kSlabLo); lo >= tmin > 0
        nacc = __builtin_amdgcn_alignbit(nacc, f2u(dd.y), 31u);
        nacc = __builtin_amdgcn_alignbit(nacc, f2u(dd.x), 31u);
    }
    const uint32_t hits = ~nacc & 0xffu;
    const uint32_t imask = w >> 24;
    // internal hits, permuted so that bit j = slot (j ^ oct): lowest set bit = first child to visit
    uint32_t m = hits & imask;
    if (ORDERED) {
        if (nx) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
        if (ny) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
        if (nz) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    } else if (aord) {
        const bool far = aord == 2u;
        if (nx != far) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
        if (ny != far) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
        if (nz != far) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    }
    G.base = N.h1.x; G.bits = m | (imask << 8);
    // leaf hits: spread each bit to its nibble and keep the triangles that exist
    uint32_t x = hits & ~imask;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    T.base = N.h1.y; T.valid = N.h1.z; T.bits = (x * 15u) & N.h1.z;
}

// traversal stack of sibling groups: per-lane column in LDS (conflict-free 8-byte accesses), or a private array (scratch)
// (round 4) 6 B per entry: a 32-bit base and the 16 bits that are used of `bits`, in two arrays.  Two bytes per lane and level less than the 8-B entry are 4.5-5 KB per
// workgroup on the benchmark scenes: the first three levels of the wide tree (73 nodes, 5.8 KB) now fit beside the stack WITHOUT giving up a workgroup per CU.
// (round 5) RTX_OPT_STACK_CAP: the LDS column holds the first `cap` entries only — what almost every ray needs — and deeper entries go to a per-lane column in global memory
// (OVF).  A tree whose exact stack bound is 12 then costs the LDS of 9 entries like every other tree, i.e. the first three levels of the wide tree (73 nodes) stay staged at
// eight workgroups per CU (hard street scene, GPU-built tree: 24 staged nodes at a bound of 12 -> 73).  OVF = false (the hot kernels of trees within the cap): no test at all.
template <bool OVF>
struct StackLdsT { lds_u32* cb; lds_u16* ck; unsigned long long* ov; int cap; uint32_t os;
                   __device__ __forceinline__ void init(const TraceLds& L) { cb = L.stack + threadIdx.x; ck = L.stack_bits + threadIdx.x; ov = L.ovf; cap = (int)L.cap; os = L.ovf_stride; }
                   __device__ __forceinline__ void put(int i, Grp g) {
                       if (!OVF || i < cap) { cb[i * kBlock] = g.base; ck[i * kBlock] = (uint16_t)g.bits; }
                       else ov[(size_t)(i - cap) * os] = (unsigned long long)g.base | ((unsigned long long)g.bits << 32);
                   }
                   __device__ __forceinline__ Grp get(int i) const {
                       if (!OVF || i < cap) return Grp{cb[i * kBlock], (uint32_t)ck[i * kBlock]};
                       const unsigned long long e = ov[(size_t)(i - cap) * os];
                       return Grp{(uint32_t)e, (uint32_t)(e >> 32)};
                   } };
using StackLds = StackLdsT<true>;          // the general-purpose paths (debug queries, literal ReSTIR kernels, fused experiments): always safe, whatever the scene's cap
constexpr int kPrivStack = 32;
struct StackPriv { Grp a[kPrivStack]; __device__ __forceinline__ void put(int i, Grp g) { a[i] = g; } __device__ __forceinline__ Grp get(int i) const { return a[i]; } };

// pick the first child of group G (which has internal hits), keep the remaining siblings on the stack, test the child's
// eight children: G / T become the child's groups
// the rays' octant bits as three wave masks (lane l of x = ray l points towards -x ...): wave-uniform, refreshed by refill() whenever it has fetched new rays
struct OctMasks { unsigned long long x, y, z; };
template <bool ORDERED, class STK>
__device__ __forceinline__ void descend8(const DevScene& sc, const TraceLds& L, f3 o, f3 idir, uint32_t oct, float tmin, float tbest,
                                         Grp& G, TriGrp& T, STK& stk, int& sp, const OctMasks* om = nullptr) {
    const uint32_t k = (uint32_t)__builtin_ctz(G.bits);
    const uint32_t rest = G.bits & (G.bits - 1u);
    if (rest & 0xffu) { stk.put(sp, Grp{G.base, rest}); sp++; }
    const uint32_t aord = ORDERED ? 0u : sc.any_order;
    const uint32_t slot = ORDERED ? (k ^ oct) : (aord ? (k ^ oct ^ (aord == 2u ? 7u : 0u)) : k);
    const uint32_t idx = G.base + (uint32_t)__builtin_popcount((G.bits >> 8) & ((1u << slot) - 1u));
    const Node8R N = load_node8(sc, L, idx);
#ifdef RTX_NO_OCT_MASKS          // (A/B build: the conditions derived from the octant inside the step, as before round 3)
    om = nullptr;
#endif
    if (om) node8_hits<ORDERED>(N, o, idir, __builtin_amdgcn_inverse_ballot_w64(om->x), __builtin_amdgcn_inverse_ballot_w64(om->y), __builtin_amdgcn_inverse_ballot_w64(om->z), tmin, tbest, G, T);
    else node8_hits<ORDERED>(N, o, idir, (oct & 1u) != 0u, (oct & 2u) != 0u, (oct & 4u) != 0u, tmin, tbest, G, T, aord);
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
    StackLds stk; stk.init(L);
    int sp = 0;
    Grp G{0u, (ANY ? (sc.any_order ? 1u << (oct ^ (sc.any_order == 2u ? 7u : 0u)) : 1u) : (1u << oct)) | (1u << 8)};    // the root as slot 0 of a virtual parent (any-hit rays: DevScene::any_order)
    TriGrp T{0u, 0u, 0u};
    while (true) {
        if (G.bits & 0xffu) descend8<!ANY>(sc, L, o, idir, oct, tmin, bt, G, T, stk, sp);
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

// traversal that also counts node steps and triangle tests (rtx_debug_trace_stats: tree-quality measurements; ANY: the any-hit order probe of a GPU-built tree, rtx_api.hip)
template <bool ANY = false>
__device__ __forceinline__ void traverse_stats(const DevScene& sc, const TraceLds& L, f3 o, f3 d, float tmin, float tmax,
                                         float& bt, float& bu, float& bv, uint32_t& bprim) {
    uint32_t nsteps = 0, ntris = 0;
    // zero direction components -> huge finite reciprocal (keeps the slab test NaN-free and conservative)
    const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    const float dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    const float dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    const f3 idir = mk3(__builtin_amdgcn_rcpf(dxs), __builtin_amdgcn_rcpf(dys), __builtin_amdgcn_rcpf(dzs));
    const uint32_t oct = ray_octant(idir);
    bt = tmax; bu = 0.0f; bv = 0.0f; bprim = kMissPrim;
    StackLds stk; stk.init(L);
    int sp = 0;
    Grp G{0u, (ANY ? (sc.any_order ? 1u << (oct ^ (sc.any_order == 2u ? 7u : 0u)) : 1u) : (1u << oct)) | (1u << 8)};    // the root as slot 0 of a virtual parent (any-hit rays: DevScene::any_order)
    TriGrp T{0u, 0u, 0u};
    while (true) {
        if (G.bits & 0xffu) { descend8<!ANY>(sc, L, o, idir, oct, tmin, bt, G, T, stk, sp); nsteps++; }
        while (T.bits) {
            const uint32_t bit = (uint32_t)__builtin_ctz(T.bits);
            T.bits &= T.bits - 1u; ntris++;
            const uint32_t slot = tri_slot8(T, bit);
            v4f v0, e1, e2;
            if (slot < sc.lds_tris) { const lds_v4f* t = L.tris + slot * 3u; v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            else { const v4f* t = (const v4f*)(sc.tris + slot); v0 = t[0]; e1 = t[1]; e2 = t[2]; }
            float t, u, w;
            if (tri_test(o, d, v0, e1, e2, tmin, tmax, t, u, w)) {
                if (ANY) { bprim = 0u; bu = (float)nsteps; bv = (float)ntris; return; }
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
                                               float& bt, float& bu, float& bv, uint32_t& bprim, uint32_t nrec, unsigned long long keep = ~0ull, Prof* pf = nullptr, int pf_sec = 0) {
    // keep (wave-uniform): bit r clear = no ray of this wave can touch record r (primary-ray packet culling); nrec = sc.nsmall, or sc.nsmall_occ for NEE shadow segments (both end points inside the scene's convex hull: the records
    // after the first nsmall_occ are faces OF that hull and cannot lie between them, rtx_scene_host.cpp)
    bt = tmax; bu = 0.0f; bv = 0.0f; bprim = kMissPrim;
    uint32_t cand_lo = 0u, cand_hi = 0u;
    const uint32_t npairs = (nrec + 1u) >> 1;
    const f2v dx = splat2(d.x), dy = splat2(d.y), dz = splat2(d.z), ox = splat2(o.x), oy = splat2(o.y), oz = splat2(o.z);
    const f2v vtmin = splat2(tmin), vtmax = splat2(tmax);
    const float cm = sc.small_cm;
    // One record pair.  Instruction budget (the loop is the hottest code of the tiny-scene kernels, ~48 VALU per pair): the edge
    // constants are ADDED from SGPRs after the three FMAs (a VOP3P instruction takes one scalar operand, so an FMA addend from
    // SGPRs costs two v_mov), |x| rides on source modifiers of unpacked instructions, six slacks reduce through two min3 + one
    // min, and each record's verdict is shifted into the candidate word with ONE v_addc_co (carry-in = the compare mask).
    uint32_t near = ~0u;
    auto pair = [&](uint32_t kp, uint32_t& acc) {
        const f2v* __restrict__ R = (const f2v*)sp[kp].r;
        const f2v nd = fma2(R[2], dz, fma2(R[1], dy, R[0] * dx));
        const f2v no = R[3] - fma2(R[2], oz, fma2(R[1], oy, R[0] * ox));
        f2v ind; ind.x = __builtin_amdgcn_rcpf(nd.x); ind.y = __builtin_amdgcn_rcpf(nd.y);
        const f2v t = no * ind;
        const f2v px = fma2(t, dx, ox), py = fma2(t, dy, oy), pz = fma2(t, dz, oz);
        const f2v e0 = fma2(R[6], pz, fma2(R[5], py, R[4] * px)) + R[7];
        const f2v e1 = fma2(R[10], pz, fma2(R[9], py, R[8] * px)) + R[11];
        const f2v e2 = fma2(R[14], pz, fma2(R[13], py, R[12] * px)) + R[15];
        const f2v e3 = fma2(R[18], pz, fma2(R[17], py, R[16] * px)) + R[19];
        f2v mt; mt.x = margin_t(cm, ind.x, t.x); mt.y = margin_t(cm, ind.y, t.y);
        // all slack values must be >= 0: t in [tmin - mt, tmax + mt] and P within delta of the inside of every edge (e_k carries + delta)
        // (the distance tolerance delta of the edge planes is folded into their constants at build time: e_k >= 0 means "within delta")
        // The EDGE slacks get the margin mt as well (e_k >= -mt): the exact test's barycentrics carry an in-plane error of about
        // eps * scale / |n.d|, which for a ray a few milliradians off the plane exceeds the fixed tolerance delta folded into e_k (a shadow ray
        // at |n.d| = 1.1e-3 past the edge of a lamp flush with a wall: accepted by the float test 1e-4 outside, 1 pixel in 5000 random scenes).
        const f2v a0 = (t + mt) - vtmin, a1 = (vtmax + mt) - t;
        const float ma0 = fminf(a0.x, a1.x), ma1 = fminf(a0.y, a1.y);
        const float me0 = fminf(e0.x, fminf(fminf(e1.x, e2.x), e3.x)), me1 = fminf(e0.y, fminf(fminf(e1.y, e2.y), e3.y));
        const unsigned long long c0 = (__builtin_amdgcn_ballot_w64(ma0 >= 0.0f) & __builtin_amdgcn_ballot_w64(me0 >= -mt.x)) | __builtin_amdgcn_ballot_w64(fabsf(nd.x) < 1e-3f);   // grazing rays always go to the exact test
        const unsigned long long c1 = (__builtin_amdgcn_ballot_w64(ma1 >= 0.0f) & __builtin_amdgcn_ballot_w64(me1 >= -mt.y)) | __builtin_amdgcn_ballot_w64(fabsf(nd.y) < 1e-3f);
        acc = shift_in(shift_in(acc, c1), c0);                        // records run downwards, so record r ends up at bit r (mod 32)
        if (!ANY) {     // nearest candidate so far, as one sortable word: plane distance bits (positive floats order like integers) | record index
            const uint32_t k0 = (f2u(t.x) & ~63u) | (2u * kp), k1 = (f2u(t.y) & ~63u) | (2u * kp + 1u);
            const bool in0 = (ma0 >= 0.0f && me0 >= -mt.x) || (fabsf(nd.x) < 1e-3f), in1 = (ma1 >= 0.0f && me1 >= -mt.y) || (fabsf(nd.y) < 1e-3f);
            near = min(near, min(in0 ? k0 : ~0u, in1 ? k1 : ~0u));
        }
    };
    // records 2kp+1, 2kp for kp = npairs-1 .. 0: the first 32 records collect in cand_lo, the rest in cand_hi
#pragma unroll 1
    for (uint32_t kp = npairs; kp > 16u; kp--) {        // wave-uniform
        if (!((keep >> (2u * (kp - 1u))) & 3ull)) { cand_hi <<= 2; continue; }
        pair(kp - 1u, cand_hi);
    }
#pragma unroll 1
    for (uint32_t kp = npairs < 16u ? npairs : 16u; kp > 0u; kp--) {
        if (!((keep >> (2u * (kp - 1u))) & 3ull)) { cand_lo <<= 2; continue; }
        pair(kp - 1u, cand_lo);
    }
    RTX_PROF_MARK(pf, pf_sec);
    unsigned long long cand = ((unsigned long long)cand_hi << 32) | cand_lo;
    // bits >= nrec: the padding record of an odd count (zero plane: "grazing", always forwarded) or, for NEE segments, the first hull face
    cand &= nrec >= 64u ? ~0ull : ((1ull << nrec) - 1ull);
    // (Testing only the triangle on the ray's side of a quad's diagonal would save one exact test per candidate; measured slower
    // both ways: at 4 waves/SIMD the ~4 extra VGPRs spill (28.4-28.9 ms instead of 27.6), at 3 waves/SIMD without spills 31.3 ms.
    // Round 2, second attempt with the diagonal's line per record in LDS, the side decided per lane from the plane point with the
    // pre-test's tolerance, one triangle per lane and round: 92 VGPRs, no spills, bit-identical, 17.75 vs 17.59 ms — recomputing the
    // plane point and the bookkeeping cost what the shorter rounds save.)
    auto exact = [&](uint32_t k) -> bool {              // both triangles of record k; true = an any-hit ray is done
        RTX_PROF_COUNT(pf, pf_sec + 1);
#pragma unroll
        for (uint32_t h = 0; h < 2u; h++) {
            const lds_v4f* tp = L.tris + (2u * k + h) * 3u;
            const v4f v0 = tp[0], e1 = tp[1], e2 = tp[2];
            float t, u, w;
            if (tri_test(o, d, v0, e1, e2, tmin, tmax, t, u, w)) {
                if (ANY) { bprim = 0u; return true; }
                const uint32_t gid = f2u(v0.w);
                if (t < bt || (t == bt && gid < bprim)) { bt = t; bu = u; bv = w; bprim = gid; }
            }
        }
        return false;
    };
    if (ANY) {
        while (cand) {                                 // per-lane: exact test of the triangles of each candidate record
            const uint32_t k = (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            if (exact(k)) return;
        }
        return;
    }
    // Closest hit: a wave pays for the LONGEST candidate list among its lanes, and a ray through a room crosses 1-4 record
    // planes inside their polygons.  So each lane tests its NEAREST candidate first (phase 1 tracked it), and every further
    // candidate must first pass the phase-1 distance test again with the current best hit as the far end: it is skipped when its
    // plane distance minus the margin lies beyond bt (then its exact t is > bt: it can neither win nor tie).  Iterations in
    // which no lane has a surviving candidate skip the exact tests altogether.
    if (!cand) return;
    uint32_t k = near & 63u;
    if (!((cand >> k) & 1ull)) k = (uint32_t)__builtin_ctzll(cand);      // (the masked padding record can be "nearest")
    cand &= ~(1ull << k);
    bool go = true;
    while (go) {
        exact(k);
        go = false;
        while (cand) {
            k = (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            const v4f pl = L.planes[k];
            const float nd = __builtin_fmaf(pl.z, d.z, __builtin_fmaf(pl.y, d.y, pl.x * d.x));
            const float no = pl.w - __builtin_fmaf(pl.z, o.z, __builtin_fmaf(pl.y, o.y, pl.x * o.x));
            const float ind = __builtin_amdgcn_rcpf(nd);
            const float t = no * ind;
            const float mt = margin_t(cm, ind, t);
            if ((fabsf(nd) < 1e-3f) || !((t - mt) > bt)) { go = true; break; }
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
        // a polygon seen (nearly) edge-on gets a wider margin: for rays a few milliradians off its plane the exact test accepts hits up to
        // ~eps scale / |n.d| outside the polygon, which from the camera is more than the tenth of a pixel allowed below; the margin grows
        // with 1 / |n.d| from |n.d| = 0.02 on (the error reaches 1e-4 rad at ~0.01), up to 100-fold = 0.01 rad for a polygon exactly edge-on
        const f3 pn = cross(v[1] - v[0], v[2] - v[0]);
        float gmin = 1.0f;
        for (int k = 0; k < 4; k++) gmin = minf_(gmin, fabsf(dot(pn, c[k])) * rsqrt_det(maxf_(dot(pn, pn) * dot(c[k], c[k]), 1e-30f)));
        const float widen = gmin > 0.02f ? 1.0f : minf_(100.0f, 0.02f / maxf_(gmin, 1e-9f));      // |cos(angle between normal and corner ray)|
        for (int i = 0; i < 4; i++) {
            const float s = dot(n[i], mid) >= 0.0f ? 1.0f : -1.0f;     // orientation: the pyramid's inside has s * dot(n, .) >= 0
            const float nl1 = fabsf(n[i].x) + fabsf(n[i].y) + fabsf(n[i].z);
            bool all_out = true;
            for (int k = 0; k < 4; k++) {
                const float e = widen * 1e-4f * nl1 * (fabsf(v[k].x) + fabsf(v[k].y) + fabsf(v[k].z));
                all_out = all_out && (s * dot(n[i], v[k]) < -e);
            }
            culled = culled || all_out;
        }
    }
    return __ballot(r < sc.nsmall && !culled) | (sc.nsmall & 1u ? (1ull << sc.nsmall) : 0ull);   // (the padding record of an odd count is never inside anyway)
}


// ---------------------------------------------------------------------------------------------
// Persistent-wave BVH traversal with dynamic ray fetch (general scenes).  Lane utilisation of the plain
// one-ray-per-lane loop on a 262 k-triangle scene was 8/64 (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU,
// round-1 counters, see profiles/r01_pmc_bvh.md): traversal lengths have a heavy tail and internal / leaf phases diverge.  Here a
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

#ifndef RTX_PEND
#define RTX_PEND 2                                     // triangle groups a lane may hold (current + pending), speculative schedule
#endif
constexpr int kPend = RTX_PEND;
static_assert(kPend >= 2 && kPend <= 4, "RTX_PEND");
struct RayLane {                                       // per-lane traversal state
    f3 o, d, idir; float tmin, tmax, bt, bu, bv; uint32_t bprim; uint32_t oct; Grp G; TriGrp T, P[kPend - 1]; int sp; uint32_t item; bool has, done;   // P: pending triangle groups behind T (speculative schedule)
    uint32_t occluder;       // any-hit rays: slot of the triangle that occluded this lane's previous occluded ray (kNoOccluder: none yet), see ray_begin
    OctMasks om;             // WAVE-UNIFORM: ballots of the lanes' octant bits, kept current by refill() / refill_steal() (the only places where a lane gets a new ray)
};
__device__ __forceinline__ void oct_masks_update(RayLane& R) { R.om.x = __ballot((R.oct & 1u) != 0u); R.om.y = __ballot((R.oct & 2u) != 0u); R.om.z = __ballot((R.oct & 4u) != 0u); }
constexpr uint32_t kNoOccluder = 0xFFFFFFFFu;
__device__ __forceinline__ void pend_clear(RayLane& R) {
#pragma unroll
    for (int i = 0; i < kPend - 1; i++) R.P[i] = TriGrp{0u, 0u, 0u};
}
__device__ __forceinline__ void ray_begin(RayLane& R, f3 o, f3 d, float tmin, float tmax, uint32_t item, bool ordered, bool occluder_cache = false, uint32_t aord = 0u) {
    R.o = o; R.d = d; R.tmin = tmin; R.tmax = tmax; R.item = item;
    const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    const float dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    const float dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    R.idir = mk3(__builtin_amdgcn_rcpf(dxs), __builtin_amdgcn_rcpf(dys), __builtin_amdgcn_rcpf(dzs));
    R.oct = ray_octant(R.idir);
    R.bt = tmax; R.bu = 0.0f; R.bv = 0.0f; R.bprim = kMissPrim; R.sp = 0; R.has = true; R.done = false;
    R.G = Grp{0u, (ordered ? (1u << R.oct) : (aord ? 1u << (R.oct ^ (aord == 2u ? 7u : 0u)) : 1u)) | (1u << 8)}; R.T = TriGrp{0u, 0u, 0u}; pend_clear(R);      // the root as slot 0 of a virtual parent
    // OCCLUDER CACHE (any-hit rays).  A lane's consecutive rays come from neighbouring queue entries — shadow rays of neighbouring pixels towards the same light, or
    // visibility rays between neighbouring ReSTIR samples — and what blocked the last one often blocks the next.  So the lane's last occluder is handed to the new
    // ray as its first triangle group: it is tested by the first triangle step the wave takes, and a hit ends the ray before (most of) its traversal.  Any-hit is
    // existence, so the answer cannot change; a miss costs one triangle test.  (The slot persists in R.occluder across rays; spec_step records it on a hit.)
    // MEASURED (round 3, same box, alternating, RTX_OPT_OCCLUDER_CACHE 0 / 1): slower everywhere — k_trace_shadow 11.63 -> 12.23 ms per frame on C3, 10.21 -> 10.48 on C5,
    // the ReSTIR frames (27 M visibility rays at 1080p) 9.69 -> 9.79 ms on the atrium: too few rays are blocked by the SAME triangle as their queue neighbour, and the
    // extra triangle step every ray now starts with runs at the triangle steps' low lane count.  Off by default; the knob stays as the measured alternative.
    if (occluder_cache && R.occluder != kNoOccluder) R.T = TriGrp{R.occluder, 1u, 1u};
}
__device__ __forceinline__ void ray_idle(RayLane& R) {
    R.has = false; R.done = false; R.sp = 0; R.item = 0; R.o = mk3(0, 0, 0); R.d = mk3(0, 0, 1); R.idir = mk3(0, 0, 1); R.oct = 0;
    R.tmin = 0.0f; R.tmax = 0.0f; R.bt = 0.0f; R.bu = 0.0f; R.bv = 0.0f; R.bprim = kMissPrim; R.G = Grp{0u, 0u}; R.T = TriGrp{0u, 0u, 0u}; pend_clear(R);
    R.occluder = kNoOccluder; R.om = OctMasks{0ull, 0ull, 0ull};
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
        descend8<!ANY>(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, R.T, stk, R.sp);
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
// Triangle step of the speculative schedule: tests the next pending triangle and only REPORTS what it found (closest hit: a record that
// beats the lane's best; any hit: a hit) — the caller applies it after the wave-uniform node / triangle branch has merged.  Updating the
// lane's best record inside the branch made the compiler keep it in a second register set there and copy it on every edge of the
// (nested) control flow: 12-24 v_mov per iteration, ~10 % of the loop's VALU instructions.
template <bool ANY>
__device__ __forceinline__ bool tri_candidate(const DevScene& sc, const TraceLds& L, RayLane& R, float& t, float& u, float& w, uint32_t& gid) {
    const uint32_t bit = (uint32_t)__builtin_ctz(R.T.bits);
    R.T.bits &= R.T.bits - 1u;
    const uint32_t slot = tri_slot8(R.T, bit);
    v4f v0, e1, e2;
    if (slot < sc.lds_tris) { const lds_v4f* tp = L.tris + slot * 3u; v0 = tp[0]; e1 = tp[1]; e2 = tp[2]; }
    else { const v4f* tp = (const v4f*)(sc.tris + slot); v0 = tp[0]; e1 = tp[1]; e2 = tp[2]; }
    const bool hit = tri_test_flat(R.o, R.d, v0, e1, e2, R.tmin, R.tmax, t, u, w);
    gid = ANY ? slot : f2u(v0.w);                      // any-hit: the occluder's slot (occluder cache); closest hit: the global triangle id
    if (ANY) return hit;
    return hit & ((t < R.bt) | ((t == R.bt) & (gid < R.bprim)));
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
        if (in_node) { descend8<!ANY>(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, R.T, stk, R.sp); next_group(R, stk); }
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
#ifdef RTX_WAVE_CLOCK      // tooling build (make VARIANT=wclk VARFLAGS=-DRTX_WAVE_CLOCK): start / end (s_memrealtime, 100 MHz) of every WAVE of the closest-hit launches after the camera rays' (tools/wave_timeline.py renders two bounces, so there is one)
__device__ unsigned long long g_wgt[2 * 65536];
#endif
#ifdef RTX_PROFILE_SECTIONS
__device__ unsigned long long g_trv[8];      // tooling (PROFILE build): node iterations, lanes in them, triangle iterations, lanes in them, busy lanes summed over iterations, iterations
#endif
// COUNT (the generic instantiations with RTX_OPT_TRACE_COUNTERS): the lane's node steps and triangle tests are tallied in cnt_nodes / cnt_tris (work per ray for the bench record)
template <bool ANY, class STK, bool COUNT = false>
__device__ __forceinline__ void spec_step(const DevScene& sc, const TraceLds& L, RayLane& R, STK& stk, uint32_t sched, uint32_t* cnt_nodes = nullptr, uint32_t* cnt_tris = nullptr) {
    const bool busy = R.has && !R.done;
    const bool has_tri = busy && R.T.bits != 0u;
    const bool can_node = busy && R.P[kPend - 2].bits == 0u && ((R.G.bits & 0xffu) != 0u || R.sp > 0);
    const uint32_t ni = (uint32_t)__popcll(__ballot(can_node)), nl = (uint32_t)__popcll(__ballot(has_tri));
#ifdef RTX_PROFILE_SECTIONS
    if (!ANY) {
        const uint32_t nbusy = (uint32_t)__popcll(__ballot(busy)), nboth = (uint32_t)__popcll(__ballot(can_node && has_tri));
        if (lane_id() == 0) {
            const uint32_t wn_ = sched == 7u ? 2u : 1u, wl_ = sched == 5u ? 1u : sched == 6u ? 2u : 1u;
            const bool node_it = ni * wn_ >= nl * wl_ && ni;
            atomicAdd(&g_trv[node_it ? 0 : 2], 1ull); atomicAdd(&g_trv[node_it ? 1 : 3], (unsigned long long)(node_it ? ni : nl));
            atomicAdd(&g_trv[4], (unsigned long long)nbusy); atomicAdd(&g_trv[5], 1ull);
            atomicAdd(&g_trv[6], (unsigned long long)nboth);     // lanes that could do either
        }
        if (busy && R.sp > 0) atomicMax(&g_trv[7], (unsigned long long)R.sp);       // deepest stack seen (the LDS column holds exactly sc.stack_depth entries)
    }
#endif
    const uint32_t wn = sched == 7u ? 2u : 1u, wl = sched == 5u ? 1u : sched == 6u ? 2u : 1u;
#if RTX_PEND == 2
    // Two slots: a node step writes its triangles into P (empty, or the lane could not take the step), a triangle step works on T, and ONE
    // place after both moves P up when T is empty.  (Placing the new group into "T if empty, else P" inside the node branch and shifting inside
    // the triangle branch is the same state machine, but the compiler then kept two register copies of both groups and paid 12-20 v_mov per
    // iteration to shuttle between them.)
    bool upd = false; float ct, cu, cw; uint32_t cg;                        // (set by the triangle branch only)
    if (ni * wn >= nl * wl && ni) {
        if (can_node) {
            if (!(R.G.bits & 0xffu)) { R.sp--; R.G = stk.get(R.sp); }
            descend8<!ANY>(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, R.P[0], stk, R.sp, ANY ? nullptr : &R.om);
            if (COUNT) *cnt_nodes += 1u;
        }
    } else if (has_tri) { upd = tri_candidate<ANY>(sc, L, R, ct, cu, cw, cg); if (COUNT) *cnt_tris += 1u; }
    uint32_t updv = upd ? 1u : 0u;
    asm volatile("" : "+v"(updv));                                          // opaque: keeps the update below OUT of the branch above (the optimiser would thread it back in)
    if (updv) {
        if (ANY) { R.bprim = 0u; R.done = true; R.T.bits = 0u; R.occluder = cg; }
        else { R.bt = ct; R.bu = cu; R.bv = cw; R.bprim = cg; }
    }
    if (!R.T.bits) { R.T = R.P[0]; R.P[0].bits = 0u; }
#else
    if (ni * wn >= nl * wl && ni) {
        if (can_node) {
            if (!(R.G.bits & 0xffu)) { R.sp--; R.G = stk.get(R.sp); }
            TriGrp Tn;
            descend8<!ANY>(sc, L, R.o, R.idir, R.oct, R.tmin, R.bt, R.G, Tn, stk, R.sp, ANY ? nullptr : &R.om);
            if (Tn.bits) {                                                        // first free slot: T, then P[0], P[1], ...
                bool placed = false;
                if (!R.T.bits) { R.T = Tn; placed = true; }
#pragma unroll
                for (int i = 0; i < kPend - 1; i++) if (!placed && !R.P[i].bits) { R.P[i] = Tn; placed = true; }
            }
        }
    } else if (has_tri) {
        tri_step<ANY>(sc, L, R);
        if (!R.T.bits) {
            R.T = R.P[0];
#pragma unroll
            for (int i = 0; i + 1 < kPend - 1; i++) R.P[i] = R.P[i + 1];
            R.P[kPend - 2] = TriGrp{0u, 0u, 0u};
        }
    }
#endif
    if (R.has && !R.done && !(R.G.bits & 0xffu) && R.sp == 0 && !R.T.bits) R.done = true;
}
// A workgroup of the persistent traversal kernels may own `merge` CONSECUTIVE sub-queues (thin launches: the late bounces of a frame, where Russian roulette has left a few
// dozen rays per sub-queue — one round of workgroups then costs a ray's latency whatever it holds, so fewer, fuller workgroups win; the host picks `merge` per launch from the
// previous frame's counts, rtx_render).  The sub-queues appear to refill() as one queue of n entries; locate() maps an entry back to (sub-queue, offset).  Everything is
// wave-uniform but idx.  Results do not depend on merge: every entry of every sub-queue is fetched exactly once either way.
struct MergedQ {
    uint32_t q0, n, pre[kMaxMerge];                 // first sub-queue, total entries, pre[t] = entries in the sub-queues before q0 + t
    __device__ __forceinline__ void init(const uint32_t* __restrict__ cnt, uint32_t nq, uint32_t merge) {
        q0 = blockIdx.x * merge; n = 0;
#pragma unroll
        for (uint32_t t = 0; t < kMaxMerge; t++) { pre[t] = n; if (t < merge && q0 + t < nq) n += cnt[q0 + t]; }
    }
    __device__ __forceinline__ void locate(uint32_t idx, uint32_t& q, uint32_t& off) const {
        uint32_t j = 0, b = 0;
#pragma unroll
        for (uint32_t t = 1; t < kMaxMerge; t++) if (idx >= pre[t]) { j = t; b = pre[t]; }      // the LAST t with pre[t] <= idx: skips empty sub-queues
        q = q0 + j; off = idx - b;
    }
};

// RTX_OPT_OCTANT_SORT: counting sort of ONE sub-queue's entries by their key byte (direction octant, or the cell of the ray's origin), by the whole workgroup: LDS atomics into a
// 256-bin histogram, an exclusive scan by one wave, a second pass that hands out ranks.  perm[0 .. n) = the entries grouped by key (not stable: no result depends on the order).
// scratch: 512 words of LDS (the traversal stack's area, idle before the first ray).  Ends with a barrier; perm is read back by the same workgroup only.
__device__ __forceinline__ void sort_by_key(const uint8_t* __restrict__ key, uint32_t* __restrict__ perm, uint32_t n, lds_u32* scratch) {
    lds_u32* hist = scratch; lds_u32* off = scratch + 256;
    hist[threadIdx.x] = 0u;                                      // kBlock = 256 threads, 256 bins
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) atomicAdd((uint32_t*)(hist + key[i]), 1u);
    __syncthreads();
    if (threadIdx.x < 64u) {                                     // exclusive scan of 256 counts by one wave: four bins per lane
        const uint32_t l = threadIdx.x;
        const uint32_t c0 = hist[4u * l], c1 = hist[4u * l + 1u], c2 = hist[4u * l + 2u], c3 = hist[4u * l + 3u];
        uint32_t incl = c0 + c1 + c2 + c3;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((int)l >= d) incl += t; }
        const uint32_t base = incl - (c0 + c1 + c2 + c3);
        off[4u * l] = base; off[4u * l + 1u] = base + c0; off[4u * l + 2u] = base + c0 + c1; off[4u * l + 3u] = base + c0 + c1 + c2;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kBlock) perm[atomicAdd((uint32_t*)(off + key[i]), 1u)] = i;
    __threadfence_block();
    __syncthreads();
}

// wave-level refill: returns false when the wave may exit (queue exhausted and nothing in flight)
template <bool MASKS, class Fetch>      // MASKS: keep R.om current (closest-hit kernels; the any-hit kernel measured 5 % SLOWER with the masks on C3 and derives the conditions from the octant)
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
            if (MASKS) oct_masks_update(R);
        }
    }
    return __ballot(R.has) != 0ull;
}

// Refill with work stealing (the separate trace kernels of general scenes).  Every sub-queue has a fetch cursor in GLOBAL memory
// (heads[q], zeroed before the launch) from which a WAVE takes chunks of kStealChunk entries into a wave-private range (no atomic
// per refill), and a bitmap (heads[G ...]) marks the sub-queues found exhausted.  A wave whose sub-queue is exhausted does not
// drain: it loads the bitmap, picks the first sub-queue not yet exhausted behind a pseudo-random position (thieves must spread over
// the victims: versions that sent all waves, or herds of them, to the same victim ran 5-80x slower — every wave of the herd pays an
// atomic and a bitmap load per victim), and fetches chunks from it like its owner
// would.  Lanes stay filled until EVERY sub-queue of the launch is empty, so the low-occupancy tail that each workgroup had at the
// end of its own sub-queue happens once per launch (busy lanes per iteration 50.3 -> 55.3 of 64, profiles/r02_traversal.md).
// A workgroup dispatched after its sub-queue was taken joins the stealing, or exits at once when all are exhausted.
// Results do not depend on which wave traces a ray (per-path state only).
// MEASURED (MI355X, same box, tools/steal_probe.py): wave iterations -8 %, but C3 56.6 vs 47.4 ms and C5 51.0 vs 44.7 ms per frame —
// the iterations it removes are the latency-bound ones of nearly empty waves, which cost few issue slots next to the full waves of
// the other workgroups on the SIMD, and with stealing all waves of a launch reach that tail together (+0.3-0.6 ms per launch).
// Hence RTX_OPT_WORK_STEALING defaults to 0.
constexpr uint32_t kStealChunk = 256;
struct RaySource { uint32_t* heads; const uint32_t* counts; uint32_t G, cur, n, lo, hi; };      // wave-uniform; heads[0..G) cursors, heads[G ...] exhausted bitmap
__device__ __forceinline__ uint32_t steal_words(uint32_t G) { return (G + 31u) / 32u; }
__device__ __forceinline__ bool all_exhausted(const uint32_t* heads, uint32_t G) {              // whole wave
    const uint32_t nw = steal_words(G);
    bool open = false;
    for (uint32_t k = lane_id(); k < nw; k += 64u) {
        const uint32_t valid = (G - k * 32u >= 32u) ? ~0u : ((1u << (G - k * 32u)) - 1u);
        open = open || (~__hip_atomic_load(heads + G + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & valid) != 0u;
    }
    return __ballot(open) == 0ull;
}
// first sub-queue not marked exhausted at or after bit `start` of the bitmap (cyclic); G = none left.  Whole wave.
__device__ __forceinline__ uint32_t pick_victim(const uint32_t* heads, uint32_t G, uint32_t start) {
    const uint32_t nw = steal_words(G), sw = (start >> 5) % nw, sb = start & 31u;
    for (uint32_t r = 0; r < nw + 1u; r += 64u) {                       // nw + 1 words: the start word is looked at twice (bits >= sb first, bits < sb last)
        const uint32_t k = r + lane_id();
        uint32_t w = 0, wi = 0;
        if (k <= nw) {
            wi = sw + k; while (wi >= nw) wi -= nw;
            uint32_t valid = (G - wi * 32u >= 32u) ? ~0u : ((1u << (G - wi * 32u)) - 1u);
            if (k == 0u) valid &= ~0u << sb;
            if (k == nw) valid &= ~(~0u << sb);
            w = ~__hip_atomic_load(heads + G + wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & valid;
        }
        const unsigned long long m = __ballot(w != 0u);
        if (m) {
            const int src = __builtin_ctzll(m);
            const uint32_t ww = (uint32_t)__builtin_amdgcn_readlane((int)w, src), wwi = (uint32_t)__builtin_amdgcn_readlane((int)wi, src);
            return wwi * 32u + (uint32_t)__builtin_ctz(ww);
        }
    }
    return G;
}
// a wave's pseudo-random sequence of start positions, seeded from workgroup and wave number
__device__ __forceinline__ uint32_t steal_seed() { return (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + 0x9e3779b9u; }
template <bool MASKS, class Fetch>
__device__ __forceinline__ bool refill_steal(RayLane& R, RaySource& W, bool& drained, uint32_t refill_min, uint32_t& rng, Fetch fetch) {
    unsigned long long idle = __ballot(!R.has);
    uint32_t nidle = (uint32_t)__popcll(idle);
    if (!drained && (nidle >= refill_min || nidle == 64u)) {            // wave-uniform
        for (;;) {
            if (W.lo < W.hi) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t take = nidle < W.hi - W.lo ? nidle : W.hi - W.lo;
                if (!R.has && rank < take) fetch(W.cur, W.lo + rank);
                W.lo += take;
                if (take == nidle) break;
                idle = __ballot(!R.has); nidle -= take;
                if (nidle < refill_min) break;
            }
            uint32_t base = 0;                                          // next chunk of the current sub-queue
            if (lane_id() == 0) base = atomicAdd(W.heads + W.cur, kStealChunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base < W.n) { W.lo = base; W.hi = base + kStealChunk < W.n ? base + kStealChunk : W.n; continue; }
            if (lane_id() == 0) atomicOr(W.heads + W.G + (W.cur >> 5), 1u << (W.cur & 31u));      // exhausted: mark it and take another
            rng = rng * 1664525u + 1013904223u;
            const uint32_t v = pick_victim(W.heads, W.G, (rng >> 8) % W.G);
            if (v >= W.G) { drained = true; break; }
            W.cur = v; W.n = W.counts[v];
        }
        if (MASKS) oct_masks_update(R);
    }
    return __ballot(R.has) != 0ull;
}

}  // namespace rtx
