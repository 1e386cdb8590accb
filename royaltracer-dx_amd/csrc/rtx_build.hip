// rtx_build.hip — the BVH build on the GPU (RTX_OPT_GPU_BUILD, see rtx_build.hpp).
//
//   triangles (object space) + instance matrices
//     -> k_gb_prims     world boxes, scene box, coordinate scale                       (the arithmetic of k_refit_tris: xform_point)
//     -> k_gb_morton    63-bit Morton keys of the box centres; rocPRIM radix sort (stable: ties keep the triangle order)
//     -> PLOC           parallel locally-ordered clustering (Meister & Bittner, TVCG 2018): per round  k_ploc_nn (nearest neighbour within +- radius places, boxes
//                       staged through LDS) | k_ploc_flags | exclusive scan | k_ploc_merge (mutual pairs become nodes; the SAH dynamic program's record of every new
//                       node is computed right there — its children are older nodes), until at most `ploc_top` clusters are left
//     -> HOST           the top of the tree over those <= 16 384 clusters: top-down binned SAH + re-insertion passes (build_cluster_top, milliseconds) and its records;
//                       the top is where every ray passes (5.7 of 12.4 node steps in the first three wide levels of the atrium), so it gets the expensive builder
//     -> layout         level by level from the root: k_lay_count (the children of each wide node by the recorded decisions, their octant slots) | scan | k_lay_emit
//                       (Node8GPU topology, the next level's nodes, the leaf-slot order of the triangles); k_lay_need (traversal stack bound, bottom-up)
//   and the caller's full refit (k_refit_tris, k_refit_nodes) derives the world triangles and quantises every node.
//
// No kernel depends on another workgroup of the same launch (MI355X: eight XCDs with an L2 each — cross-workgroup hand-offs inside one launch would need device-scope
// fences per node); every dependency is a launch boundary: PLOC rounds create nodes whose children are older, the layout runs level by level.
// Every decision is taken by code shared with the host (rtx_wide.hpp) from the same inputs in the same order, so the tree equals the one its HOST TWIN builds
// (BvhBuildOptions::ploc_radius on the host builder) node for node: tests/test_gpu_parity.py::test_gpu_build_equals_its_host_twin.
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <chrono>
#include "rtx_build.hpp"
#include "rtx_staging.hpp"
#include "rtx_kernels.hpp"
#include "rtx_wide.hpp"

namespace rtx {

namespace {
constexpr uint32_t kB = 256;
struct GBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) { if (n <= bytes && p) return hipSuccess; if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } if (!n) n = 16; hipError_t e = hipMalloc(&p, n); if (e == hipSuccess) bytes = n; return e; }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};
struct Pool { F4* mn; F4* mx; int32_t* left; int32_t* right; WideDp* dp; uint32_t nleaf; };      // ids [0, nleaf): leaves in Morton order (left = -1, right = global triangle id)

__device__ __forceinline__ uint32_t enc_f(float f) { const uint32_t b = f2u(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }      // unsigned order == float order
__device__ __host__ inline float dec_f(uint32_t e) { const uint32_t b = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e; float f; memcpy(&f, &b, 4); return f; }
__device__ __forceinline__ WBox pool_box(const Pool& P, int32_t id) { const F4 a = P.mn[id], b = P.mx[id]; WBox w; w.mn[0] = a.x; w.mn[1] = a.y; w.mn[2] = a.z; w.mx[0] = b.x; w.mx[1] = b.y; w.mx[2] = b.z; return w; }
__device__ __host__ inline WBox padded(const WBox& b, float pad) { WBox w; for (int a = 0; a < 3; a++) { w.mn[a] = b.mn[a] - pad; w.mx[a] = b.mx[a] + pad; } return w; }      // the build's leaf-box padding (put_box)

// ---- world boxes, scene box, scale ----
__global__ __launch_bounds__(kB) void k_gb_prims(const F4* __restrict__ objtris, const TriShade* __restrict__ shade, const InstGPU* __restrict__ insts, uint32_t n,
                                                F4* __restrict__ bmn, F4* __restrict__ bmx, uint32_t* __restrict__ bounds /* 3 min + 3 max, encoded */, uint32_t* __restrict__ scale_bits) {
    __shared__ uint32_t s_b[6], s_max;
    if (threadIdx.x < 6) s_b[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint32_t g = blockIdx.x * kB + threadIdx.x;
    if (g < n) {
        const float* M = insts[shade[g].inst].o2w;
        const F4 a = objtris[(size_t)g * 3], b = objtris[(size_t)g * 3 + 1], c = objtris[(size_t)g * 3 + 2];
        const f3 w0 = xform_point(M, mk3(a.x, a.y, a.z)), w1 = xform_point(M, mk3(b.x, b.y, b.z)), w2 = xform_point(M, mk3(c.x, c.y, c.z));
        const float mn[3] = {fminf(w0.x, fminf(w1.x, w2.x)), fminf(w0.y, fminf(w1.y, w2.y)), fminf(w0.z, fminf(w1.z, w2.z))};
        const float mx[3] = {fmaxf(w0.x, fmaxf(w1.x, w2.x)), fmaxf(w0.y, fmaxf(w1.y, w2.y)), fmaxf(w0.z, fmaxf(w1.z, w2.z))};
        bmn[g] = {mn[0], mn[1], mn[2], 0.0f}; bmx[g] = {mx[0], mx[1], mx[2], 0.0f};
        const float amax = fmaxf(fmaxf(fmaxf(fabsf(w0.x), fabsf(w0.y)), fmaxf(fabsf(w0.z), fabsf(w1.x))), fmaxf(fmaxf(fabsf(w1.y), fabsf(w1.z)), fmaxf(fmaxf(fabsf(w2.x), fabsf(w2.y)), fabsf(w2.z))));
        for (int k = 0; k < 3; k++) { atomicMin(&s_b[k], enc_f(mn[k])); atomicMax(&s_b[3 + k], enc_f(mx[k])); }
        atomicMax(&s_max, f2u(amax));
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
    if (threadIdx.x == 0 && s_max) atomicMax(scale_bits, s_max);
}
__global__ __launch_bounds__(kB) void k_gb_morton(const F4* __restrict__ bmn, const F4* __restrict__ bmx, uint32_t n, const uint32_t* __restrict__ bounds,
                                                 unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t g = blockIdx.x * kB + threadIdx.x;
    if (g >= n) return;
    float smn[3], smx[3], lo[3], inv[3];
    for (int k = 0; k < 3; k++) { smn[k] = dec_f(bounds[k]); smx[k] = dec_f(bounds[3 + k]); }
    ploc_grid(smn, smx, lo, inv);
    const F4 a = bmn[g], b = bmx[g];
    WBox w; w.mn[0] = a.x; w.mn[1] = a.y; w.mn[2] = a.z; w.mx[0] = b.x; w.mx[1] = b.y; w.mx[2] = b.z;
    keys[g] = ploc_morton(w, lo, inv); vals[g] = g;
}
__global__ __launch_bounds__(kB) void k_gb_leaves(const uint32_t* __restrict__ vals, const F4* __restrict__ bmn, const F4* __restrict__ bmx, uint32_t n, Pool P, int32_t* __restrict__ cl) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = vals[i];
    P.mn[i] = bmn[g]; P.mx[i] = bmx[g]; P.left[i] = -1; P.right[i] = (int32_t)g; cl[i] = (int32_t)i;
}

// ---- PLOC rounds ----
__global__ __launch_bounds__(kB) void k_ploc_nn(const int32_t* __restrict__ cl, int m, int radius, Pool P, int32_t* __restrict__ nn) {
    extern __shared__ float tile[];                               // 6 floats per place, places [base - radius, base + 256 + radius)
    const int base = (int)(blockIdx.x * kB), lo = base - radius, cnt = (int)kB + 2 * radius;
    for (int t = (int)threadIdx.x; t < cnt; t += (int)kB) {
        const int j = lo + t;
        if (j >= 0 && j < m) { const WBox w = pool_box(P, cl[j]); for (int k = 0; k < 3; k++) { tile[t * 6 + k] = w.mn[k]; tile[t * 6 + 3 + k] = w.mx[k]; } }
    }
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    if (i >= m) return;
    nn[i] = ploc_nearest(i, m, radius, [&](int j) { WBox w; const float* q = tile + (j - lo) * 6; for (int k = 0; k < 3; k++) { w.mn[k] = q[k]; w.mx[k] = q[3 + k]; } return w; });
}
// low word: the place survives (it is not the higher partner of a mutual pair); high word: it is the LOWER partner of one, i.e. a new node is made here
__global__ __launch_bounds__(kB) void k_ploc_flags(const int32_t* __restrict__ nn, int m, unsigned long long* __restrict__ flags) {
    const int i = (int)(blockIdx.x * kB + threadIdx.x);
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    flags[i] = (unsigned long long)((mutual && i > j) ? 0u : 1u) | ((unsigned long long)((mutual && i < j) ? 1u : 0u) << 32);
}
__device__ __forceinline__ WideDpChild dp_child(const Pool& P, int32_t id, float pad) {
    WideDpChild c; c.area = wbox_area(padded(pool_box(P, id), pad));
    if ((uint32_t)id < P.nleaf) { c.leaf_cnt = 1u; c.dp = nullptr; } else { c.leaf_cnt = 0u; c.dp = P.dp + id; }
    return c;
}
__global__ __launch_bounds__(kB) void k_ploc_merge(const int32_t* __restrict__ cl, const int32_t* __restrict__ nn, const unsigned long long* __restrict__ flags,
                                                  const unsigned long long* __restrict__ offs, int m, uint32_t base_id, Pool P, float pad, double tri_cost,
                                                  int32_t* __restrict__ cl_out, uint32_t* __restrict__ counts) {
    const int i = (int)(blockIdx.x * kB + threadIdx.x);
    if (i >= m) return;
    const unsigned long long f = flags[i], o = offs[i];
    if (i == m - 1) { counts[0] = (uint32_t)o + (uint32_t)(f & 1ull); counts[1] = (uint32_t)(o >> 32) + (uint32_t)(f >> 32); }
    if (!(f & 1ull)) return;                                      // absorbed by its partner
    const uint32_t pos = (uint32_t)o;
    if (f >> 32) {
        const int32_t id = (int32_t)(base_id + (uint32_t)(o >> 32)), a = cl[i], b = cl[nn[i]];
        const WBox ba = pool_box(P, a), bb = pool_box(P, b), u = wbox_union(ba, bb);
        P.mn[id] = {u.mn[0], u.mn[1], u.mn[2], 0.0f}; P.mx[id] = {u.mx[0], u.mx[1], u.mx[2], 0.0f}; P.left[id] = a; P.right[id] = b;
        WideDp rec;
        wide_dp_combine(dp_child(P, a, pad), dp_child(P, b, pad), wbox_area(wbox_union(padded(ba, pad), padded(bb, pad))), 1.0, tri_cost, rec);
        P.dp[id] = rec;
        cl_out[pos] = id;
    } else cl_out[pos] = cl[i];
}
// what the host needs of the clusters PLOC stopped at: pool id, box, record
__global__ __launch_bounds__(kB) void k_gb_clusters(const int32_t* __restrict__ cl, uint32_t m, Pool P, int32_t* __restrict__ ids, float* __restrict__ boxes6, WideDp* __restrict__ recs) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= m) return;
    const int32_t id = cl[i]; ids[i] = id;
    const WBox w = pool_box(P, id);
    for (int k = 0; k < 3; k++) { boxes6[(size_t)i * 6 + k] = w.mn[k]; boxes6[(size_t)i * 6 + 3 + k] = w.mx[k]; }
    if ((uint32_t)id >= P.nleaf) recs[i] = P.dp[id]; else { WideDp z; memset(&z, 0, sizeof(z)); recs[i] = z; }
}

// ---- layout ----
struct DRef { WBox box; int32_t c; int32_t merged; };             // box: padded; c: pool id; merged: taken as ONE leaf slot with all its (<= 4) triangles
struct DAcc {
    Pool P; float pad;
    __device__ DRef ref(int32_t id) const { DRef r; r.box = padded(pool_box(P, id), pad); r.c = id; r.merged = 0; return r; }
    __device__ bool is_leaf(const DRef& r) const { return r.merged != 0 || (uint32_t)r.c < P.nleaf; }
    __device__ void children(const DRef& r, DRef& L, DRef& R) const { L = ref(P.left[r.c]); R = ref(P.right[r.c]); }
    __device__ uint8_t choice(const DRef& r, int i) const { return P.dp[r.c].choice[i]; }
    __device__ DRef merged(const DRef& r) const { DRef q = r; q.merged = 1; return q; }
};
// per wide node of one level: its children in slot order.  codes[8 h + slot]: >= 0 the pool id of an internal child, -1 empty, <= -2 a leaf slot holding the triangles of
// subtree -2 - code.  packed[h] = internal children | triangles << 32 (the scan gives child_base / tri_base); meta[h] = (imask, trivalid)
__global__ __launch_bounds__(kB) void k_lay_count(const int32_t* __restrict__ src, uint32_t cnt, DAcc A, int32_t* __restrict__ codes, uint2* __restrict__ meta, unsigned long long* __restrict__ packed) {
    const uint32_t h = blockIdx.x * kB + threadIdx.x;
    if (h >= cnt) return;
    const int32_t x = src[h];
    DRef ch[8]; bool internal[8];
    DRef L, R; A.children(A.ref(x), L, R);
    const int m = wide_children(A, L, R, (int)A.P.dp[x].choice[8], ch, internal);      // (m <= 8: a budget of 8 slots)
    WBox cb[8]; for (int k = 0; k < m && k < 8; k++) cb[k] = ch[k].box;
    float bmn[3], bmx[3]; int slot_of[8];
    wide_assign_slots(cb, m < 8 ? m : 8, bmn, bmx, slot_of);
    int32_t code[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    uint32_t imask = 0, trivalid = 0, nint = 0, ntri = 0;
    for (int k = 0; k < m && k < 8; k++) {
        const int sl = slot_of[k];
        if (internal[k]) { code[sl] = ch[k].c; imask |= 1u << sl; nint++; }
        else {
            const uint32_t c = ((uint32_t)ch[k].c < A.P.nleaf) ? 1u : A.P.dp[ch[k].c].prims;      // a binary leaf holds one triangle; a merged slot its subtree's <= 4
            code[sl] = -2 - ch[k].c; trivalid |= ((1u << c) - 1u) << (4 * sl); ntri += c;
        }
    }
    for (int sl = 0; sl < 8; sl++) codes[(size_t)h * 8 + sl] = code[sl];
    meta[h] = make_uint2(imask, trivalid);
    packed[h] = (unsigned long long)nint | ((unsigned long long)ntri << 32);
}
__global__ __launch_bounds__(kB) void k_lay_emit(const int32_t* __restrict__ codes, const uint2* __restrict__ meta, const unsigned long long* __restrict__ packed,
                                                const unsigned long long* __restrict__ offs, uint32_t cnt, uint32_t node_base, uint32_t tri_before, Pool P,
                                                Node8GPU* __restrict__ nodes, int32_t* __restrict__ src_next, TriGPU* __restrict__ tris, uint32_t* __restrict__ counts) {
    const uint32_t h = blockIdx.x * kB + threadIdx.x;
    if (h >= cnt) return;
    const unsigned long long o = offs[h];
    if (h == cnt - 1) { const unsigned long long t = o + packed[h]; counts[0] = (uint32_t)t; counts[1] = (uint32_t)(t >> 32); }
    Node8GPU N; memset(&N, 0, sizeof(N));
    N.e_imask = meta[h].x << 24; N.trivalid = meta[h].y;
    N.child_base = node_base + cnt + (uint32_t)o; N.tri_base = tri_before + (uint32_t)(o >> 32);
    nodes[node_base + h] = N;
    uint32_t rank = 0, at = N.tri_base;
    for (int sl = 0; sl < 8; sl++) {
        const int32_t c = codes[(size_t)h * 8 + sl];
        if (c >= 0) { src_next[(uint32_t)o + rank] = c; rank++; }
        else if (c <= -2) {                                       // the subtree's triangles, depth-first left to right (the leaf order of the binary tree)
            int32_t st[8]; int sp = 0; st[sp++] = -2 - c;
            while (sp > 0) {
                const int32_t id = st[--sp];
                if ((uint32_t)id < P.nleaf) { TriGPU T; memset(&T, 0, sizeof(T)); T.v0.w = u2f((uint32_t)P.right[id]); tris[at++] = T; }
                else if (sp + 2 <= 8) { st[sp++] = P.right[id]; st[sp++] = P.left[id]; }
            }
        }
    }
}
// traversal stack bound: a level adds ONE entry where a node has two or more internal children (collapse_bvh8: need[])
__global__ __launch_bounds__(kB) void k_lay_need(const Node8GPU* __restrict__ nodes, uint32_t first, uint32_t count, uint32_t* __restrict__ need) {
    const uint32_t i = blockIdx.x * kB + threadIdx.x;
    if (i >= count) return;
    const Node8GPU N = nodes[first + i];
    const uint32_t nint = (uint32_t)__builtin_popcount(N.e_imask >> 24);
    uint32_t deep = 0;
    for (uint32_t r = 0; r < nint; r++) deep = max(deep, need[N.child_base + r]);
    need[first + i] = (nint > 1u ? 1u : 0u) + deep;
}
}  // namespace

struct GpuBvhBuilder::Impl {
    GBuf bmn, bmx, bounds, keys[2], vals[2], sort_tmp, pmn, pmx, pleft, pright, pdp, cl[2], nn, flags, offs, scan_tmp, counts, cids, cboxes, crecs, src[2], codes, meta, packed, nodes, need;
    uint32_t* h_counts = nullptr;         // pinned: the two counters a PLOC round / a layout level reports
    Staging staging;                      // host arrays (cluster records down, the top of the tree up) pass through pinned chunks (rtx_staging.hpp)
    std::vector<GBuf*> all() { return {&bmn, &bmx, &bounds, &keys[0], &keys[1], &vals[0], &vals[1], &sort_tmp, &pmn, &pmx, &pleft, &pright, &pdp, &cl[0], &cl[1], &nn, &flags, &offs, &scan_tmp,
                                       &counts, &cids, &cboxes, &crecs, &src[0], &src[1], &codes, &meta, &packed, &nodes, &need}; }
};
GpuBvhBuilder::GpuBvhBuilder() : m(new Impl) {}
GpuBvhBuilder::~GpuBvhBuilder() { release(); delete m; }
void GpuBvhBuilder::release() { for (GBuf* b : m->all()) b->release(); if (m->h_counts) { (void)hipHostFree(m->h_counts); m->h_counts = nullptr; } m->staging.release(); }
const Node8GPU* GpuBvhBuilder::nodes() const { return m->nodes.as<Node8GPU>(); }

#define GBCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return std::string("gpu build: ") + #call + ": " + hipGetErrorString(e_); } while (0)

std::string GpuBvhBuilder::build(hipStream_t st, const F4* d_objtris, const TriShade* d_shade, const InstGPU* d_insts, uint32_t n, const BvhBuildOptions& opt, TriGPU* d_tris_out, GpuBuildResult& R) {
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    R = GpuBuildResult();
    if (n < 2) return "gpu build: needs at least two triangles";
    if (n > 0x3fffffffu) return "gpu build: too many triangles";
    const int radius = opt.ploc_radius > 0 ? std::min(opt.ploc_radius, 64) : 16;
    const uint32_t stop_at = std::max(1u, opt.ploc_top);
    Impl& B = *m;
    const uint32_t top_cap = 2u * std::min(stop_at, n) + 16u, cap = 2u * n + top_cap;
    const uint32_t nb = (n + kB - 1) / kB;
    auto t0 = clk::now();
    GBCHK(B.bmn.ensure((size_t)n * 16)); GBCHK(B.bmx.ensure((size_t)n * 16)); GBCHK(B.bounds.ensure(32));
    for (int k = 0; k < 2; k++) { GBCHK(B.keys[k].ensure((size_t)n * 8)); GBCHK(B.vals[k].ensure((size_t)n * 4)); GBCHK(B.cl[k].ensure((size_t)n * 4)); GBCHK(B.src[k].ensure((size_t)n * 4)); }
    GBCHK(B.pmn.ensure((size_t)cap * 16)); GBCHK(B.pmx.ensure((size_t)cap * 16)); GBCHK(B.pleft.ensure((size_t)cap * 4)); GBCHK(B.pright.ensure((size_t)cap * 4)); GBCHK(B.pdp.ensure((size_t)cap * sizeof(WideDp)));
    GBCHK(B.nn.ensure((size_t)n * 4)); GBCHK(B.flags.ensure((size_t)n * 8)); GBCHK(B.offs.ensure((size_t)n * 8)); GBCHK(B.counts.ensure(16));
    if (!B.h_counts) GBCHK(hipHostMalloc((void**)&B.h_counts, 16, hipHostMallocDefault));
    // ---- boxes, keys, sort ----
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0x3f800000u /* scale starts at 1.0 */, 0u};
    GBCHK(B.staging.to_device(st, B.bounds.p, init, 32));
    uint32_t* d_bounds = B.bounds.as<uint32_t>();
    hipLaunchKernelGGL(k_gb_prims, dim3(nb), dim3(kB), 0, st, d_objtris, d_shade, d_insts, n, B.bmn.as<F4>(), B.bmx.as<F4>(), d_bounds, d_bounds + 6);
    hipLaunchKernelGGL(k_gb_morton, dim3(nb), dim3(kB), 0, st, B.bmn.as<F4>(), B.bmx.as<F4>(), n, d_bounds, B.keys[0].as<unsigned long long>(), B.vals[0].as<uint32_t>());
    GBCHK(hipGetLastError());
    uint32_t hb[8];
    GBCHK(B.staging.to_host(st, hb, B.bounds.p, 32)); GBCHK(hipStreamSynchronize(st));
    float scale; memcpy(&scale, &hb[6], 4); R.scale = scale;
    const float pad = 2e-6f * scale;                                    // the host build's bvh_pad (rtx_scene_host.cpp)
    R.ms_prims = ms_since(t0); t0 = clk::now();
    {
        size_t tmp = 0;
        GBCHK(rocprim::radix_sort_pairs(nullptr, tmp, B.keys[0].as<unsigned long long>(), B.keys[1].as<unsigned long long>(), B.vals[0].as<uint32_t>(), B.vals[1].as<uint32_t>(), (size_t)n, 0u, 63u, st));
        GBCHK(B.sort_tmp.ensure(tmp));
        GBCHK(rocprim::radix_sort_pairs(B.sort_tmp.p, tmp, B.keys[0].as<unsigned long long>(), B.keys[1].as<unsigned long long>(), B.vals[0].as<uint32_t>(), B.vals[1].as<uint32_t>(), (size_t)n, 0u, 63u, st));
    }
    Pool P{B.pmn.as<F4>(), B.pmx.as<F4>(), B.pleft.as<int32_t>(), B.pright.as<int32_t>(), B.pdp.as<WideDp>(), n};
    hipLaunchKernelGGL(k_gb_leaves, dim3(nb), dim3(kB), 0, st, B.vals[1].as<uint32_t>(), B.bmn.as<F4>(), B.bmx.as<F4>(), n, P, B.cl[0].as<int32_t>());
    GBCHK(hipGetLastError()); GBCHK(hipStreamSynchronize(st));
    R.ms_sort = ms_since(t0); t0 = clk::now();
    // ---- PLOC ----
    {
        size_t tmp = 0;
        GBCHK(rocprim::exclusive_scan(nullptr, tmp, B.flags.as<unsigned long long>(), B.offs.as<unsigned long long>(), 0ull, (size_t)n, rocprim::plus<unsigned long long>(), st));
        GBCHK(B.scan_tmp.ensure(tmp));
    }
    uint32_t mcl = n, next_id = n; int cur = 0;
    while (mcl > stop_at) {
        const uint32_t gb = (mcl + kB - 1) / kB;
        int32_t* cin = B.cl[cur].as<int32_t>(); int32_t* cout = B.cl[cur ^ 1].as<int32_t>();
        hipLaunchKernelGGL(k_ploc_nn, dim3(gb), dim3(kB), (size_t)(kB + 2 * radius) * 24, st, cin, (int)mcl, radius, P, B.nn.as<int32_t>());
        hipLaunchKernelGGL(k_ploc_flags, dim3(gb), dim3(kB), 0, st, B.nn.as<int32_t>(), (int)mcl, B.flags.as<unsigned long long>());
        size_t tmp = B.scan_tmp.bytes;
        GBCHK(rocprim::exclusive_scan(B.scan_tmp.p, tmp, B.flags.as<unsigned long long>(), B.offs.as<unsigned long long>(), 0ull, (size_t)mcl, rocprim::plus<unsigned long long>(), st));
        hipLaunchKernelGGL(k_ploc_merge, dim3(gb), dim3(kB), 0, st, cin, B.nn.as<int32_t>(), B.flags.as<unsigned long long>(), B.offs.as<unsigned long long>(), (int)mcl, next_id, P, pad, opt.tri_cost,
                           cout, B.counts.as<uint32_t>());
        GBCHK(hipGetLastError());
        GBCHK(B.staging.to_host(st, B.h_counts, B.counts.p, 8)); GBCHK(hipStreamSynchronize(st));
        const uint32_t left = B.h_counts[0], merges = B.h_counts[1];
        if (merges == 0 || left + merges != mcl) return "gpu build: a PLOC round made no progress";
        mcl = left; next_id += merges; cur ^= 1; R.ploc_iterations++;
        if (R.ploc_iterations > 4096) return "gpu build: PLOC does not converge";
    }
    R.ms_ploc = ms_since(t0); t0 = clk::now();
    R.clusters_top = mcl;
    // ---- the top of the tree on the host ----
    int32_t root;
    if (mcl == 1) { GBCHK(B.staging.to_host(st, B.h_counts, B.cl[cur].p, 4)); GBCHK(hipStreamSynchronize(st)); root = (int32_t)B.h_counts[0]; }
    else {
        GBCHK(B.cids.ensure((size_t)mcl * 4)); GBCHK(B.cboxes.ensure((size_t)mcl * 24)); GBCHK(B.crecs.ensure((size_t)mcl * sizeof(WideDp)));
        hipLaunchKernelGGL(k_gb_clusters, dim3((mcl + kB - 1) / kB), dim3(kB), 0, st, B.cl[cur].as<int32_t>(), mcl, P, B.cids.as<int32_t>(), B.cboxes.as<float>(), B.crecs.as<WideDp>());
        GBCHK(hipGetLastError());
        std::vector<int32_t> ids(mcl); std::vector<float> boxes((size_t)mcl * 6); std::vector<WideDp> recs(mcl);
        GBCHK(B.staging.to_host(st, ids.data(), B.cids.p, (size_t)mcl * 4)); GBCHK(B.staging.to_host(st, boxes.data(), B.cboxes.p, (size_t)mcl * 24));
        GBCHK(B.staging.to_host(st, recs.data(), B.crecs.p, (size_t)mcl * sizeof(WideDp))); GBCHK(hipStreamSynchronize(st));
        std::vector<ClusterTopNode> top;
        build_cluster_top(boxes.data(), mcl, opt, top);
        const uint32_t nt = (uint32_t)top.size();
        if (nt == 0 || next_id + nt > cap) return "gpu build: top tree does not fit the node pool";
        // records of the top nodes (children have larger indices: one reverse sweep) and their pool form
        std::vector<WideDp> tdp(nt); std::vector<F4> tmn(nt), tmx(nt); std::vector<int32_t> tl(nt), tr(nt);
        auto child = [&](int32_t c, WideDpChild& out, int32_t& pool_id) {
            if (c >= 0) { WBox w; for (int a = 0; a < 3; a++) { w.mn[a] = top[c].mn[a]; w.mx[a] = top[c].mx[a]; } out.area = wbox_area(padded(w, pad)); out.leaf_cnt = 0; out.dp = &tdp[c]; pool_id = (int32_t)(next_id + (uint32_t)c); }
            else { const uint32_t k = (uint32_t)~c; WBox w; for (int a = 0; a < 3; a++) { w.mn[a] = boxes[(size_t)k * 6 + a]; w.mx[a] = boxes[(size_t)k * 6 + 3 + a]; } out.area = wbox_area(padded(w, pad));
                   pool_id = ids[k]; if ((uint32_t)ids[k] < n) { out.leaf_cnt = 1; out.dp = nullptr; } else { out.leaf_cnt = 0; out.dp = &recs[k]; } }
        };
        for (uint32_t t = nt; t-- > 0;) {
            WideDpChild cl_, cr_; child(top[t].left, cl_, tl[t]); child(top[t].right, cr_, tr[t]);
            auto cbox = [&](int32_t c) { WBox w; if (c >= 0) { for (int a = 0; a < 3; a++) { w.mn[a] = top[c].mn[a]; w.mx[a] = top[c].mx[a]; } } else { const uint32_t k = (uint32_t)~c; for (int a = 0; a < 3; a++) { w.mn[a] = boxes[(size_t)k * 6 + a]; w.mx[a] = boxes[(size_t)k * 6 + 3 + a]; } } return w; };
            wide_dp_combine(cl_, cr_, wbox_area(wbox_union(padded(cbox(top[t].left), pad), padded(cbox(top[t].right), pad))), 1.0, opt.tri_cost, tdp[t]);
            tmn[t] = {top[t].mn[0], top[t].mn[1], top[t].mn[2], 0.0f}; tmx[t] = {top[t].mx[0], top[t].mx[1], top[t].mx[2], 0.0f};
        }
        GBCHK(B.staging.to_device(st, P.mn + next_id, tmn.data(), (size_t)nt * 16)); GBCHK(B.staging.to_device(st, P.mx + next_id, tmx.data(), (size_t)nt * 16));
        GBCHK(B.staging.to_device(st, P.left + next_id, tl.data(), (size_t)nt * 4)); GBCHK(B.staging.to_device(st, P.right + next_id, tr.data(), (size_t)nt * 4));
        GBCHK(B.staging.to_device(st, P.dp + next_id, tdp.data(), (size_t)nt * sizeof(WideDp)));
        GBCHK(hipStreamSynchronize(st));                                 // (the sources are locals)
        root = (int32_t)next_id;
    }
    R.ms_top_host = ms_since(t0); t0 = clk::now();
    // ---- layout, level by level ----
    GBCHK(B.codes.ensure((size_t)n * 32)); GBCHK(B.meta.ensure((size_t)n * 8)); GBCHK(B.packed.ensure((size_t)n * 8)); GBCHK(B.nodes.ensure((size_t)n * sizeof(Node8GPU))); GBCHK(B.need.ensure((size_t)n * 4));
    GBCHK(B.staging.to_device(st, B.src[0].p, &root, 4)); GBCHK(hipStreamSynchronize(st));
    DAcc A{P, pad};
    uint32_t level_cnt = 1, node_base = 0, tri_total = 0; int sc = 0;
    R.level_start8.clear(); R.level_start8.push_back(0);
    while (level_cnt) {
        if ((size_t)node_base + level_cnt > n) return "gpu build: more wide nodes than triangles";
        const uint32_t gb = (level_cnt + kB - 1) / kB;
        hipLaunchKernelGGL(k_lay_count, dim3(gb), dim3(kB), 0, st, B.src[sc].as<int32_t>(), level_cnt, A, B.codes.as<int32_t>(), B.meta.as<uint2>(), B.packed.as<unsigned long long>());
        size_t tmp = B.scan_tmp.bytes;
        GBCHK(rocprim::exclusive_scan(B.scan_tmp.p, tmp, B.packed.as<unsigned long long>(), B.offs.as<unsigned long long>(), 0ull, (size_t)level_cnt, rocprim::plus<unsigned long long>(), st));
        hipLaunchKernelGGL(k_lay_emit, dim3(gb), dim3(kB), 0, st, B.codes.as<int32_t>(), B.meta.as<uint2>(), B.packed.as<unsigned long long>(), B.offs.as<unsigned long long>(), level_cnt, node_base, tri_total, P,
                           B.nodes.as<Node8GPU>(), B.src[sc ^ 1].as<int32_t>(), d_tris_out, B.counts.as<uint32_t>());
        GBCHK(hipGetLastError());
        GBCHK(B.staging.to_host(st, B.h_counts, B.counts.p, 8)); GBCHK(hipStreamSynchronize(st));
        node_base += level_cnt; R.level_start8.push_back(node_base);
        tri_total += B.h_counts[1]; level_cnt = B.h_counts[0]; sc ^= 1;
        if (tri_total > n || R.level_start8.size() > 256) return "gpu build: layout out of bounds";
    }
    if (tri_total != n) return "gpu build: the layout lost triangles (" + std::to_string(tri_total) + " of " + std::to_string(n) + ")";
    R.nnodes8 = node_base; R.ntris8 = tri_total;
    for (size_t l = R.level_start8.size() - 1; l-- > 0;) {
        const uint32_t first = R.level_start8[l], count = R.level_start8[l + 1] - first;
        if (count) hipLaunchKernelGGL(k_lay_need, dim3((count + kB - 1) / kB), dim3(kB), 0, st, B.nodes.as<Node8GPU>(), first, count, B.need.as<uint32_t>());
    }
    GBCHK(hipGetLastError());
    GBCHK(B.staging.to_host(st, B.h_counts, B.need.p, 4)); GBCHK(hipStreamSynchronize(st));
    R.stack8 = B.h_counts[0];
    R.ms_layout = ms_since(t0);
    return "";
}

// ---- flatten (rtx_build.hpp) ----
__global__ void __launch_bounds__(256) k_flatten(const float* __restrict__ verts7, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ matids, uint32_t nmatids,
                                                 const FlatInst* __restrict__ insts, uint32_t ninst, uint32_t ntri, F4* __restrict__ objtris, TriShade* __restrict__ shade) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= ntri) return;
    uint32_t lo = 0, hi = ninst - 1u;                       // the LAST instance whose first triangle is <= g (instances without triangles share a base with their successor)
    while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (insts[mid].tri_base <= g) lo = mid; else hi = mid - 1u; }
    const FlatInst F = insts[lo];
    const uint32_t t = g - F.tri_base;
    const uint32_t vi[3] = {idx[(size_t)F.idx_base + 3u * t], idx[(size_t)F.idx_base + 3u * t + 1u], idx[(size_t)F.idx_base + 3u * t + 2u]};
    f3 p[3], n[3];
    for (int k = 0; k < 3; k++) {
        const float* v = verts7 + ((size_t)F.vert_base + vi[k]) * 7u;
        p[k] = mk3(v[0], v[1], v[2]); n[k] = mk3(v[3], v[4], v[5]);
        objtris[(size_t)g * 3u + k] = F4{p[k].x, p[k].y, p[k].z, 0.0f};
    }
    TriShade s;
    const uint32_t mi = F.matid_base + 3u * t;              // == 3*PrimitiveIndex() + uint(v0.normal.w), Hit_v6.hlsl:16-17
    s.mat = mi < nmatids ? matids[mi] : kMissMat;
    s.inst = lo;
    const f3 cr = cross(p[1] - p[0], p[2] - p[0]);          // :28-30
    s.area = fabsf(length(cr) * 0.5f);                      // :31
    const f3 flat = normalize(cr);                          // :32
    s.flat[0] = flat.x; s.flat[1] = flat.y; s.flat[2] = flat.z;
    float* dst[3] = {s.n0, s.n1, s.n2};
    for (int k = 0; k < 3; k++) {                           // :40-46 (all(n != 0) is per component)
        const f3 use = (n[k].x != 0.0f && n[k].y != 0.0f && n[k].z != 0.0f) ? n[k] : flat;
        dst[k][0] = use.x; dst[k][1] = use.y; dst[k][2] = use.z;
    }
    s.guard_tau = 0.0f;
    shade[g] = s;
}
void launch_flatten(hipStream_t st, const float* verts7, const uint32_t* idx, const uint32_t* matids, uint32_t nmatids, const FlatInst* insts, uint32_t ninst, uint32_t ntri,
                    F4* objtris_out, TriShade* shade_out) {
    if (!ntri || !ninst) return;
    hipLaunchKernelGGL(k_flatten, dim3((ntri + 255u) / 256u), dim3(256), 0, st, verts7, idx, matids, nmatids, insts, ninst, ntri, objtris_out, shade_out);
}

}  // namespace rtx
