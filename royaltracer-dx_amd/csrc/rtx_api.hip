// rtx_api.hip — the C-ABI of include/rtx.h: context, scene upload, the wavefront render loop.
// Host code only (kernels are in rtx_kernels.hip).  No CPU rendering path exists here by design.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rtx.h"
#include "rtx_kernels.hpp"
#include "rtx_scene_host.hpp"
#include <mutex>
#include "rtx_build.hpp"
#include "rtx_staging.hpp"

using namespace rtx;

// message of the calls that have no context to hold one (rtx_create failing, rtx_shard_slab_bytes, rtx_restir_state_slab_bytes): PER THREAD, so the N threads of
// the native multi-GPU frame (host/MultiGpu.cpp) that ask for their slab sizes at once never write the same string; rtx_last_error(NULL) reads the caller's own
static thread_local std::string g_create_err;

// RTX_DEBUG_POISON=<byte> in the environment (tooling: the hunt for reads of memory no kernel of the frame wrote): every fresh device allocation is filled with that byte —
// 255 makes stale floats NaN and stale indices huge, 127 large finite values — so that a result which depends on what a previous context (or process) left in HBM turns
// from a once-in-20 000 mismatch into a reproducible one.  Unset (the product): allocations stay as hipMalloc returns them.
static int poison_byte() { static const int b = [] { const char* e = getenv("RTX_DEBUG_POISON"); return e && *e ? atoi(e) & 255 : -1; }(); return b; }
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (!n) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) { bytes = n; if (poison_byte() >= 0) { e = hipMemset(p, poison_byte(), n); if (e == hipSuccess) e = hipDeviceSynchronize(); } }     // (the fill runs on the null stream, the context's streams are non-blocking: join before anything is uploaded)
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct TimedLaunch { int cls; hipEvent_t a, b; };

// Streams are BORROWED from a process-wide pool and returned idle; the library never calls hipStreamDestroy.  Round 5 (profiles/r05_determinism.md): in a process that creates
// and destroys thousands of contexts, once in ~700 contexts two words of a live 912-byte heap block — a mesh's index array, the builder's leaf order — changed during a later
// rtx_commit_scene: a write through a stale pointer by code OUTSIDE this library (with the library's own allocations of that size on fenced pages, nothing of ours touched freed
// memory and nothing of ours was hit).  Not releasing events, device or pinned memory left the rate unchanged; not destroying the two streams of a context made it vanish
// (0 findings in 5 500 x 2 contexts against 25 in 17 700 x 2).  A pooled stream also saves the ~50 us its creation costs.
// A context borrows a SET of five streams (its own, the shadow-overlap stream, ReSTIR lanes 1 .. 3) that were created back to back: the runtime spreads streams over its
// (four) hardware queues in creation order, so the streams of one set run concurrently — two streams picked from a pool one by one may share a queue and serialise
// (measured: the two-lane ReSTIR frame of the atrium 8.08 -> 9.82 ms with single pooled streams, kernel times unchanged).
struct StreamSet { int device = -1; hipStream_t s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
struct StreamPool {
    std::mutex mu; std::vector<StreamSet> idle;
    hipError_t acquire(int device, StreamSet& out) {        // the caller has the device bound
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < idle.size(); i++) if (idle[i].device == device) { out = idle[i]; idle.erase(idle.begin() + (long)i); return hipSuccess; }
        out = StreamSet(); out.device = device;
        for (hipStream_t& st : out.s) { const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); if (e != hipSuccess) return e; }
        return hipSuccess;
    }
    void release(StreamSet& set) {
        if (set.device < 0) return;
        for (hipStream_t st : set.s) if (st) (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> g(mu); idle.push_back(set); set = StreamSet();
    }
};
static StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }      // (never destructed: no order of static destructors to get wrong at exit)

struct rtx_ctx {
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    int num_cus = 256;
    SceneHost host; BuiltScene built;
    bool committed = false, camera_set = false;
    DevBuf d_nodes, d_tris, d_small, d_small_tris, d_small_poly, d_shade, d_mats, d_insts, d_lights, d_cdf, d_cam;
    bool committed_once = false;
    // the wide tree as the DEVICE holds it (the host mirror B.nodes8 / B.tris8 is empty after a GPU build): counts, the root record (octant-sort grid), who built it
    uint32_t n_nodes8 = 0, n_tris8 = 0; Node8GPU root8{}; bool dev_built = false; int gpu_build = 0; GpuBvhBuilder* builder = nullptr; GpuBuildResult build_info;
    // RTX_OPT_GPU_BUILD: the meshes as they were handed over, resident on the device (append-only like the host's list: a commit uploads only what was added since the last one),
    // and the per-instance ranges k_flatten reads (csrc/rtx_build.hip)
    DevBuf d_pool_verts, d_pool_idx, d_pool_matids, d_flat_insts; size_t pool_verts = 0, pool_idx = 0, pool_matids = 0, pool_meshes = 0; std::vector<uint32_t> pool_vert_base, pool_idx_base; std::vector<FlatInst> h_flat;
    Staging staging;                                            // the two pinned chunks every copy from / to caller memory passes through (rtx_staging.hpp)
    std::vector<float> h_cdf; std::vector<uint32_t> h_one;     // host sources of small asynchronous uploads
    DevBuf d_inst_moved, d_tri_dirty, d_node_dirty; bool node_aabb_valid = false; int partial_refit = 1;     // partial GPU refit (RTX_OPT_PARTIAL_REFIT): node_aabb / d_scale hold the last full refit's state
    DevBuf d_objtris, d_node_aabb, d_scale;          // GPU refit: object-space vertices (uploaded on first use), per-node float boxes, max |coordinate|
    bool gpu_refit = true, device_scene_valid = false, objtris_uploaded = false;
    uint32_t refill_min = 12, trace_sched = 6, sort_materials = 0, blocks_per_cu = 0 /* 0 = auto */, occluder_cache = 0; int shade_dense = 0;       // persistent-traversal knobs (RTX_OPT_REFILL_MIN, RTX_OPT_TRACE_SCHED)
    DevScene dsc{};
    float view[16], proj[16];
    // path state
    DevBuf d_hitmask, d_order, d_pmask;
    // RTX_OPT_ASYNC: what finish_render needs of the frame that rtx_render enqueued
    struct Pending { bool active = false; size_t ncnt = 0; uint32_t nbatches = 0, G = 0, mb = 0, nee = 0, nee1 = 1; bool fused = false, fused_bvh = false; } pending;
    bool async = false;
    int node_stride = 0; DevBuf d_nodes_wide; bool wide_nodes = false;
    uint32_t lds_nodes_closest = 0; int lds_closest_opt = -1;     // RTX_OPT_LDS_NODES_CLOSEST        // RTX_OPT_NODE_STRIDE
    int restir_keys = 1; DevBuf d_rs_key_a, d_rs_key_b;  // RTX_OPT_RESTIR_KEYS
    int sample_interleave = 1;                       // RTX_OPT_SAMPLE_INTERLEAVE
    int octant_sort = 0; DevBuf d_oct[2], d_perm;    // RTX_OPT_OCTANT_SORT (2 = tooling: all keys zero, i.e. the machinery's overhead without a re-ordering)
    bool trace_counters = false; DevBuf d_trace_cnt;      // RTX_OPT_TRACE_COUNTERS
    uint32_t stack_cap = 11; DevBuf d_stack_ovf;           // RTX_OPT_STACK_CAP: traversal-stack entries kept in LDS (0 = all of them); the overflow columns of deeper trees
    int any_order_opt = -1;         // RTX_OPT_ANYHIT_ORDER: -1 = what the commit-time probe chose (BuiltScene::any_order)
    bool lpt_order = true;          // RTX_OPT_LPT_ORDER: fused kernels take their sub-queues longest first
    // ReSTIR work lists (x | y << 16 per pixel, 8 x 8 pixel blocks in MORTON order so that consecutive chunks are compact screen regions): the shard's own pixels
    // (pass 3) and — on shards — its tiles dilated by the 20-px radius of the spatial pass (passes 1 and 2); key = (width, height, tile, rank, count, deal)
    DevBuf d_halo, d_own; uint32_t halo_count = 0, own_count = 0; uint32_t halo_key[6] = {0, 0, 0, 0, 0, 0};
    // where this context holds last frame's ReSTIR history (pixel rectangle, exclusive upper bounds): the whole image after a reset / an unsharded frame / rtx_restir_unpack_state,
    // the own rectangle after a sharded frame, + halo_px after rtx_restir_unpack_halo; hist_all = the whole image whatever its size
    uint32_t hist[4] = {0, 0, 0, 0}; bool hist_all = true;
    bool bounce_ring = true;        // RTX_OPT_BOUNCE_VARIANT
    bool fused_bvh = false;         // RTX_OPT_FUSED_BVH: general path = one k_bounce_bvh launch per batch (trace -> shade -> shadow per sub-queue and bounce); measured SLOWER, default off
    DevBuf d_hitq;
    bool work_stealing = false;     // RTX_OPT_WORK_STEALING: trace kernels of general scenes continue with other sub-queues instead of draining (refill_steal); measured SLOWER, default off
    DevBuf d_heads;                 // per trace launch of a batch: G fetch cursors + the retired count
    bool compact_state = true;      // RTX_OPT_COMPACT_STATE: separate-kernel path keeps ray / throughput / hit records by queue position, ping-pong (DevPaths::out_*)
    DevBuf d_alt_o, d_alt_d, d_alt_thr;
    // RTX_OPT_MERGE_RAYS: thin launches of the traversal kernels take several sub-queues per workgroup (MergedQ).  The host cannot see a launch's ray count (it is on the
    // device), so it predicts it from the counters of the previous rtx_render of this context: per path entering the batch, how many were still alive at bounce b and how many
    // shadow rays slot j of bounce b cast.  A wrong prediction costs time only.
    bool taper = true; uint32_t taper_levels = 4;             // RTX_OPT_TAPER
    uint32_t merge_rays = 1024; uint64_t pred_paths = 0; std::vector<uint64_t> pred_q, pred_s; uint32_t pred_nee1 = 0;
    bool overlap_shadow = true; hipStream_t aux = nullptr;       // RTX_OPT_OVERLAP_SHADOW: k_trace_shadow of bounce b on a second stream, beside k_trace_closest of bounce b + 1 (not while kernels are timed)
    DevBuf d_ray_o, d_ray_d, d_thr, d_rad, d_hit, d_sh_o, d_sh_d, d_sh_c, d_queue[2], d_counters;
    uint32_t* h_counters = nullptr; size_t h_counters_words = 0;
    // accumulation
    DevBuf d_accum; void* ext_accum = nullptr; size_t ext_accum_bytes = 0; uint32_t acc_w = 0, acc_h = 0;
    DevBuf d_srgb, d_res_di, d_res_gi, d_sdata, d_last_di, d_last_gi, d_last_sd, d_p1cnt, d_p1scratch; size_t p1_slots = 0, last_slots = 0;
    float prev_view[16], prev_proj[16];
    // wavefront ReSTIR (rtx_restir_wave.hpp): path state by queue position (two sets), hit records, per-item records, the any-hit ray queue, queue lengths
    bool restir_wave = true;        // RTX_OPT_RESTIR_WAVEFRONT
    uint32_t restir_chunks = 4;     // RTX_OPT_RESTIR_CHUNKS: 256-item chunks per sub-queue (= workgroup) of the ReSTIR stages
    struct RsArea { DevBuf state, hit, cls, fin, cold, occ, cand, sho, shd, pay, cnt; } rs_area[4];      // one per pipeline lane (RTX_OPT_RESTIR_LANES)
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};      // lanes 1 .. 3 (lane 0 runs on the context's stream)
    StreamSet streams;              // borrowed from the process-wide pool: [0] the context's own stream, [1] aux, [2 .. 4] the lanes
    uint32_t restir_lane_min = 1u << 16;   // RTX_OPT_RESTIR_LANE_MIN: pixel lists shorter than this run as one chain
    uint32_t restir_lanes = 2;      // RTX_OPT_RESTIR_LANES: the work list of a ReSTIR frame as 1 .. 4 independent parts on as many streams (the tails of one part's many short launches fill with the others' work)
    // options
    bool timing = false; uint64_t paths_per_batch = 128u << 20; int lds_nodes_opt = -1; bool small_scene = true; bool fused = true; int stack_private = -1;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<TimedLaunch> timed;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    rtx_stats stats{};
    std::string err;
    F4* accum_ptr() { return (F4*)(ext_accum ? ext_accum : d_accum.p); }
};

#define HIPCHK(c, call)                                                                          \
    do { hipError_t e_ = (call);                                                                 \
         if (e_ != hipSuccess) { (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);  \
                                 return e_ == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP; } } while (0)
// RTX_OPT_ASYNC: an rtx_render that only ENQUEUED its frame leaves statistics to be collected (finish_render: stream sync + counter read-back).  Every entry point joins
// first (BIND) — except the calls a frame's epilogue is made of, which must stay stream-ordered behind the render without a host join (BIND_NOWAIT: pack / unpack)
static int finish_render(rtx_ctx* c);
#define BIND_NOWAIT(c) do { if (!(c)) return RTX_ERR_INVALID; HIPCHK(c, hipSetDevice((c)->device)); } while (0)
#define BIND(c) do { BIND_NOWAIT(c); if ((c)->pending.active) { const int r_ = finish_render(c); if (r_ != RTX_OK) return r_; } } while (0)

// EVERY copy between host arrays and the device goes through these two (rtx_staging.hpp: pinned chunks of the context).  to_device: the source is consumed when it returns
// and the copy is ordered on the context's stream — no lifetime rule, no synchronise.  to_host: complete when it returns (it waits for the stream up to the copy).
#define TO_DEVICE(c, dst, src, bytes) HIPCHK(c, (c)->staging.to_device((c)->stream, (dst), (src), (bytes)))
#define TO_HOST(c, dst, src, bytes) HIPCHK(c, (c)->staging.to_host((c)->stream, (dst), (src), (bytes)))
template <class T> static int upload(rtx_ctx* c, DevBuf& b, const std::vector<T>& v) {
    HIPCHK(c, b.ensure(v.size() * sizeof(T)));
    TO_DEVICE(c, b.p, v.data(), v.size() * sizeof(T));
    return RTX_OK;
}

extern "C" {

int rtx_create(int device_ordinal, rtx_ctx** out) {
    if (!out) return RTX_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                       " (this library has no CPU fallback)";
        return RTX_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) { g_create_err = "device ordinal out of range"; return RTX_ERR_NO_DEVICE; }
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { g_create_err = hipGetErrorString(e); return RTX_ERR_NO_DEVICE; }
    rtx_ctx* c = new rtx_ctx();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if ((e = stream_pool().acquire(device_ordinal, c->streams)) != hipSuccess) {
        g_create_err = hipGetErrorString(e); delete c; return RTX_ERR_HIP;
    }
    c->stream = c->streams.s[0]; c->own_stream = true;
    (void)hipEventCreate(&c->ev_begin); (void)hipEventCreate(&c->ev_end);
    *out = c;
    return RTX_OK;
}

void rtx_destroy(rtx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // a caller-owned stream (rtx_set_stream) may already be gone (torch destroys its streams first): every entry point that enqueues on it
    // synchronises before it returns or documents that it only enqueues, so only the context's own stream is drained here
    if (c->own_stream && c->stream) (void)hipStreamSynchronize(c->stream);
    else (void)hipDeviceSynchronize();
    DevBuf* all[] = {&c->d_nodes, &c->d_tris, &c->d_small, &c->d_small_tris, &c->d_small_poly, &c->d_objtris, &c->d_node_aabb, &c->d_scale, &c->d_shade, &c->d_mats, &c->d_insts, &c->d_lights, &c->d_cdf, &c->d_cam, &c->d_ray_o, &c->d_ray_d,
                     &c->d_thr, &c->d_rad, &c->d_hit, &c->d_hitmask, &c->d_order, &c->d_pmask, &c->d_sh_o, &c->d_sh_d, &c->d_sh_c, &c->d_queue[0], &c->d_queue[1], &c->d_counters,
                     &c->d_accum, &c->d_srgb, &c->d_res_di, &c->d_res_gi, &c->d_sdata, &c->d_last_di, &c->d_last_gi, &c->d_last_sd, &c->d_p1cnt, &c->d_p1scratch, &c->d_hitq, &c->d_halo, &c->d_own, &c->d_heads, &c->d_alt_o, &c->d_alt_d, &c->d_alt_thr, &c->d_oct[0], &c->d_oct[1], &c->d_perm, &c->d_trace_cnt,
                     &c->d_nodes_wide, &c->d_rs_key_a, &c->d_rs_key_b, &c->d_inst_moved, &c->d_tri_dirty, &c->d_node_dirty, &c->d_pool_verts, &c->d_pool_idx, &c->d_pool_matids, &c->d_flat_insts, &c->d_stack_ovf};
    for (auto& A : c->rs_area) for (DevBuf* b : {&A.state, &A.hit, &A.cls, &A.fin, &A.cold, &A.occ, &A.cand, &A.sho, &A.shd, &A.pay, &A.cnt}) b->release();
    for (DevBuf* b : all) b->release();
    delete c->builder; c->builder = nullptr;
    c->staging.release();
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    stream_pool().release(c->streams);
    delete c;
}

const char* rtx_last_error(rtx_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

static void pick_lds_closest(rtx_ctx* c);
int rtx_set_option(rtx_ctx* c, int option, int64_t value) {
    if (!c) return RTX_ERR_INVALID;
    switch (option) {
    case RTX_OPT_KERNEL_TIMING: c->timing = value != 0; return RTX_OK;
    case RTX_OPT_ASYNC: c->async = value != 0; return RTX_OK;
    case RTX_OPT_OCTANT_SORT: c->octant_sort = (int)value; return RTX_OK;
    case RTX_OPT_NODE_STRIDE: if (value != 0 && value != 80 && value != 128) { c->err = "node stride must be 0 (auto), 80 or 128"; return RTX_ERR_INVALID; } if (c->node_stride != (int)value) { c->node_stride = (int)value; c->committed = false; } return RTX_OK;
    case RTX_OPT_RESTIR_KEYS: c->restir_keys = value != 0; return RTX_OK;
    case RTX_OPT_SAMPLE_INTERLEAVE: c->sample_interleave = value != 0; return RTX_OK;
    case RTX_OPT_TRACE_COUNTERS:
        c->trace_counters = value != 0;
        if (c->trace_counters) { HIPCHK(c, c->d_trace_cnt.ensure(4 * sizeof(unsigned long long))); HIPCHK(c, hipMemsetAsync(c->d_trace_cnt.p, 0, 32, c->stream)); }
        c->dsc.trace_cnt = c->trace_counters ? (unsigned long long*)c->d_trace_cnt.p : nullptr;
        return RTX_OK;
    case RTX_OPT_BVH_REINSERT: if (value < 0 || value > 16) { c->err = "bvh_reinsert must be in [0, 16]"; return RTX_ERR_INVALID; } c->host.bvh.reinsert_passes = (int)value; c->host.topo_dirty = true; c->committed = false; return RTX_OK;
    case RTX_OPT_BVH_SPLIT: if (value < 0 || value > 1000000000) { c->err = "bvh_split must be in [0, 1e9] (parts per billion of the scene's surface area)"; return RTX_ERR_INVALID; } c->host.bvh.split_alpha = (double)value * 1e-9; c->host.topo_dirty = true; c->committed = false; return RTX_OK;
    case RTX_OPT_ANYHIT_ORDER: if (value < -1 || value > 2) { c->err = "anyhit_order must be -1 (probe), 0, 1 or 2"; return RTX_ERR_INVALID; } c->any_order_opt = (int)value; if (c->committed) { c->dsc.any_order = value < 0 ? c->built.any_order : (uint32_t)value; c->dsc.any_order_occ = value < 0 ? 0u : (uint32_t)value; } return RTX_OK;
    case RTX_OPT_PATHS_PER_BATCH: if (value < 4096) { c->err = "paths_per_batch must be >= 4096"; return RTX_ERR_INVALID; } c->paths_per_batch = (uint64_t)value; return RTX_OK;
    case RTX_OPT_SORT_MATERIALS: c->sort_materials = value != 0; c->dsc.sort_materials = c->sort_materials; return RTX_OK;
    case RTX_OPT_LDS_NODES: c->lds_nodes_opt = (int)value; c->committed = false; return RTX_OK;
    case RTX_OPT_PARTIAL_REFIT: c->partial_refit = value != 0; return RTX_OK;
    case RTX_OPT_LDS_NODES_CLOSEST: c->lds_closest_opt = (int)value; if (c->committed) pick_lds_closest(c); return RTX_OK;
    case RTX_OPT_SMALL_SCENE: c->small_scene = value != 0; c->committed = false; return RTX_OK;
    case RTX_OPT_FUSED_BOUNCE: c->fused = value != 0; return RTX_OK;
    case RTX_OPT_BOUNCE_VARIANT: c->bounce_ring = value == 0; return RTX_OK;     // 0 (default): LDS hit ring between trace and shading; 1: trace and shade the same 256 entries
    case RTX_OPT_STACK_PRIVATE: c->stack_private = (int)value; c->committed = false; return RTX_OK;
    case RTX_OPT_LPT_ORDER: c->lpt_order = value != 0; return RTX_OK;
    case RTX_OPT_FUSED_BVH: c->fused_bvh = value != 0; return RTX_OK;
    case RTX_OPT_WORK_STEALING: c->work_stealing = value != 0; return RTX_OK;
    case RTX_OPT_COMPACT_STATE: c->compact_state = value != 0; return RTX_OK;
    case RTX_OPT_OVERLAP_SHADOW: c->overlap_shadow = value != 0; return RTX_OK;
    case RTX_OPT_BLOCKS_PER_CU: if (value < 0 || value > 64) { c->err = "blocks_per_cu must be in [0, 64]"; return RTX_ERR_INVALID; } c->blocks_per_cu = (uint32_t)value; return RTX_OK;
    case RTX_OPT_TAPER: if (value < 0 || value > 8) { c->err = "taper must be in [0, 8]"; return RTX_ERR_INVALID; } c->taper = value != 0; c->taper_levels = value == 1 ? 4u : (uint32_t)std::max<long long>(value, 1); return RTX_OK;
    case RTX_OPT_MERGE_RAYS: if (value < 0 || value > (1 << 20)) { c->err = "merge_rays must be in [0, 2^20]"; return RTX_ERR_INVALID; } c->merge_rays = (uint32_t)value; return RTX_OK;
    case RTX_OPT_GPU_REFIT: c->gpu_refit = value != 0; return RTX_OK;
    case RTX_OPT_GPU_BUILD: c->gpu_build = value != 0; return RTX_OK;
    case RTX_OPT_STACK_CAP: if (value < 0 || value > 30 || (value > 0 && value < 4)) { c->err = "stack_cap must be 0 (the whole stack in LDS) or in [4, 30]"; return RTX_ERR_INVALID; } c->stack_cap = (uint32_t)value; c->committed = false; return RTX_OK;
    case RTX_OPT_SHADE_DENSE: c->shade_dense = (int)value; c->dsc.shade_dense = value > 0 ? 1u : 0u; return RTX_OK;
    case RTX_OPT_OCCLUDER_CACHE: c->occluder_cache = value != 0; c->dsc.occluder_cache = c->occluder_cache; return RTX_OK;
    case RTX_OPT_RESTIR_WAVEFRONT: c->restir_wave = value != 0; return RTX_OK;
    case RTX_OPT_RESTIR_LANE_MIN: if (value < 256 || value > (1ll << 31)) { c->err = "restir_lane_min must be in [256, 2^31]"; return RTX_ERR_INVALID; } c->restir_lane_min = (uint32_t)value; return RTX_OK;
    case RTX_OPT_RESTIR_LANES: if (value < 1 || value > 4) { c->err = "restir_lanes must be in [1, 4]"; return RTX_ERR_INVALID; } c->restir_lanes = (uint32_t)value; return RTX_OK;
    case RTX_OPT_RESTIR_CHUNKS: if (value < 1 || value > 64) { c->err = "restir_chunks must be in [1, 64]"; return RTX_ERR_INVALID; } c->restir_chunks = (uint32_t)value; return RTX_OK;
    case RTX_OPT_TRACE_SCHED: if (value > 7) { c->err = "trace_sched must be in [0, 7]"; return RTX_ERR_INVALID; } c->trace_sched = (uint32_t)value; c->dsc.trace_sched = c->trace_sched; return RTX_OK;
    case RTX_OPT_REFILL_MIN: if (value < 1 || value > 64) { c->err = "refill_min must be in [1, 64]"; return RTX_ERR_INVALID; } c->refill_min = (uint32_t)value; c->dsc.refill_min = c->refill_min; return RTX_OK;
    default: c->err = "unknown option"; return RTX_ERR_INVALID;
    }
}

int rtx_set_stream(rtx_ctx* c, void* s) {
    BIND(c);
    if (c->own_stream && c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    else HIPCHK(c, hipDeviceSynchronize());                // the old caller-owned stream may no longer exist: drain the device instead of touching it
    if (c->own_stream) { c->stream = nullptr; c->own_stream = false; }
    if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
    else { c->stream = c->streams.s[0]; c->own_stream = true; }
    return RTX_OK;
}

int rtx_set_materials(rtx_ctx* c, const void* mats128, uint32_t count) {
    if (!c) return RTX_ERR_INVALID;
    if (!c->host.set_materials(mats128, count)) { c->err = c->host.err; return RTX_ERR_INVALID; }
    c->committed = false; return RTX_OK;
}
int rtx_add_mesh(rtx_ctx* c, const void* verts28, uint32_t nverts, const uint32_t* indices, uint32_t nidx, const uint32_t* material_ids, uint32_t* mesh_out) {
    if (!c) return RTX_ERR_INVALID;
    if (!c->host.add_mesh(verts28, nverts, indices, nidx, material_ids, mesh_out)) { c->err = c->host.err; return RTX_ERR_INVALID; }
    c->committed = false; return RTX_OK;
}
int rtx_add_instance(rtx_ctx* c, uint32_t mesh, const float o2w[16], uint32_t* inst_out) {
    if (!c || !o2w) return RTX_ERR_INVALID;
    if (!c->host.add_instance(mesh, o2w, inst_out)) { c->err = c->host.err; return RTX_ERR_INVALID; }
    c->committed = false; return RTX_OK;
}
int rtx_set_instance_transform(rtx_ctx* c, uint32_t inst, const float o2w[16]) {
    if (!c || !o2w) return RTX_ERR_INVALID;
    if (!c->host.set_instance_transform(inst, o2w)) { c->err = c->host.err; return RTX_ERR_INVALID; }
    c->committed = false; return RTX_OK;
}

static int upload_lights(rtx_ctx* c) {         // the light records and their CDF as a dense float array (DevScene::cdf)
    const BuiltScene& B = c->built;
    std::vector<float>& cdf = c->h_cdf;           // (a member: the source of an asynchronous copy must outlive the call)
    cdf.resize(B.lights.size());
    for (size_t i = 0; i < cdf.size(); i++) cdf[i] = B.lights[i].cdf;
    int r = upload(c, c->d_lights, B.lights);
    if (r) return r;
    return upload(c, c->d_cdf, cdf);
}
static int upload_built(rtx_ctx* c) {          // every device array of a freshly built (or freshly loaded) scene
    BuiltScene& B = c->built;
    int r;
    if ((r = upload(c, c->d_nodes, B.nodes8))) return r;
    if ((r = upload(c, c->d_tris, B.tris8))) return r;
    c->n_nodes8 = (uint32_t)B.nodes8.size(); c->n_tris8 = (uint32_t)B.tris8.size(); c->dev_built = false;
    if (!B.nodes8.empty()) c->root8 = B.nodes8[0]; else memset(&c->root8, 0, sizeof(c->root8));
    if ((r = upload(c, c->d_shade, B.shade))) return r;
    if ((r = upload(c, c->d_small, B.small_recs))) return r;
    if ((r = upload(c, c->d_small_tris, B.small_tris))) return r;
    if ((r = upload(c, c->d_small_poly, B.small_poly))) return r;
    if ((r = upload(c, c->d_mats, B.mats))) return r;
    if ((r = upload(c, c->d_insts, B.insts))) return r;
    return upload_lights(c);
}
static int finalise_scene(rtx_ctx* c);

struct Scratch { DevBuf a, b, c; ~Scratch() { a.release(); b.release(); c.release(); } };

// a device array that only grows at its end: capacity in steps of 1.5 x, the `used` bytes survive a reallocation
static int grow_keep(rtx_ctx* c, DevBuf& b, size_t used, size_t need) {
    if (need <= b.bytes && b.p) return RTX_OK;
    DevBuf nb; HIPCHK(c, nb.ensure(std::max(need, b.bytes + b.bytes / 2)));
    if (used && b.p) HIPCHK(c, hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.release(); b = nb;
    return RTX_OK;
}
// the meshes added since the last commit -> the device pool; the per-instance ranges; then the flatten itself
static int flatten_on_device(rtx_ctx* c, uint32_t ntri) {
    const SceneHost& H = c->host;
    if (H.meshes.size() < c->pool_meshes) { c->pool_meshes = 0; c->pool_verts = c->pool_idx = c->pool_matids = 0; c->pool_vert_base.clear(); c->pool_idx_base.clear(); }     // (another scene: start over)
    size_t nv = c->pool_verts, ni = c->pool_idx;
    for (size_t m = c->pool_meshes; m < H.meshes.size(); m++) { nv += H.meshes[m].verts.size() / 7; ni += H.meshes[m].idx.size(); }
    if (nv > 0xFFFFFFFFull || ni > 0xFFFFFFFFull) { c->err = "commit: more than 2^32 vertices or indices"; return RTX_ERR_INVALID; }
    int r;
    if ((r = grow_keep(c, c->d_pool_verts, c->pool_verts * 28, nv * 28))) return r;
    if ((r = grow_keep(c, c->d_pool_idx, c->pool_idx * 4, ni * 4))) return r;
    if ((r = grow_keep(c, c->d_pool_matids, c->pool_matids * 4, H.matids.size() * 4))) return r;
    for (size_t m = c->pool_meshes; m < H.meshes.size(); m++) {
        const MeshHost& M = H.meshes[m];
        c->pool_vert_base.push_back((uint32_t)c->pool_verts); c->pool_idx_base.push_back((uint32_t)c->pool_idx);
        TO_DEVICE(c, (char*)c->d_pool_verts.p + c->pool_verts * 28, M.verts.data(), M.verts.size() * 4);
        TO_DEVICE(c, (char*)c->d_pool_idx.p + c->pool_idx * 4, M.idx.data(), M.idx.size() * 4);
        c->pool_verts += M.verts.size() / 7; c->pool_idx += M.idx.size();
    }
    c->pool_meshes = H.meshes.size();
    if (H.matids.size() > c->pool_matids) { TO_DEVICE(c, (char*)c->d_pool_matids.p + c->pool_matids * 4, H.matids.data() + c->pool_matids, (H.matids.size() - c->pool_matids) * 4); c->pool_matids = H.matids.size(); }
    c->h_flat.resize(H.insts.size());
    for (size_t ii = 0; ii < H.insts.size(); ii++) {
        const InstHost& in = H.insts[ii]; const MeshHost& M = H.meshes[in.mesh];
        c->h_flat[ii] = FlatInst{in.tri_base, (uint32_t)(M.idx.size() / 3), c->pool_vert_base[in.mesh], c->pool_idx_base[in.mesh], M.matid_base, {0u, 0u, 0u}};
    }
    if ((r = upload(c, c->d_flat_insts, c->h_flat))) return r;
    HIPCHK(c, c->d_objtris.ensure((size_t)ntri * 3 * sizeof(F4))); HIPCHK(c, c->d_shade.ensure((size_t)ntri * sizeof(TriShade)));
    launch_flatten(c->stream, (const float*)c->d_pool_verts.p, (const uint32_t*)c->d_pool_idx.p, (const uint32_t*)c->d_pool_matids.p, (uint32_t)H.matids.size(), (const FlatInst*)c->d_flat_insts.p,
                   (uint32_t)c->h_flat.size(), ntri, (F4*)c->d_objtris.p, (TriShade*)c->d_shade.p);
    HIPCHK(c, hipGetLastError());
    return RTX_OK;
}

// probe_anyhit_order (csrc/rtx_scene_host.cpp) for a tree the host holds no mirror of (RTX_OPT_GPU_BUILD): the same 2 048 NEE-like segments — a point on a random triangle to a
// CDF-sampled point on a light —, traced ON THE DEVICE in the three visiting orders by the counting form of the any-hit traversal, judged by the same cost model
static int probe_anyhit_order_on_device(rtx_ctx* c, uint32_t& best_out) {
    const BuiltScene& B = c->built;
    best_out = 0u;
    const uint32_t nt = B.built_tris;
    if (B.lights.empty() || !nt || c->h_flat.empty()) return RTX_OK;
    auto h32 = [](uint32_t a, uint32_t b) { uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15; return h; };
    auto r01 = [&](uint32_t a, uint32_t b) { return (float)(h32(a, b) >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> rays; rays.reserve(2048 * 8);
    for (uint32_t i = 0; i < 2048u; i++) {
        const uint32_t g = h32(i, 1u) % nt;
        size_t ii = (size_t)(std::upper_bound(c->h_flat.begin(), c->h_flat.end(), g, [](uint32_t v, const FlatInst& F) { return v < F.tri_base; }) - c->h_flat.begin()) - 1;      // the last instance starting at or before g
        const float* M = B.insts[ii].o2w; const MeshHost& mesh = c->host.meshes[c->host.insts[ii].mesh];
        const uint32_t t = g - c->h_flat[ii].tri_base;
        f3 w[3]; for (int k = 0; k < 3; k++) { const float* o = &mesh.verts[(size_t)mesh.idx[(size_t)t * 3 + k] * 7]; w[k] = xform_point(M, mk3(o[0], o[1], o[2])); }
        const f3 e1 = w[1] - w[0], e2 = w[2] - w[0];
        float u = r01(i, 2u), v = r01(i, 3u); if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
        const f3 p = mk3(w[0].x + u * e1.x + v * e2.x, w[0].y + u * e1.y + v * e2.y, w[0].z + u * e1.z + v * e2.z);
        f3 n = normalize(cross(e1, e2));
        const float xi = r01(i, 4u);
        size_t li = 0; while (li + 1 < B.lights.size() && B.lights[li].cdf < xi) li++;
        const LightGPU& Lg = B.lights[li];
        float a = r01(i, 5u), b = r01(i, 6u); if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; }
        const f3 lp = mk3(Lg.xv[0] + a * (Lg.yv[0] - Lg.xv[0]) + b * (Lg.zv[0] - Lg.xv[0]), Lg.xv[1] + a * (Lg.yv[1] - Lg.xv[1]) + b * (Lg.zv[1] - Lg.xv[1]), Lg.xv[2] + a * (Lg.yv[2] - Lg.xv[2]) + b * (Lg.zv[2] - Lg.xv[2]));
        f3 dir = lp - p;
        if (dot(n, dir) < 0.0f) n = mk3(-n.x, -n.y, -n.z);
        const f3 org = mk3(p.x + kSBias * n.x, p.y + kSBias * n.y, p.z + kSBias * n.z);
        dir = lp - org;
        const float dist = length(dir);
        if (!(dist > 10.0f * kSBias)) continue;
        const float r8[8] = {org.x, org.y, org.z, 0.5f * kSBias, dir.x / dist, dir.y / dist, dir.z / dist, dist - 5.0f * kSBias};
        rays.insert(rays.end(), r8, r8 + 8);
    }
    const uint32_t n = (uint32_t)(rays.size() / 8);
    if (!n) return RTX_OK;
    Scratch s;
    HIPCHK(c, s.a.ensure((size_t)n * 32)); HIPCHK(c, s.b.ensure((size_t)n * 16 * 3));
    TO_DEVICE(c, s.a.p, rays.data(), (size_t)n * 32);
    for (uint32_t ord = 0; ord < 3u; ord++) { DevScene sc = c->dsc; sc.any_order = ord; launch_dbg_trace(c->stream, sc, (const F4*)s.a.p, n, 3, (F4*)s.b.p + (size_t)ord * n); }
    HIPCHK(c, hipGetLastError());
    std::vector<float> h((size_t)n * 4 * 3);
    TO_HOST(c, h.data(), s.b.p, h.size() * 4);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double cost[3] = {0.0, 0.0, 0.0};
    for (uint32_t ord = 0; ord < 3u; ord++) for (uint32_t i = 0; i < n; i++) { const float* q = &h[((size_t)ord * n + i) * 4]; cost[ord] += (double)q[1] * (205.0 * 64.0 / 47.0) + (double)q[2] * (70.0 * 64.0 / 24.0); }
    for (uint32_t ord = 1; ord < 3u; ord++) if (cost[ord] < 0.95 * cost[0] && cost[ord] < cost[best_out]) best_out = ord;
    return RTX_OK;
}

int rtx_commit_scene(rtx_ctx* c) {
    BIND(c);
    if (c->host.topo_dirty || c->host.mats_dirty || !c->committed_once)      // (a transform-only commit changes neither the ids nor the table: not 11 M comparisons per frame)
        for (size_t i = 0; i < c->host.matids.size(); i++)
            if (c->host.matids[i] >= c->host.mats128.size() / 32) { c->err = "commit: material id out of range"; return RTX_ERR_INVALID; }
    c->committed_once = true;
    BuiltScene& B = c->built;
    int r;
    // Transform-only commit of a scene that is already resident (and not a tiny one, whose pre-test records depend on world
    // positions): refit ON THE GPU — the kernels re-derive the world triangles and re-quantise the wide nodes bottom-up; the host
    // only re-derives the instance matrices and the light list.  Anything else: host build (or host refit) + upload.
    const bool gpu_path = c->gpu_refit && c->device_scene_valid && !c->host.topo_dirty && B.small_nrec == 0 && c->n_nodes8 != 0 && B.level_start8.size() >= 2;
    if (gpu_path) {
        const bool mats_changed = c->host.mats_dirty;
        if (!c->host.refresh_transforms(B)) { c->err = c->host.err; return RTX_ERR_INVALID; }
        if (mats_changed && (r = upload(c, c->d_mats, B.mats))) return r;      // rtx_set_materials on a resident scene: new table beside the new light list
        if ((r = upload(c, c->d_insts, B.insts))) return r;
        if ((r = upload_lights(c))) return r;
        if (!c->objtris_uploaded) {
            if (B.objtris.empty()) c->host.fill_objtris(B);             // scene came from a cache file: derive them from the meshes now
            if ((r = upload(c, c->d_objtris, B.objtris))) return r; c->objtris_uploaded = true;
        }
        HIPCHK(c, c->d_node_aabb.ensure((size_t)c->n_nodes8 * 32));
        // the first refit after a build is a full one (it fills node_aabb); later ones touch the moved instances only, unless every instance moved anyway
        size_t nmoved = 0; for (uint32_t m : B.inst_moved) nmoved += m;
        const bool partial = c->partial_refit && c->node_aabb_valid && B.inst_moved.size() == B.insts.size() && nmoved < B.insts.size();
        if (partial) {
            if ((r = upload(c, c->d_inst_moved, B.inst_moved))) return r;
            HIPCHK(c, c->d_tri_dirty.ensure(c->n_tris8)); HIPCHK(c, c->d_node_dirty.ensure(c->n_nodes8));
        } else {
            c->h_one.assign(1, 0x3f800000u);                       // scale starts at 1.0 like the host's max(1, |coordinates|)
            if ((r = upload(c, c->d_scale, c->h_one))) return r;
        }
        launch_refit(c->stream, (Node8GPU*)c->d_nodes.p, B.level_start8.data(), (uint32_t)B.level_start8.size() - 1, (TriGPU*)c->d_tris.p, c->n_tris8,
                     (const TriShade*)c->d_shade.p, (const InstGPU*)c->d_insts.p, (const F4*)c->d_objtris.p, (F4*)c->d_node_aabb.p, (uint32_t*)c->d_scale.p,
                     partial ? (const uint32_t*)c->d_inst_moved.p : nullptr, (uint8_t*)c->d_tri_dirty.p, (uint8_t*)c->d_node_dirty.p);
        HIPCHK(c, hipGetLastError());
        c->node_aabb_valid = true;
    } else {
        c->device_scene_valid = false; c->objtris_uploaded = false; c->node_aabb_valid = false;
        size_t ntri_all = 0; for (const InstHost& in : c->host.insts) ntri_all += c->host.meshes[in.mesh].idx.size() / 3;
        // RTX_OPT_GPU_BUILD: the tree on the device (csrc/rtx_build.hip).  Not for tiny scenes (their pre-test records are built from the host tree's leaf order) nor with
        // spatial splits (a host-builder feature); there the host builds as before.
        const bool on_gpu = c->gpu_build && ntri_all > 4096u && c->host.bvh.split_alpha <= 0.0;
        if (!(on_gpu ? c->host.prepare_device_build(B) : c->host.build(B))) { c->err = c->host.err; return RTX_ERR_INVALID; }
        if (!on_gpu) { if ((r = upload_built(c))) return r; }
        else {
            const uint32_t nt = B.built_tris;
            if ((r = upload(c, c->d_mats, B.mats))) return r;
            if ((r = upload(c, c->d_insts, B.insts))) return r;
            if ((r = upload_lights(c))) return r;
            if ((r = flatten_on_device(c, nt))) return r; c->objtris_uploaded = true;          // object-space triangles + shade records, from the resident meshes
            for (DevBuf* b : {&c->d_small, &c->d_small_tris, &c->d_small_poly}) HIPCHK(c, b->ensure(16));
            HIPCHK(c, c->d_tris.ensure((size_t)nt * sizeof(TriGPU)));
            if (!c->builder) c->builder = new GpuBvhBuilder();
            BvhBuildOptions bo = c->host.bvh; if (bo.ploc_radius <= 0) bo.ploc_radius = 16;
            const std::string e = c->builder->build(c->stream, (const F4*)c->d_objtris.p, (const TriShade*)c->d_shade.p, (const InstGPU*)c->d_insts.p, nt, bo, (TriGPU*)c->d_tris.p, c->build_info);
            if (!e.empty()) { c->err = e; return RTX_ERR_HIP; }
            const GpuBuildResult& G = c->build_info;
            HIPCHK(c, c->d_nodes.ensure((size_t)G.nnodes8 * sizeof(Node8GPU)));
            HIPCHK(c, hipMemcpyAsync(c->d_nodes.p, c->builder->nodes(), (size_t)G.nnodes8 * sizeof(Node8GPU), hipMemcpyDeviceToDevice, c->stream));
            c->n_nodes8 = G.nnodes8; c->n_tris8 = G.ntris8; c->dev_built = true; B.bvh_pad = 2e-6f * G.scale;
            B.level_start8 = G.level_start8; B.stack8 = G.stack8;
            // the boxes: a FULL refit — world triangles from the object-space ones, every node quantised bottom-up (what a transform-only commit runs)
            HIPCHK(c, c->d_node_aabb.ensure((size_t)c->n_nodes8 * 32));
            c->h_one.assign(1, 0x3f800000u);
            if ((r = upload(c, c->d_scale, c->h_one))) return r;
            launch_refit(c->stream, (Node8GPU*)c->d_nodes.p, B.level_start8.data(), (uint32_t)B.level_start8.size() - 1, (TriGPU*)c->d_tris.p, c->n_tris8,
                         (const TriShade*)c->d_shade.p, (const InstGPU*)c->d_insts.p, (const F4*)c->d_objtris.p, (F4*)c->d_node_aabb.p, (uint32_t*)c->d_scale.p, nullptr, nullptr, nullptr);
            HIPCHK(c, hipGetLastError());
            TO_HOST(c, &c->root8, c->d_nodes.p, sizeof(Node8GPU));
            c->node_aabb_valid = true;
            if (getenv("RTX_BUILD_TIMES")) fprintf(stderr, "[build] GPU: prims %.2f ms, sort %.2f ms, PLOC %.2f ms (%u rounds -> %u clusters), top on the host %.2f ms, layout %.2f ms: %u wide nodes, stack %u\n",
                                                   G.ms_prims, G.ms_sort, G.ms_ploc, G.ploc_iterations, G.clusters_top, G.ms_top_host, G.ms_layout, G.nnodes8, G.stack8);
        }
    }
    r = finalise_scene(c);
    if (r == RTX_OK && c->dev_built && c->n_nodes8) {          // the visiting order of any-hit rays, probed on the device (the host probe replays its mirror of the tree)
        uint32_t best = 0;
        if ((r = probe_anyhit_order_on_device(c, best))) return r;
        c->built.any_order = best;
        if (c->any_order_opt < 0) c->dsc.any_order = best;
    }
    return r;
}

// the binary scene cache.  Save = the committed scene (inputs + everything rtx_commit_scene derived); load = replace the
// context's scene by the file's and upload it, instead of rtx_set_materials / rtx_add_mesh / rtx_add_instance / rtx_commit_scene.
int rtx_save_scene_cache(rtx_ctx* c, const char* path) {
    if (!c) return RTX_ERR_INVALID;
    if (!c->committed) { c->err = "save_scene_cache: scene not committed"; return RTX_ERR_STATE; }
    if (c->dev_built) { c->err = "save_scene_cache: the tree was built on the GPU (RTX_OPT_GPU_BUILD) and has no host mirror; commit with the host builder to save a cache"; return RTX_ERR_STATE; }
    if (!save_scene_cache(c->host, c->built, path, c->err)) return RTX_ERR_INVALID;
    return RTX_OK;
}
int rtx_load_scene_cache(rtx_ctx* c, const char* path) {
    BIND(c);
    if (!load_scene_cache(path, c->host, c->built, c->err)) return RTX_ERR_INVALID;     // on failure the previous scene is untouched
    c->committed = false; c->device_scene_valid = false; c->objtris_uploaded = false; c->node_aabb_valid = false;
    int r = upload_built(c);
    if (r) return r;
    return finalise_scene(c);
}

static void pick_lds_closest(rtx_ctx* c) {
    DevScene& s = c->dsc;
    // ... and the CLOSEST-HIT launches of the path tracer take the other side of that trade (round 4, after the queue order was tightened): with the first three levels of the
    // wide tree in LDS (73 nodes) and six workgroups per CU they run C3 20.5 -> 19.6 ms and C5 17.3 -> 17.0 ms per frame against eight / seven workgroups with 24 / 31 nodes,
    // while the shadow launches lose (11.1 -> 12.2 ms on C3): a closest-hit ray crosses the top of the tree at every step of its front-to-back walk, an any-hit ray leaves at
    // its first occluder.  So the count is per kind of launch (DevScene goes by value).  Fewer than six workgroups lose again (128 nodes on C3: 20.4 ms; 80 on C5: 17.6).
    // LDS granule: the measurements fit 1 KB (80 nodes on C5 "fit" six workgroups at 512 B and ran like five).
    c->lds_nodes_closest = 0;
    // In a frame the shadow launch of bounce b runs BESIDE the closest-hit launch of bounce b + 1 (RTX_OPT_OVERLAP_SHADOW), and six closest-hit workgroups of 25 KB leave it
    // no LDS on that CU: the street scene (30 MB of nodes, closest-hit kernel co-limited by memory, so the overlap is worth more there) LOSES 0.4 ms per frame with 73
    // nodes although the kernel alone gains 0.3; the atrium (2.2 MB of nodes) keeps 0.15-0.2 of the kernel's 0.9 ms (one context, option switched between rounds).  Auto therefore applies to trees that fit L2 (<= 16 MB,
    // the same line the wide node copy draws); RTX_OPT_LDS_NODES_CLOSEST sets it by hand.
    const bool small_tree = (size_t)s.nnodes * sizeof(Node8GPU) <= ((size_t)16 << 20);
    if (!s.nsmall && s.nnodes > s.lds_nodes && c->lds_nodes_opt < 0 && (c->lds_closest_opt >= 0 || small_tree)) {
        auto fit1k = [&](uint32_t nodes) { DevScene t = s; t.lds_nodes = nodes; return (160u * 1024u) / (uint32_t)((trace_lds_bytes(t) + 64 + 1023) & ~(size_t)1023); };
        uint32_t n = c->lds_closest_opt >= 0 ? std::min<uint32_t>((uint32_t)c->lds_closest_opt, s.nnodes) : std::min<uint32_t>(73u, s.nnodes);
        if (c->lds_closest_opt < 0) while (n > s.lds_nodes && fit1k(n) < 6u) n--;
        DevScene t = s; t.lds_nodes = n;
        if (n > s.lds_nodes && trace_lds_bytes(t) <= 64 * 1024) c->lds_nodes_closest = n;
    }
}
static int finalise_scene(rtx_ctx* c) {
    BuiltScene& B = c->built;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->device_scene_valid = true;
    DevScene& s = c->dsc;
    s.nodes = (const Node8GPU*)c->d_nodes.p; s.nnodes = c->n_nodes8;
    s.nodes_f = (const F4*)c->d_nodes.p; s.node_v4 = 5u;
    // RTX_OPT_NODE_STRIDE: a second copy of the nodes with ONE node per 128-B line (80-B nodes at an 80-B stride straddle a line in 4 of 8 positions: 1.5 lines per visit),
    // refreshed after every build / refit (stream order: before any frame).  Auto: made for trees of more than 16 MB, and fetched by the path tracer's closest-hit launches of
    // bounces >= 1 only — incoherent rays on a tree far larger than L2 gain (street scene, 3.8 M triangles: k_trace_closest -2.4 %), coherent ones (camera rays, ReSTIR's
    // stages) and the any-hit kernel like neighbours sharing lines (+1 %), small trees do not care (profiles/r04_node_stride_ab.md).  128: every traversal fetches the wide copy.
    c->wide_nodes = s.nnodes && (c->node_stride == 128 || (c->node_stride == 0 && (size_t)s.nnodes * sizeof(Node8GPU) > ((size_t)16 << 20)));
    if (c->wide_nodes) {
        HIPCHK(c, c->d_nodes_wide.ensure((size_t)s.nnodes * 128));
        HIPCHK(c, hipMemcpy2DAsync(c->d_nodes_wide.p, 128, c->d_nodes.p, sizeof(Node8GPU), sizeof(Node8GPU), s.nnodes, hipMemcpyDeviceToDevice, c->stream));
        if (c->node_stride == 128) { s.nodes_f = (const F4*)c->d_nodes_wide.p; s.node_v4 = 8u; }
    }
    s.tris = (const TriGPU*)c->d_tris.p; s.ntris = c->n_tris8;
    s.shade = (const TriShade*)c->d_shade.p;
    s.small = (const SmallRecPair*)c->d_small.p; s.small_tris = (const TriGPU*)c->d_small_tris.p; s.small_poly = (const F4*)c->d_small_poly.p; s.small_cm = B.small_cm; s.small_delta = B.small_delta;
    s.mats = (const MatGPU*)c->d_mats.p; s.nmat = (uint32_t)B.mats.size();
    s.insts = (const InstGPU*)c->d_insts.p; s.ninst = (uint32_t)B.insts.size();
    s.lights = (const LightGPU*)c->d_lights.p; s.nlights = (uint32_t)B.lights.size(); s.cdf = (const float*)c->d_cdf.p;
    s.total_weight = B.total_weight;
    // LDS budget per workgroup: stack + top of tree + first triangles, kept <= 64 KiB
    // exact bound of the 8-wide tree, no slack: a level adds ONE entry (the rest of its hit siblings) and only where a node has >= 2 internal
    // children (collapse_bvh8: need[]); a pop precedes every descent from an exhausted group.  Each entry costs 1.5 KB of LDS per workgroup (6 B per lane: kStackEntryBytes), and
    // LDS decides how many workgroups live on a CU: two entries of slack cost C3 2.3 % (5 instead of 6 workgroups) and C5 1.3 %.
    s.stack_depth = B.stack8;
    // RTX_OPT_STACK_CAP (round 5): LDS pays for `stack_cap` entries at most; a tree whose exact bound is deeper keeps its remaining entries in per-lane columns in global memory
    // (StackLdsT<true>, rtx_traverse.hpp).  The bound is reached by a handful of rays, the LDS it costs is paid by every workgroup as staged nodes (73 at a bound of 9, 44 at 11,
    // 24 at 12).  Measured (tools/frame_ms.py, hard street scene): GPU-built tree, bound 12: 40.2 -> 39.4 ms with a cap of 9; host-built, bound 11: 39.75 -> 40.0; the
    // street stand-in, bound 10: 30.2 -> 30.4 — the overflow test on every push and pop costs about what 30 more staged nodes bring, so the default cap of 11 only catches the
    // deep trees, for which it is also the difference between running and "BVH too deep for the LDS traversal stack".  Columns: 2^22 lanes (16 384 workgroups: more than any
    // launch of this library keeps resident) x 8 B per entry beyond the cap.
    s.stack_ovf = nullptr; s.stack_ovf_stride = 0;
    if (c->stack_cap && B.stack8 > c->stack_cap && B.stack8 <= 30) {
        const uint32_t stride = 1u << 22;
        HIPCHK(c, c->d_stack_ovf.ensure((size_t)(B.stack8 - c->stack_cap) * stride * 8));
        s.stack_depth = c->stack_cap; s.stack_ovf = (unsigned long long*)c->d_stack_ovf.p; s.stack_ovf_stride = stride;
    }
    s.stack_private = c->stack_private == 1 ? 1u : 0u;    // 1 (private / scratch) is a tuning knob; it measured slower than the LDS column
    if (s.stack_depth > 30) { c->err = "commit: BVH too deep for the traversal stack (more than 30 levels of 8-wide nodes with two or more internal children)"; return RTX_ERR_INVALID; }
    // LDS per workgroup = traversal stack (6 B per entry and lane) + top of the tree (+ all triangles of a small scene), <= 64 KiB.
    const size_t stack_bytes = (size_t)s.stack_depth * 256 * kStackEntryBytes;
    const size_t hard = 64 * 1024;
    size_t budget = hard > stack_bytes ? hard - stack_bytes : 0;
    uint32_t want_nodes;
    if (c->lds_nodes_opt >= 0) want_nodes = (uint32_t)c->lds_nodes_opt;
    else {
        want_nodes = 73;                                   // root + 8 + 64: the first three levels of the wide tree; trimmed below for occupancy
    }
    s.lds_nodes = std::min<uint32_t>(std::min<uint32_t>(want_nodes, s.nnodes), (uint32_t)(budget / 80));
    budget -= (size_t)s.lds_nodes * 80;
    uint32_t want_tris = s.ntris <= 256 ? s.ntris : 0u;                    // triangles only when ALL of them fit
    s.lds_tris = (size_t)want_tris * 48 <= budget ? want_tris : 0u;
    s.nsmall = 0; s.nsmall_occ = 0;
    s.any_order = c->any_order_opt < 0 ? B.any_order : (uint32_t)c->any_order_opt;
    s.any_order_occ = c->any_order_opt < 0 ? 0u : (uint32_t)c->any_order_opt;
    {   // grid of RTX_OPT_OCTANT_SORT 3 over the root's box: 8 bits handed to the axes one at a time, always to the axis whose cells are longest
        float ext[3] = {1.0f, 1.0f, 1.0f}; uint32_t bits[3] = {0, 0, 0};
        s.cell_o[0] = s.cell_o[1] = s.cell_o[2] = 0.0f;
        if (c->n_nodes8) {
            const Node8GPU& R0 = c->root8;
            s.cell_o[0] = R0.px; s.cell_o[1] = R0.py; s.cell_o[2] = R0.pz;
            for (int a = 0; a < 3; a++) ext[a] = std::max(1e-20f, 255.0f * std::ldexp(1.0f, (int)((R0.e_imask >> (8 * a)) & 0xffu) - 127));
        }
        for (int k = 0; k < 8; k++) { int best = 0; for (int a = 1; a < 3; a++) if (ext[a] / (float)(1u << bits[a]) > ext[best] / (float)(1u << bits[best])) best = a; bits[best]++; }
        for (int a = 0; a < 3; a++) s.cell_s[a] = (float)(1u << bits[a]) / ext[a];
        s.cell_bits = bits[0] | (bits[1] << 4) | (bits[2] << 8);
    }
    s.trace_cnt = c->trace_counters ? (unsigned long long*)c->d_trace_cnt.p : nullptr;
    s.refill_min = c->refill_min; s.trace_sched = c->trace_sched; s.sort_materials = c->sort_materials; s.occluder_cache = c->occluder_cache; s.shade_dense = c->shade_dense > 0 ? 1u : 0u;
    if (c->small_scene && B.small_nrec && B.small_tris.size() * 48 <= budget + (size_t)s.lds_tris * 48) {
        s.nsmall = B.small_nrec; s.nsmall_occ = B.small_nocc; s.lds_tris = (uint32_t)B.small_tris.size();   // LDS holds the records' triangles instead of the leaf-ordered ones
    }
    if (trace_lds_bytes(s) > 64 * 1024) { c->err = "commit: BVH too deep for the LDS traversal stack"; return RTX_ERR_INVALID; }
    // Staged nodes vs workgroups per CU.  The persistent traversal kernels are limited by LDS (160 KB per CU), and they gain from every workgroup
    // (`k_trace_shadow` C3: 12.8 -> 12.0 ms for one more) more than from nodes in LDS.  So: the workgroup count that root + 8 nodes alone would reach,
    // and then as many nodes as fit beside it.  Measured per frame: C3 41.5 ms with 9 nodes, 40.9 with 50 (eight workgroups either way,
    // `k_trace_closest` 21.9 -> 21.4 ms), 41.3 with 57-73 (seven); C5 40.4 ms with 73 nodes, 38.6 with 9, 38.3 with 31.  RTX_DEBUG_LDS=1 prints the choice.
    // (LDS is granted in 512-B granules, which the runtime's occupancy query does not count: 33 nodes on C5 "fit" seven workgroups by its answer and
    // ran like six.  Hence the model below, with the query only as the upper bound the registers set.)
    if (c->lds_nodes_opt < 0 && !s.nsmall && s.lds_nodes > 9u) {
        auto fit = [&](uint32_t nodes) { DevScene t = s; t.lds_nodes = nodes; return (160u * 1024u) / (uint32_t)((trace_lds_bytes(t) + 64 + 1023) & ~(size_t)1023); };      // (1-KB granule: what round 4's sweeps fit, profiles/r04_lds_closest_ab.md)
        DevScene t9 = s; t9.lds_nodes = 9;
        const int by_regs = trace_workgroups_per_cu(t9);
        const uint32_t target = std::min<uint32_t>(fit(9), by_regs > 0 ? (uint32_t)by_regs : 8u);
        while (s.lds_nodes > 9u && fit(s.lds_nodes) < target) s.lds_nodes--;
        if (getenv("RTX_DEBUG_LDS")) fprintf(stderr, "[rtx] stack_depth %u workgroups per CU %u, %u nodes staged, %zu B of LDS\n", s.stack_depth, target, s.lds_nodes, trace_lds_bytes(s));
    }
    pick_lds_closest(c);
    c->stats.bvh_refits = B.refit_count; c->stats.bvh_nodes = s.nnodes; c->stats.triangles = B.shade.empty() ? B.built_tris : (uint32_t)B.shade.size(); c->stats.bvh_refs = s.ntris; c->stats.lights = s.nlights; c->stats.materials = s.nmat;
    c->committed = true;
    return RTX_OK;
}

int rtx_set_camera(rtx_ctx* c, const float view[16], const float proj[16]) {
    BIND(c);
    if (!view || !proj) return RTX_ERR_INVALID;
    // previous-frame matrices for the temporal pass (m_prevViewMatrix / m_prevProjMatrix, Renderer.cpp:1738-1740, 1766-1767)
    if (c->camera_set) { memcpy(c->prev_view, c->view, 64); memcpy(c->prev_proj, c->proj, 64); }
    else { memcpy(c->prev_view, view, 64); memcpy(c->prev_proj, proj, 64); }
    memcpy(c->view, view, 64); memcpy(c->proj, proj, 64);
    CameraGPU cam;
    mat4_inverse(view, cam.viewI); mat4_inverse(proj, cam.projI);     // Renderer.cpp:1735-1736
    memcpy(cam.prev_view, c->prev_view, 64); memcpy(cam.prev_proj, c->prev_proj, 64);
    HIPCHK(c, c->d_cam.ensure(sizeof(cam)));
    TO_DEVICE(c, c->d_cam.p, &cam, sizeof(cam));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->camera_set = true;
    return RTX_OK;
}

int rtx_bind_accum(rtx_ctx* c, void* dev, size_t bytes) {
    if (!c) return RTX_ERR_INVALID;
    c->ext_accum = dev; c->ext_accum_bytes = dev ? bytes : 0;
    return RTX_OK;
}

static int ensure_accum(rtx_ctx* c, uint32_t w, uint32_t h, bool clear) {
    const size_t need = (size_t)w * h * 16;
    if (c->ext_accum) {
        if (c->ext_accum_bytes < need) { c->err = "bound accumulation buffer is smaller than width*height*16 bytes"; return RTX_ERR_INVALID; }
    } else {
        const bool fresh = !c->d_accum.p || c->acc_w != w || c->acc_h != h;
        HIPCHK(c, c->d_accum.ensure(need));
        clear = clear || fresh;
    }
    c->acc_w = w; c->acc_h = h;
    if (clear) HIPCHK(c, hipMemsetAsync(c->accum_ptr(), 0, need, c->stream));
    return RTX_OK;
}

int rtx_clear_accum(rtx_ctx* c, uint32_t w, uint32_t h) {
    BIND(c);
    if (!w || !h) return RTX_ERR_INVALID;
    int r = ensure_accum(c, w, h, true);
    if (r) return r;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

// the ONE rule for the shard tiling, shared by every entry point that takes rtx_params (render, pack / unpack, rtx_shard_slab_bytes):
// tile_size a power of two in [16, 1024] (0 => 64), shard_rank < shard_count, the local slot count fits 31 bits.  All in 64-bit arithmetic.
// RTX_FLAG_BLOCK_TILES: the ranks form a gx x gy grid of tile rectangles, gx gy = shard_count with the smallest rectangle perimeter; on a TIE the
// first factorisation in ascending gx wins, i.e. the grid with FEWER columns (taller).  royaltracer-dx_amd/sharding.py block_grid mirrors this loop line for line — pack / unpack and the slab
// sizes of all ranks depend on both sides agreeing, so change them together (tests/test_multigpu_gloo.py::test_block_grid_tie_goes_to_the_grid_with_fewer_columns pins the choice).
static void block_grid(uint64_t TX, uint64_t TY, uint32_t N, uint32_t& gx, uint32_t& gy) {
    double best = 1e300; gx = N; gy = 1;
    for (uint32_t a = 1; a <= N; a++) {
        if (N % a) continue;
        const uint32_t b = N / a;
        const double cost = (double)((TX + a - 1) / a) + (double)((TY + b - 1) / b);
        if (cost < best) { best = cost; gx = a; gy = b; }
    }
}
static const char* validate_tiling(const rtx_params* p, uint32_t& ts, uint32_t& cnt, uint64_t& npl, uint32_t* gx_out = nullptr, uint32_t* gy_out = nullptr) {
    if (!p || !p->width || !p->height) return "params: width/height must be non-zero";
    ts = p->tile_size ? p->tile_size : 64;
    if (ts < 16 || ts > 1024 || (ts & (ts - 1))) return "params: tile_size must be a power of two in [16, 1024] (0 = 64)";
    cnt = p->shard_count ? p->shard_count : 1;
    if (p->shard_rank >= cnt) return "params: shard_rank >= shard_count";
    const uint64_t TX = (p->width + (uint64_t)ts - 1) / ts, TY = (p->height + (uint64_t)ts - 1) / ts;
    uint64_t per = (TX * TY + cnt - 1) / cnt;
    uint32_t gx = 0, gy = 0;
    if ((p->flags & RTX_FLAG_BLOCK_TILES) && cnt > 1) { block_grid(TX, TY, cnt, gx, gy); per = ((TX + gx - 1) / gx) * ((TY + gy - 1) / gy); }
    if (gx_out) *gx_out = gx;
    if (gy_out) *gy_out = gy;
    if (per > 0x7FFFFFFFull / ((uint64_t)ts * ts)) return "params: image too large";      // (checked before the multiplication: 2^56 tiles of 16 x 16 would wrap)
    npl = per * ts * ts;
    return nullptr;
}

static int make_frame(rtx_ctx* c, const rtx_params* p, DevFrame& f) {
    uint32_t ts = 0, cnt = 0; uint64_t npl64 = 0;
    if (const char* e = validate_tiling(p, ts, cnt, npl64, &f.blk_gx, &f.blk_gy)) { c->err = e; return RTX_ERR_INVALID; }
    f.width = p->width; f.height = p->height; f.tile_size = ts;
    f.tile_shift = 0; while ((1u << f.tile_shift) < ts) f.tile_shift++;
    f.nblocks = 1; f.qcap = 0; f.chunks_per_sample = 0; f.taper_levels = 0; f.interleave = 0;
    f.tiles_x = (p->width + ts - 1) / ts; f.tiles_y = (p->height + ts - 1) / ts;
    f.shard_rank = p->shard_rank; f.shard_count = cnt;
    f.npl = (uint32_t)npl64;
    f.chunks_per_sample = f.npl / 256;          // tile_size >= 16 makes npl a multiple of 256
    f.batch_spp = 1; f.sample_first = p->sample_base;
    f.max_bounces = p->max_bounces; f.nee_samples = p->nee_samples; f.rr_start = p->rr_start;
    f.frame_seed = p->frame_seed; f.flags = p->flags;
    f.hist_x0 = f.hist_y0 = 0; f.hist_x1 = p->width; f.hist_y1 = p->height; f.hist_stale = nullptr;
    return RTX_OK;
}
// pixel rectangle [x0, x1) x [y0, y1) of rank r in the RTX_FLAG_BLOCK_TILES deal (shard_tile's rule, clipped to the image)
static void block_rect(uint32_t W, uint32_t H, uint32_t ts, uint32_t TX, uint32_t TY, uint32_t gx, uint32_t gy, uint32_t r, uint32_t out[4]) {
    const uint32_t bx = r % gx, by = r / gx;
    out[0] = std::min(W, (bx * TX / gx) * ts); out[2] = std::min(W, ((bx + 1u) * TX / gx) * ts);
    out[1] = std::min(H, (by * TY / gy) * ts); out[3] = std::min(H, ((by + 1u) * TY / gy) * ts);
}

static hipEvent_t take_event(rtx_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; c->ev_pool.push_back(e); }
    return c->ev_pool[c->ev_used++];
}
struct Timed {
    rtx_ctx* c; int cls; hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    Timed(rtx_ctx* c_, int cls_, hipStream_t s_ = nullptr) : c(c_), cls(cls_), s(s_ ? s_ : c_->stream) { c->stats.kernel_launches[cls]++; if (c->timing) { a = take_event(c); b = take_event(c); if (a) (void)hipEventRecord(a, s); } }
    ~Timed() { if (c->timing && a && b) { (void)hipEventRecord(b, s); c->timed.push_back({cls, a, b}); } }
};

int rtx_render(rtx_ctx* c, const rtx_params* p) {
    BIND(c);
    if (!c->committed) { c->err = "render: scene not committed"; return RTX_ERR_STATE; }
    if (!c->camera_set) { c->err = "render: camera not set"; return RTX_ERR_STATE; }
    DevFrame f;
    int r = make_frame(c, p, f);
    if (r) return r;
    if (p->max_bounces == 0 || p->max_bounces > 64) { c->err = "params: max_bounces must be in [1, 64]"; return RTX_ERR_INVALID; }
    if (p->nee_samples > 16) { c->err = "params: nee_samples must be <= 16"; return RTX_ERR_INVALID; }
    if ((r = ensure_accum(c, p->width, p->height, false))) return r;
    memset(c->stats.kernel_ms, 0, sizeof(c->stats.kernel_ms));
    memset(c->stats.kernel_launches, 0, sizeof(c->stats.kernel_launches));
    memset(c->stats.kernel_items, 0, sizeof(c->stats.kernel_items));
    c->stats.rays_primary = c->stats.rays_extension = c->stats.rays_shadow = c->stats.paths = c->stats.primary_hits = 0; c->stats.render_ms = 0;
    if (p->spp == 0) return RTX_OK;

    const uint32_t nee = c->dsc.nlights ? p->nee_samples : 0;
    uint32_t bspp = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(p->spp, c->paths_per_batch / f.npl));
    // work distribution: G workgroups, each with a private sub-queue (no global atomics in the loop)
    // sub-queues per CU: shorter tails with more, but more per-workgroup overhead; the fused tiny-scene kernels (5 workgroups resident per
    // CU, longest-first dispatch, all bounces >= 1 in one launch) measured 18.39 / 18.15 / 18.13 / 18.30 / 18.36 ms at 24 / 32 / 40 / 48 / 64
    // (a 1/4 shard: 5.22 / 4.96 / 4.99 / 4.92 / 4.96 ms; 30 is an outlier, its sub-queues alias with the 8100 image regions); the
    // general path: 16 was best in round 1 (49.3 / 42.2 ms vs 53.0 / 44.2 at 8) with 4-6 workgroups resident per CU; with the 7-8 of the end of round 2
    // (LDS trimmed) it is 32: C3 39.8 vs 40.7 ms, C5 37.8 vs 38.2 (48: 40.4 / 39.0)
    const bool fused_bvh = !c->dsc.nsmall && c->fused_bvh && c->trace_sched >= 5u;      // (the other wave schedules are experiment knobs of the separate kernels)
    // (round 3) ... and never so many that a sub-queue starts with fewer than ~8 (tiny scenes: ~16) chunks of 256 paths while there are 8 workgroups per CU: a 1/8 shard of the
    // 1080p frame (16 200 chunks at 16 spp) ran 7.29 / 6.71 / 6.30 / 6.09 / 6.08 ms with 32 / 24 / 16 / 12 / 8 sub-queues per CU (C3), Cornell at 64 spp 2.54 / 2.41 ms with 40 / 16
    // (tools/shard_kernels.py): short sub-queues leave the persistent waves of a workgroup half empty, and every round of workgroups costs one ray latency
    const bool tiny_fused = c->dsc.nsmall && c->fused;
    uint32_t nchunks = 0, G = 1, taper_levels = 0, qchunks = 1, max_blocks = 1;
    auto plan = [&](uint32_t spp_batch) {
        nchunks = f.chunks_per_sample * spp_batch;
        const uint32_t bpc_hi = tiny_fused ? 40u : 32u, per_wg = tiny_fused ? 16u : 8u;
        const uint32_t bpc = c->blocks_per_cu ? c->blocks_per_cu
                                              : std::max<uint32_t>(8u, std::min<uint32_t>(bpc_hi, (nchunks / per_wg + (uint32_t)c->num_cus - 1u) / (uint32_t)c->num_cus));
        max_blocks = (uint32_t)c->num_cus * bpc;
        G = std::max<uint32_t>(1, std::min<uint32_t>(nchunks, max_blocks));
        // tapered sub-queue sizes (taper_row_width, rtx_kernels.hpp); needs a few chunks in the shortest sub-queue to mean anything.  The fused tiny-scene kernels take their
        // sub-queues longest first anyway (k_order_queues): tapered, that order has something to work with — headline frame 17.18 -> 17.00 ms (same context, option switched)
        taper_levels = 0; qchunks = (nchunks + G - 1) / G;
        if (c->taper && !fused_bvh && G >= 64u && nchunks >= 4u * G) {
            taper_levels = c->taper_levels;
            qchunks = 0;                                                                   // sub-queue 0 takes part in every row
            for (uint32_t k = 0, row0 = 0; row0 < nchunks; k++) { row0 += taper_row_width(k, G, taper_levels); qchunks++; }
        }
    };
    plan(bspp);
    // MEMORY of the taper: every per-queue-position buffer has the uniform stride G * qcap with qcap = the LONGEST sub-queue, so the weights 8 | 4 | 2 | 1 cost
    // 8 / 5.375 = 1.49 x the entries of the even deal (both compact path-state sets, hit records, both queues, nee x 3 shadow streams: ~ +10 GB at the 128 Mi-path cap).
    // Where the path state lives by queue position (the general path: 176 B + 48 B per NEE slot and entry) RTX_OPT_PATHS_PER_BATCH therefore caps the ENTRIES: the batch shrinks
    // until the tapered layout fits it, so a frame that fitted before the taper fits now.  The fused tiny-scene kernels keep their state by path id and only
    // the two 4-byte queues grow (1.06 -> 1.58 GB at the cap): their batch stays whole — the headline frame is one batch of 132.7 M paths.
    const bool compact = c->compact_state && !(c->dsc.nsmall && c->fused) && !fused_bvh;
    while (taper_levels && compact && bspp > 1u && (uint64_t)G * qchunks * 256u > c->paths_per_batch) {
        const uint32_t next = (uint32_t)std::max<uint64_t>(1, (uint64_t)bspp * c->paths_per_batch / ((uint64_t)G * qchunks * 256u));
        bspp = next < bspp ? next : bspp - 1u;
        plan(bspp);
    }
    const uint64_t cap64 = (uint64_t)f.npl * bspp;
    if (cap64 > 0x7FFFFFFFull) { c->err = "render: batch too large"; return RTX_ERR_INVALID; }
    const uint32_t cap = (uint32_t)cap64;
    HIPCHK(c, c->d_rad.ensure((size_t)cap * 16));
    const uint32_t qcap = qchunks * 256u;
    f.nblocks = G; f.qcap = qcap; f.taper_levels = taper_levels;
    const size_t qtot = (size_t)G * qcap;
    // path state: by path id (cap entries), or — separate kernels of the default configuration — by queue position (qtot >= cap entries) in two sets
    const size_t nstate = compact ? qtot : (size_t)cap;
    if (nstate > 0xFFFFFFFFull) { c->err = "render: batch too large"; return RTX_ERR_INVALID; }
    HIPCHK(c, c->d_ray_o.ensure(nstate * 16)); HIPCHK(c, c->d_ray_d.ensure(nstate * 16)); HIPCHK(c, c->d_thr.ensure(nstate * 16)); HIPCHK(c, c->d_hit.ensure(nstate * 16));
    if (compact) { HIPCHK(c, c->d_alt_o.ensure(nstate * 16)); HIPCHK(c, c->d_alt_d.ensure(nstate * 16)); HIPCHK(c, c->d_alt_thr.ensure(nstate * 16)); }
    HIPCHK(c, c->d_queue[0].ensure(qtot * 4)); HIPCHK(c, c->d_queue[1].ensure(qtot * 4));
    HIPCHK(c, c->d_order.ensure((size_t)G * 4));
    if (fused_bvh) HIPCHK(c, c->d_hitq.ensure(qtot * 4));
    const bool stealing = !c->dsc.nsmall && !fused_bvh && c->work_stealing && p->max_bounces > 0;
    const size_t hstride = (size_t)G + (G + 31) / 32;                      // per trace launch: G fetch cursors + the exhausted bitmap
    const size_t nheads = stealing ? (size_t)p->max_bounces * (1 + std::max<uint32_t>(nee, 1)) * hstride : 0;
    if (stealing) HIPCHK(c, c->d_heads.ensure(nheads * 4));
    HIPCHK(c, c->d_pmask.ensure(((size_t)f.npl / 64 + 1) * 8));
    const uint32_t nee1 = std::max<uint32_t>(nee, 1);
    const size_t shn = qtot * nee1;
    HIPCHK(c, c->d_sh_o.ensure(shn * 16)); HIPCHK(c, c->d_sh_d.ensure(shn * 16)); HIPCHK(c, c->d_sh_c.ensure(shn * 16));
    DevPaths P;
    P.ray_o = (F4*)c->d_ray_o.p; P.ray_d = (F4*)c->d_ray_d.p; P.thr = (F4*)c->d_thr.p; P.rad = (F4*)c->d_rad.p; P.hit = (F4*)c->d_hit.p;
    P.hitmask = nullptr;
    P.out_o = P.out_d = P.out_thr = nullptr; P.oct_out = nullptr; P.oct_in = nullptr; P.perm = nullptr; P.key_mode = (uint32_t)c->octant_sort;
    const bool osort = c->octant_sort && compact && !stealing;
    if (osort) { HIPCHK(c, c->d_oct[0].ensure(qtot)); HIPCHK(c, c->d_oct[1].ensure(qtot)); HIPCHK(c, c->d_perm.ensure(qtot * 4)); }
    if (osort && c->octant_sort == 2) { HIPCHK(c, hipMemsetAsync(c->d_oct[0].p, 0, qtot, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_oct[1].p, 0, qtot, c->stream)); }
    if (c->dsc.nsmall && c->fused) { HIPCHK(c, c->d_hitmask.ensure(((size_t)cap / 64 + 1) * 8)); P.hitmask = (unsigned long long*)c->d_hitmask.p; }
    P.sh_o = (F4*)c->d_sh_o.p; P.sh_d = (F4*)c->d_sh_d.p; P.sh_c = (F4*)c->d_sh_c.p;
    uint32_t* queue[2] = {(uint32_t*)c->d_queue[0].p, (uint32_t*)c->d_queue[1].p};

    // per-workgroup counters of one batch: Q[b][G] queue lengths entering bounce b (b = 0..mb),
    // S[b][j][G] shadow queue lengths; every workgroup stores its own entry, so nothing needs zeroing per batch
    const uint32_t mb = p->max_bounces;
    const size_t ncnt = ((size_t)(mb + 1) + (size_t)mb * nee1 + 1) * G;    // + one row: paths generated (fused raygen+trace)
    const uint32_t nbatches = (p->spp + bspp - 1) / bspp;
    HIPCHK(c, c->d_counters.ensure(ncnt * 4));
    if (c->h_counters_words < ncnt * nbatches) {
        if (c->h_counters) (void)hipHostFree(c->h_counters);
        c->h_counters = nullptr; c->h_counters_words = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->h_counters, ncnt * nbatches * 4, hipHostMallocDefault));
        c->h_counters_words = ncnt * nbatches;
    }
    uint32_t* cnt = (uint32_t*)c->d_counters.p;
    const CameraGPU* cam = (const CameraGPU*)c->d_cam.p;
    c->ev_used = 0; c->timed.clear();
    hipStream_t st = c->stream;
    auto Q = [&](uint32_t b) { return cnt + (size_t)b * G; };
    auto S = [&](uint32_t b, uint32_t j) { return cnt + ((size_t)(mb + 1) + (size_t)b * nee1 + j) * G; };

    // every early return below (HIPCHK) must not leave shadow-ray launches of the internal `aux` stream running behind the caller's back: they read the
    // shadow entries and update `rad`, which the next call re-uses
    struct AuxJoin { rtx_ctx* c; bool armed = true; ~AuxJoin() { if (armed && c->aux) (void)hipStreamSynchronize(c->aux); } } aux_join{c};
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    HIPCHK(c, hipMemsetAsync(cnt, 0, ncnt * 4, st));
    if (c->dsc.nsmall && c->fused) launch_packet_masks(st, c->dsc, f, cam, (unsigned long long*)c->d_pmask.p);   // per 8x8 block, shared by all samples
    for (uint32_t bi = 0; bi < nbatches; bi++) {
        DevFrame fb = f;
        fb.sample_first = p->sample_base + bi * bspp;
        fb.batch_spp = std::min(bspp, p->spp - bi * bspp);
        fb.interleave = 0;                                     // (k_raygen only: the tiny-scene raygen keeps its packet order)
        if (c->sample_interleave) while (fb.interleave < 4u && !((fb.batch_spp >> fb.interleave) & 1u)) fb.interleave++;       // S = the largest power of two <= 16 dividing the batch's sample count
        const bool fused = c->dsc.nsmall && c->fused;
        uint32_t* gen_row = cnt + ((size_t)(mb + 1) + (size_t)mb * nee1) * G;
        if (fused) { Timed t(c, RTX_K_RAYGEN); launch_raygen_trace_small(st, c->dsc, fb, P, cam, queue[0], Q(0), gen_row, (const unsigned long long*)c->d_pmask.p); }
        else { Timed t(c, RTX_K_RAYGEN); launch_raygen(st, fb, P, cam, queue[0], Q(0), compact); }
        // dispatch order of the fused bounce kernels: longest sub-queue first, from the lengths after the primary rays (the later
        // bounces keep the ranking: survivors are a near-constant fraction)
        const uint32_t* order = nullptr;
        if (fused && c->lpt_order && G > 1) { launch_order_queues(st, Q(0), G, (uint32_t*)c->d_order.p); order = (const uint32_t*)c->d_order.p; }
        if (fused) {          // tiny scene: trace + shade + shadow fused; bounce 0 (traced by raygen) and then ALL later bounces in one launch each
            { Timed t(c, RTX_K_BOUNCE); launch_bounce_small(st, c->dsc, fb, P, 0, 1, queue[0], queue[1], Q(0), S(0, 0), order); }
            if (mb > 1) { Timed t(c, RTX_K_BOUNCE); launch_bounce_small(st, c->dsc, fb, P, 1, mb, queue[0], queue[1], Q(0), S(0, 0), order, c->bounce_ring); }
        }
        if (fused_bvh) {      // general scenes: all bounces of every sub-queue in one launch (k_bounce_bvh), sub-queues longest first
            const uint32_t* ord = nullptr;
            if (c->lpt_order && G > 1) { launch_order_queues(st, Q(0), G, (uint32_t*)c->d_order.p); ord = (const uint32_t*)c->d_order.p; }
            Timed t(c, RTX_K_BOUNCE);
            launch_bounce_bvh(st, c->dsc, fb, P, 0, mb, queue[0], queue[1], (uint32_t*)c->d_hitq.p, Q(0), S(0, 0), ord);
        }
        if (stealing) HIPCHK(c, hipMemsetAsync(c->d_heads.p, 0, nheads * 4, st));          // one cursor block per trace launch of the batch
        // General path: the shadow rays of bounce b and the closest-hit rays of bounce b + 1 both depend on shade(b) only, so k_trace_shadow(b) runs
        // on a second stream beside k_trace_closest(b + 1); shade(b + 1) waits for it (it overwrites the shadow entries, and both touch rad).  The two
        // persistent kernels fill each other's tails and memory stalls: C3 40.8 -> 40.3 ms, C5 38.3 -> 36.9 ms per frame, images unchanged.  Not while
        // kernels are timed (RTX_OPT_KERNEL_TIMING): overlapping launches have no per-kernel time.
        const bool ovl = c->overlap_shadow && !c->timing && !fused && !fused_bvh;
        if (ovl && !c->aux) c->aux = c->streams.s[1];
        hipEvent_t ev_shadow_done = nullptr;
        auto Hd = [&](uint32_t b, uint32_t k) { return stealing ? (uint32_t*)c->d_heads.p + ((size_t)b * (1 + nee1) + k) * hstride : nullptr; };
        // sub-queues per workgroup of a traversal launch predicted to hold `rays` rays: double while a workgroup would start with fewer than merge_rays and at least one full
        // round of resident workgroups (8 per CU) remains.  Measured in ONE context, option switched between rounds (tools/ab_frame.py same=1): C3 39.45 -> 38.67 / 38.57 /
        // 38.51 ms per frame at 512 / 1024 / 2048, C5 36.77 -> 35.90 / 35.78 / 36.02; going below one round (4 workgroups per CU) changes nothing
        auto merge_for = [&](uint64_t pred_num, bool have) -> uint32_t {
            if (!have || !c->merge_rays || !c->pred_paths || stealing) return 1u;
            const double rays = (double)pred_num * (double)((uint64_t)fb.npl * fb.batch_spp) / (double)c->pred_paths;
            uint32_t k = 1u;
            while (k < kMaxMerge && G / (2u * k) >= (uint32_t)c->num_cus * 8u && rays * k / G < (double)c->merge_rays) k *= 2u;
            return k;
        };
        for (uint32_t b = 0; b < mb && !fused && !fused_bvh; b++) {
            uint32_t* q = queue[b & 1]; uint32_t* qn = queue[(b + 1) & 1];
            DevPaths Pb = P;                                  // compact state: bounce b reads set (b & 1) and writes the survivors into the other one
            if (compact) {
                F4* set[2][3] = {{(F4*)c->d_ray_o.p, (F4*)c->d_ray_d.p, (F4*)c->d_thr.p}, {(F4*)c->d_alt_o.p, (F4*)c->d_alt_d.p, (F4*)c->d_alt_thr.p}};
                Pb.ray_o = set[b & 1][0]; Pb.ray_d = set[b & 1][1]; Pb.thr = set[b & 1][2];
                Pb.out_o = set[(b + 1) & 1][0]; Pb.out_d = set[(b + 1) & 1][1]; Pb.out_thr = set[(b + 1) & 1][2];
                if (osort) {           // shade(b) notes the survivors' octants for bounce b + 1; trace(b), b >= 1, sorts by what shade(b - 1) noted (camera rays of one block share their octant anyway)
                    Pb.oct_out = c->octant_sort == 2 ? nullptr : (uint8_t*)c->d_oct[(b + 1) & 1].p;
                    if (b >= (c->octant_sort == 3 ? 2u : 1u)) { Pb.oct_in = (const uint8_t*)c->d_oct[b & 1].p; Pb.perm = (uint32_t*)c->d_perm.p; }     // (origin cells: bounce 1 starts at the camera rays' hits, already in image order)
                }
            }
            DevScene scb = c->dsc;
            if (c->wide_nodes && b >= 1u) { scb.nodes_f = (const F4*)c->d_nodes_wide.p; scb.node_v4 = 8u; }
            if (c->lds_nodes_closest) scb.lds_nodes = c->lds_nodes_closest;          // closest-hit launches stage more of the tree's top than the shadow launches (finalise_scene)
            { Timed t(c, RTX_K_TRACE); launch_trace_closest(st, fb, scb, Pb, b, q, Q(b), Hd(b, 0), merge_for(b < c->pred_q.size() ? c->pred_q[b] : 0, b < c->pred_q.size())); }
            if (ovl && ev_shadow_done) HIPCHK(c, hipStreamWaitEvent(st, ev_shadow_done, 0));      // shade(b) overwrites the shadow entries and touches rad: after shadow(b - 1)
            { Timed t(c, RTX_K_SHADE); launch_shade(st, c->dsc, fb, Pb, b, q, Q(b), qn, Q(b + 1), S(b, 0)); }
            hipStream_t ss = st;
            if (ovl && nee) {
                hipEvent_t e = take_event(c); ev_shadow_done = take_event(c);
                if (!e || !ev_shadow_done) { c->err = "render: out of events"; return RTX_ERR_HIP; }
                HIPCHK(c, hipEventRecord(e, st)); HIPCHK(c, hipStreamWaitEvent(c->aux, e, 0));
                ss = c->aux;
            }
            for (uint32_t j = 0; j < nee; j++) { const size_t ps = (size_t)b * nee1 + j; Timed t(c, RTX_K_SHADOW, ss); launch_trace_shadow(ss, fb, c->dsc, Pb, j, S(b, j), Hd(b, 1 + j), merge_for(ps < c->pred_s.size() ? c->pred_s[ps] : 0, ps < c->pred_s.size() && c->pred_nee1 == nee1)); }
            if (ovl && nee) HIPCHK(c, hipEventRecord(ev_shadow_done, c->aux));
        }
        if (ovl && ev_shadow_done) { HIPCHK(c, hipStreamWaitEvent(st, ev_shadow_done, 0)); ev_shadow_done = nullptr; }
        { Timed t(c, RTX_K_ACCUM); launch_accumulate(st, max_blocks, fb, P, c->accum_ptr()); }
        HIPCHK(c, hipMemcpyAsync(c->h_counters + (size_t)bi * ncnt, cnt, ncnt * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipGetLastError());
    aux_join.armed = false;                    // the main stream waited for the last shadow launch (ev_shadow_done) before the accumulation: stream order covers the aux stream
    c->pending.active = true; c->pending.ncnt = ncnt; c->pending.nbatches = nbatches; c->pending.G = G; c->pending.mb = mb; c->pending.nee = nee; c->pending.nee1 = nee1;
    c->pending.fused = c->dsc.nsmall && c->fused; c->pending.fused_bvh = fused_bvh;
    // RTX_OPT_ASYNC on a caller-bound stream: the frame is enqueued, the caller goes on enqueueing its epilogue (rtx_pack_tiles -> collective -> rtx_unpack_tiles) behind it
    // with no host join in between; statistics and the launch-size predictions of the next frame are collected at the next call that needs them (BIND)
    if (c->async && !c->own_stream) return RTX_OK;
    return finish_render(c);
}

static int finish_render(rtx_ctx* c) {
    if (!c->pending.active) return RTX_OK;
    c->pending.active = false;
    const size_t ncnt = c->pending.ncnt; const uint32_t nbatches = c->pending.nbatches, G = c->pending.G, mb = c->pending.mb, nee = c->pending.nee, nee1 = c->pending.nee1;
    const bool fused = c->pending.fused, fused_bvh = c->pending.fused_bvh;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->stats.render_ms = ms;
    for (const TimedLaunch& t : c->timed) { float m = 0.0f; if (hipEventElapsedTime(&m, t.a, t.b) == hipSuccess) c->stats.kernel_ms[t.cls] += m; }
    if (!fused && !fused_bvh) { c->pred_paths = 0; c->pred_q.assign(mb, 0); c->pred_s.assign((size_t)mb * nee1, 0); c->pred_nee1 = nee1; }
    for (uint32_t bi = 0; bi < nbatches; bi++) {
        const uint32_t* h = c->h_counters + (size_t)bi * ncnt;
        auto sumG = [&](size_t row) { uint64_t s = 0; for (uint32_t g = 0; g < G; g++) s += h[row * G + g]; return s; };
        if (!fused && !fused_bvh) {                 // what the next call's launches are sized by (merge_for)
            c->pred_paths += sumG(0);
            for (uint32_t b = 0; b < mb; b++) { c->pred_q[b] += sumG(b); for (uint32_t j = 0; j < nee1; j++) c->pred_s[(size_t)b * nee1 + j] += sumG((size_t)(mb + 1) + (size_t)b * nee1 + j); }
        }
        const uint64_t prim = fused ? sumG((size_t)(mb + 1) + (size_t)mb * nee1) : sumG(0);
        c->stats.rays_primary += prim; c->stats.paths += prim;
        c->stats.primary_hits += fused ? sumG(0) : 0;
        for (uint32_t b = 1; b < mb; b++) c->stats.rays_extension += sumG(b);
        for (uint32_t b = 0; b < mb; b++) for (uint32_t j = 0; j < nee; j++) c->stats.rays_shadow += sumG((size_t)(mb + 1) + (size_t)b * nee1 + j);
    }
    c->stats.kernel_items[RTX_K_RAYGEN] = c->stats.paths;
    c->stats.kernel_items[RTX_K_BOUNCE] = fused ? c->stats.primary_hits + c->stats.rays_extension : fused_bvh ? c->stats.rays_primary + c->stats.rays_extension : 0;   // tiny scenes: bounce 0 shades the primary hits only
    c->stats.kernel_items[RTX_K_TRACE] = (fused || fused_bvh) ? 0 : c->stats.rays_primary + c->stats.rays_extension;
    c->stats.kernel_items[RTX_K_SHADE] = (fused || fused_bvh) ? 0 : c->stats.rays_primary + c->stats.rays_extension;
    c->stats.kernel_items[RTX_K_SHADOW] = (fused || fused_bvh) ? 0 : c->stats.rays_shadow;
    c->stats.kernel_items[RTX_K_ACCUM] = c->stats.paths;
    return RTX_OK;
}

size_t rtx_pass1_slots(uint32_t w, uint32_t h) { return (size_t)((w + 3) / 4) * ((h + 3) / 4) * 16; }
static int p1_alloc(rtx_ctx* c, size_t slots);

// ---- the ReSTIR passes as wavefront stages (csrc/rtx_restir_wave.hpp) ----------------------------------------------------------------------------------
// One pass at a time owns the work area.  `nitems` work items (pixels of the shard's own tiles, or — passes 1 and 2 on shards — of the dilated tiles) are cut
// into 256-item chunks and dealt round-robin to G workgroups, each with a private sub-queue: stage kernels and the persistent traversal kernels of a pass all run
// with G workgroups, workgroup b owning sub-queue b.  G: `restir_chunks` chunks per workgroup (more = fuller persistent waves, fewer = shorter launch tails).
struct RsPlan { RsQ q; uint32_t* cnt; uint32_t G; DevFrame fq; DevPaths P[2]; };
static int rs_plan(rtx_ctx* c, const DevFrame& f, uint32_t nitems, const uint32_t* pixels, uint32_t rows, RsPlan& R, uint32_t lane) {
    rtx_ctx::RsArea& A = c->rs_area[lane];
    const uint32_t nchunks = std::max<uint32_t>(1u, (nitems + 255u) / 256u);
    // `restir_chunks` chunks per workgroup at full frame size, but never fewer than ~8 workgroups per CU while there are that many chunks: a 1/8 shard (1 180 chunks with its
    // halo) ran 3.63 ms per frame with 295 workgroups of 4 chunks and 2.22 ms with 1 180 of one (tools/shard_time.py sponza restir 8 blocks=1 tile=32)
    const uint32_t want = std::max<uint32_t>((nchunks + c->restir_chunks - 1) / c->restir_chunks, (uint32_t)c->num_cus * 8u);
    const uint32_t G = (std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(want, nchunks), (uint32_t)c->num_cus * 64u)) + 7u) & ~7u;   // a multiple of 8: rs_wg() maps workgroups to XCD-contiguous ranges
    const uint32_t qcap = ((nchunks + G - 1) / G) * 256u, rcap = qcap * 9u;              // a pixel casts at most 9 visibility rays in one stage (pass 3, select)
    const size_t qtot = (size_t)G * qcap, rtot = (size_t)G * rcap;
    // 32-bit indices everywhere: queue positions (rtot), the per-item candidate / cold records, and the ray payload `item * kRsOcc + k` that addresses the occlusion bytes
    // (rs_push_ray: a wrapped payload would write the verdict of ANOTHER pixel's ray, silently)
    if (rtot > 0xFFFFFFFFull || (uint64_t)nitems * kRsOcc > 0xFFFFFFFFull || (uint64_t)nitems * kRsCand > 0xFFFFFFFFull || (uint64_t)nitems * 5u > 0xFFFFFFFFull) {
        c->err = "render_restir: image too large (32-bit ray payloads and record indices)"; return RTX_ERR_INVALID;
    }
    HIPCHK(c, A.state.ensure(qtot * 16 * 2 * kRsStreams)); HIPCHK(c, A.hit.ensure(qtot * 16));
    HIPCHK(c, A.cls.ensure((size_t)nitems * 4)); HIPCHK(c, A.fin.ensure((size_t)nitems * 16)); HIPCHK(c, A.cold.ensure((size_t)nitems * 16 * 5));
    HIPCHK(c, A.occ.ensure((size_t)nitems * kRsOcc)); HIPCHK(c, A.cand.ensure((size_t)nitems * 4 * kRsCand));
    HIPCHK(c, A.sho.ensure(rtot * 16)); HIPCHK(c, A.shd.ensure(rtot * 16)); HIPCHK(c, A.pay.ensure(rtot * 4));
    HIPCHK(c, A.cnt.ensure((size_t)rows * G * 4));
    RsQ& q = R.q;
    q.nitems = nitems; q.pixels = pixels; q.G = G; q.qcap = qcap; q.rcap = rcap;
    for (uint32_t set = 0; set < 2; set++) for (uint32_t k = 0; k < kRsStreams; k++) q.st[set][k] = (F4*)A.state.p + ((size_t)set * kRsStreams + k) * qtot;
    q.hit = (F4*)A.hit.p; q.cls = (uint32_t*)A.cls.p; q.fin = (F4*)A.fin.p; q.cold = (F4*)A.cold.p;
    q.occ = (uint8_t*)A.occ.p; q.cand = (uint32_t*)A.cand.p;
    q.sh_o = (F4*)A.sho.p; q.sh_d = (F4*)A.shd.p; q.sh_pay = (uint32_t*)A.pay.p;
    q.rays = (unsigned long long*)c->d_p1cnt.p;
    R.cnt = (uint32_t*)A.cnt.p; R.G = G;
    R.fq = f; R.fq.nblocks = G; R.fq.qcap = qcap;
    for (uint32_t set = 0; set < 2; set++) {          // what k_trace_closest sees of a set: rays and hit records by queue position ("compact state": out_o != nullptr is the flag)
        DevPaths P{}; P.ray_o = q.st[set][0]; P.ray_d = q.st[set][1]; P.hit = q.hit; P.out_o = q.st[set ^ 1u][0];
        R.P[set] = P;
    }
    return RTX_OK;
}
// pass 1 of one sample (RayGen_v6_pass1.hlsl:48-190): raygen | trace | ris | trace | ris_finish | trace | first | (trace | loop) x bounces | emit_final | trace | finish.
// bufs != nullptr (a ReSTIR frame): pass 2 (RayGen_v6_pass2.hlsl:46-204) rides on the last two stages.  Everything is enqueued on `st` with the work area of `lane`.
static int rs_pass1(rtx_ctx* c, const DevFrame& f, uint32_t sample_id, F4* accum, const uint32_t* pixels, uint32_t npixels, uint32_t* const* bufs, uint32_t lane, hipStream_t st) {
    const uint32_t mb = f.max_bounces, rows = 4u + mb + 1u;
    RsPlan R; int r = rs_plan(c, f, pixels ? npixels : f.npl, pixels, rows, R, lane); if (r) return r;
    const DevScene& sc = c->dsc; const RsQ& q = R.q;
    const CameraGPU* cam = (const CameraGPU*)c->d_cam.p;
    auto row = [&](uint32_t k) { return R.cnt + (size_t)k * R.G; };
    uint32_t* res_di = (uint32_t*)c->d_res_di.p; uint32_t* res_gi = (uint32_t*)c->d_res_gi.p; uint32_t* sdata = (uint32_t*)c->d_sdata.p;
    uint32_t* shrow = row(4 + mb);                          // lengths of the ray sub-queues: DI visibility (stage 2) + reconnection + temporal rays (stage 5)
    { Timed t(c, RTX_K_RAYGEN, st); launch_rs_raygen(st, R.fq, q, cam, sample_id, row(0)); }
    { Timed t(c, RTX_K_TRACE, st); launch_trace_closest(st, R.fq, sc, R.P[0], 0, nullptr, row(0), nullptr); }                    // camera rays (tmin 1e-4)
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_ris(st, sc, R.fq, q, row(0), row(1), accum, res_di, res_gi, sdata); }
    { Timed t(c, RTX_K_TRACE, st); launch_trace_closest(st, R.fq, sc, R.P[1], 1, nullptr, row(1), nullptr); }                    // the BSDF candidates of SampleRIS
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_ris_finish(st, sc, R.fq, q, row(1), row(2), shrow, res_di, sdata); }
    { Timed t(c, RTX_K_TRACE, st); launch_trace_closest(st, R.fq, sc, R.P[0], 1, nullptr, row(2), nullptr); }                    // first path vertex
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_first(st, sc, R.fq, q, row(2), row(3)); }
    for (uint32_t i = 0; i < mb; i++) {
        const uint32_t set = (i + 1u) & 1u;                 // k_rs_p1_first wrote set 1
        { Timed t(c, RTX_K_TRACE, st); launch_trace_closest(st, R.fq, sc, R.P[set], 1, nullptr, row(3 + i), nullptr); }
        { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_loop(st, sc, R.fq, q, set, i, row(3 + i), row(4 + i)); }
    }
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_emit_final(st, sc, R.fq, q, cam, bufs, shrow); }
    { Timed t(c, RTX_K_SHADOW, st); launch_trace_occ(st, sc, q, shrow); }                                                        // DI visibility, the selected reconnection, the temporal pass's two rays
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p1_finish(st, sc, R.fq, q, accum, res_di, res_gi, sdata, cam, bufs); }
    if (bufs && c->restir_keys) { Timed t(c, RTX_K_SHADE, st); launch_rs_p3_keys(st, R.fq, q, bufs, (F4*)c->d_rs_key_a.p, (F4*)c->d_rs_key_b.p); }      // what the spatial pass's neighbour tests read
    HIPCHK(c, hipGetLastError());
    return RTX_OK;
}
static int rs_pass3(rtx_ctx* c, const DevFrame& f, uint32_t* const bufs[6], F4* accum, const uint32_t* pixels, uint32_t npixels, uint32_t lane, hipStream_t st) {
    RsPlan R; int r = rs_plan(c, f, pixels ? npixels : f.npl, pixels, 2, R, lane); if (r) return r;
    const CameraGPU* cam = (const CameraGPU*)c->d_cam.p;
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p3_select(st, c->dsc, R.fq, R.q, cam, bufs, R.cnt, c->restir_keys ? (F4*)c->d_rs_key_a.p : nullptr, c->restir_keys ? (F4*)c->d_rs_key_b.p : nullptr); }
    { Timed t(c, RTX_K_SHADOW, st); launch_trace_occ(st, c->dsc, R.q, R.cnt); }
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p3_merge(st, c->dsc, R.fq, R.q, bufs, R.cnt + R.G); }
    { Timed t(c, RTX_K_SHADOW, st); launch_trace_occ(st, c->dsc, R.q, R.cnt + R.G); }
    { Timed t(c, RTX_K_SHADE, st); launch_rs_p3_shade(st, c->dsc, R.fq, R.q, bufs, accum); }
    HIPCHK(c, hipGetLastError());
    return RTX_OK;
}
// One pass over a work list as `lanes` independent parts: part 0 on the context's stream, part 1 on the internal stream, joined at the end.  A ReSTIR frame is ~20 short,
// dependent launches; run as ONE chain every launch drains before the next ramps up (k_trace_* at 4.8-5.0 of 8 waves per SIMD, VALU pipes 0.82 busy: profiles/r03_pmc_restir.md).
// Pixels are independent inside passes 1 + 2 and inside pass 3, so two chains over the two halves of the list fill each other's tails.  Not while kernels are timed.
extern "C++" {
template <class F>
static int rs_lanes(rtx_ctx* c, const uint32_t* pixels, uint32_t npixels, F&& pass) {
    const uint32_t L = (pixels && !c->timing && npixels >= c->restir_lane_min) ? c->restir_lanes : 1u;
    if (L <= 1u) return pass(pixels, npixels, 0u, c->stream);
    hipEvent_t e0 = take_event(c);
    if (!e0) { c->err = "render_restir: out of events"; return RTX_ERR_HIP; }
    HIPCHK(c, hipEventRecord(e0, c->stream));
    const uint32_t part = (((npixels + L - 1u) / L) + 255u) & ~255u;          // whole chunks
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t lo = std::min(npixels, l * part), hi = std::min(npixels, (l + 1u) * part);
        if (lo == hi) continue;
        hipStream_t st = c->stream;
        if (l) {
            if (!c->lane_stream[l - 1]) c->lane_stream[l - 1] = c->streams.s[1 + l];
            st = c->lane_stream[l - 1];
            HIPCHK(c, hipStreamWaitEvent(st, e0, 0));
        }
        int r = pass(pixels + lo, hi - lo, l, st); if (r) return r;
        if (l) { hipEvent_t e = take_event(c); if (!e) { c->err = "render_restir: out of events"; return RTX_ERR_HIP; } HIPCHK(c, hipEventRecord(e, st)); HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0)); }
    }
    return RTX_OK;
}
}  // extern "C++"
static void stats_begin(rtx_ctx* c) {
    memset(c->stats.kernel_ms, 0, sizeof(c->stats.kernel_ms)); memset(c->stats.kernel_launches, 0, sizeof(c->stats.kernel_launches)); memset(c->stats.kernel_items, 0, sizeof(c->stats.kernel_items));
    c->ev_used = 0; c->timed.clear();
}
static void stats_end_restir(rtx_ctx* c, const unsigned long long cnt[3]) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->stats.render_ms = ms;
    for (const TimedLaunch& t : c->timed) { float m = 0.0f; if (hipEventElapsedTime(&m, t.a, t.b) == hipSuccess) c->stats.kernel_ms[t.cls] += m; }
    c->stats.rays_primary = cnt[0]; c->stats.rays_extension = cnt[1]; c->stats.rays_shadow = cnt[2]; c->stats.paths = cnt[0]; c->stats.primary_hits = 0;
    c->stats.kernel_items[RTX_K_RAYGEN] = cnt[0]; c->stats.kernel_items[RTX_K_TRACE] = cnt[0] + cnt[1]; c->stats.kernel_items[RTX_K_SHADOW] = cnt[2];
}

int rtx_render_v6_pass1(rtx_ctx* c, const rtx_params* p) {
    BIND(c);
    if (!c->committed) { c->err = "render: scene not committed"; return RTX_ERR_STATE; }
    if (!c->camera_set) { c->err = "render: camera not set"; return RTX_ERR_STATE; }
    DevFrame f;
    int r = make_frame(c, p, f);
    if (r) return r;
    if (p->max_bounces > 64 || p->nee_samples > 16) { c->err = "params: max_bounces <= 64, nee_samples <= 16"; return RTX_ERR_INVALID; }
    if ((r = ensure_accum(c, p->width, p->height, false))) return r;
    const size_t slots = rtx_pass1_slots(p->width, p->height);
    if ((r = p1_alloc(c, slots))) return r;
    stats_begin(c);
    HIPCHK(c, hipMemsetAsync(c->d_p1cnt.p, 0, 24, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_begin, c->stream));
    for (uint32_t s = 0; s < p->spp; s++) {
        if (c->restir_wave) { if ((r = rs_pass1(c, f, p->sample_base + s, c->accum_ptr(), nullptr, 0, nullptr, 0u, c->stream))) return r; }
        else { Timed t(c, RTX_K_BOUNCE);
               launch_v6_pass1(c->stream, (uint32_t)c->num_cus * 8u, c->dsc, f, (const CameraGPU*)c->d_cam.p, p->sample_base + s, c->accum_ptr(),
                               (uint32_t*)c->d_res_di.p, (uint32_t*)c->d_res_gi.p, (uint32_t*)c->d_sdata.p, (unsigned long long*)c->d_p1cnt.p); }
    }
    HIPCHK(c, hipEventRecord(c->ev_end, c->stream));
    HIPCHK(c, hipGetLastError());
    unsigned long long cnt[3] = {0, 0, 0};
    TO_HOST(c, cnt, c->d_p1cnt.p, 24);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    stats_end_restir(c, cnt);
    return RTX_OK;
}

static int p1_alloc(rtx_ctx* c, size_t slots) {
    HIPCHK(c, c->d_res_di.ensure(slots * 40)); HIPCHK(c, c->d_res_gi.ensure(slots * 40)); HIPCHK(c, c->d_sdata.ensure(slots * 60));
    HIPCHK(c, c->d_p1cnt.ensure(32));          // rays by type (3 x u64) + stale history reads
    if (c->p1_slots != slots) {
        HIPCHK(c, hipMemsetAsync(c->d_res_di.p, 0, slots * 40, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_res_gi.p, 0, slots * 40, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_sdata.p, 0, slots * 60, c->stream));
        c->p1_slots = slots;
    }
    return RTX_OK;
}

int rtx_restir_reset(rtx_ctx* c) {
    BIND(c);
    c->last_slots = 0;           // the next frame starts from zeroed g_*_last buffers
    c->hist_all = true;          // ... which is what every rank holds then: valid everywhere
    return RTX_OK;
}

int rtx_render_restir(rtx_ctx* c, const rtx_params* p) {
    BIND(c);
    if (!c->committed) { c->err = "render: scene not committed"; return RTX_ERR_STATE; }
    if (!c->camera_set) { c->err = "render: camera not set"; return RTX_ERR_STATE; }
    DevFrame f;
    int r = make_frame(c, p, f);
    if (r) return r;
    if (p->max_bounces > 64 || p->nee_samples > 16) { c->err = "params: max_bounces <= 64, nee_samples <= 16"; return RTX_ERR_INVALID; }
    // ReSTIR ON SHARDS (shard_count > 1).  The spatial pass of a pixel reads this frame's pass-1 / pass-2 records of neighbours within 20 px
    // (RayGen_v6_pass3.hlsl:46-372) and the temporal pass reads last frame's history at an arbitrary reprojected pixel (RayGen_v6_pass2.hlsl:46-204).  So a shard
    //   * runs passes 1 and 2 on its tiles DILATED by 20 px (the halo is recomputed: seeds depend on the pixel only, results are what the owner computes),
    //   * runs pass 3 (and the accumulation) on its own tiles,
    //   * and after the frame the shards exchange the history of their own tiles: rtx_restir_pack_state -> one all-gather -> rtx_restir_unpack_state,
    // which the caller does between frames — hence one frame per call.  Images and histories are bit-identical to the unsharded run.
    // With RTX_FLAG_BLOCK_TILES the tiles of a shard form ONE rectangle, so the dilation adds a 20-px rim (8 shards at 1080p: 1.16 x the own pixels) instead of a rim
    // around every 64-px tile (2.6 x with the round-robin deal).
    const bool sharded = p->shard_count > 1;
    if (sharded && p->spp != 1) { c->err = "render_restir: on shards the history has to be exchanged after every frame (rtx_restir_pack_state / unpack_state): spp must be 1"; return RTX_ERR_INVALID; }
    if ((r = ensure_accum(c, p->width, p->height, false))) return r;
    const uint32_t* halo = nullptr; const uint32_t* own = nullptr; uint32_t nhalo = 0, nown = 0;
    // (tiny scenes keep the slot order when unsharded: the Cornell frame measured 3.19 ms that way and 3.69 ms through the Morton list; the BVH scenes gain ~1 %)
    if (sharded || (c->restir_wave && !c->dsc.nsmall)) {
        const uint32_t key[6] = {p->width, p->height, f.tile_size, p->shard_rank, f.shard_count, p->flags & RTX_FLAG_BLOCK_TILES};
        if (memcmp(key, c->halo_key, sizeof(key)) != 0 || !c->d_own.p) {
            const uint32_t W = p->width, H = p->height, ts = f.tile_size, R = 20u;           // spatial radius: RayGen_v6_pass3.hlsl (random pixel within 20)
            if (W > 65535u || H > 65535u) { c->err = "render_restir: images are limited to 65535 x 65535"; return RTX_ERR_INVALID; }
            std::vector<uint8_t> mask((size_t)W * H, 0);                                      // bit 0: own pixel, bit 1: own or within the halo
            for (uint32_t k = 0; k < f.npl >> (2u * f.tile_shift); k++) {                     // the shard's tiles, by the one rule of slot_to_pixel
                uint32_t tx, ty;
                if (!shard_tile(f, k, tx, ty)) continue;
                const uint32_t x0 = tx * ts > R ? tx * ts - R : 0u, y0 = ty * ts > R ? ty * ts - R : 0u;
                const uint32_t x1 = std::min(W, (tx + 1) * ts + R), y1 = std::min(H, (ty + 1) * ts + R);
                for (uint32_t y = y0; y < y1; y++) memset(&mask[(size_t)y * W + x0], 2, x1 - x0);
            }
            for (uint32_t k = 0; k < f.npl >> (2u * f.tile_shift); k++) {
                uint32_t tx, ty;
                if (!shard_tile(f, k, tx, ty)) continue;
                for (uint32_t y = ty * ts; y < std::min(H, (ty + 1) * ts); y++) memset(&mask[(size_t)y * W + tx * ts], 3, std::min(W, (tx + 1) * ts) - tx * ts);
            }
            // 8 x 8 pixel blocks (one wave each) in Morton order: a 256-pixel chunk is a 16 x 16 px square, the 256 consecutive chunks a range of workgroups on one
            // XCD takes (rs_wg) a 256 x 256 px square — the neighbour gathers of the spatial pass stay in that XCD's L2
            const uint32_t BX = (W + 7) / 8, BY = (H + 7) / 8;
            uint32_t side = 1; while (side < std::max(BX, BY)) side <<= 1;
            std::vector<uint32_t> lown, lhalo;
            auto spread = [](uint32_t v) { v &= 0xFFFFu; v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
            std::vector<std::pair<uint32_t, uint32_t>> order; order.reserve((size_t)BX * BY);
            for (uint32_t by = 0; by < BY; by++) for (uint32_t bx = 0; bx < BX; bx++) order.push_back({spread(bx) | (spread(by) << 1), bx | (by << 16)});
            std::sort(order.begin(), order.end());
            for (const auto& e : order) {
                const uint32_t bx = (e.second & 0xFFFFu) * 8u, by = (e.second >> 16) * 8u;
                for (uint32_t y = by; y < std::min(H, by + 8); y++) for (uint32_t x = bx; x < std::min(W, bx + 8); x++) {
                    const uint8_t mk = mask[(size_t)y * W + x];
                    if (mk & 1) lown.push_back(x | (y << 16));
                    if (mk & 2) lhalo.push_back(x | (y << 16));
                }
            }
            if ((r = upload(c, c->d_own, lown))) return r;
            if (sharded) { if ((r = upload(c, c->d_halo, lhalo))) return r; }
            c->own_count = (uint32_t)lown.size(); c->halo_count = sharded ? (uint32_t)lhalo.size() : 0u; memcpy(c->halo_key, key, sizeof(key));
        }
        own = (const uint32_t*)c->d_own.p; nown = c->own_count;
        if (sharded) { halo = (const uint32_t*)c->d_halo.p; nhalo = c->halo_count; } else { halo = own; nhalo = nown; }
    }
    const size_t slots = rtx_pass1_slots(p->width, p->height);
    if ((r = p1_alloc(c, slots))) return r;
    HIPCHK(c, c->d_last_di.ensure(slots * 40)); HIPCHK(c, c->d_last_gi.ensure(slots * 40)); HIPCHK(c, c->d_last_sd.ensure(slots * 60));
    if (c->restir_wave && c->restir_keys) { HIPCHK(c, c->d_rs_key_a.ensure(slots * 32)); HIPCHK(c, c->d_rs_key_b.ensure(slots * 32)); }
    if (c->last_slots != slots) {
        HIPCHK(c, hipMemsetAsync(c->d_last_di.p, 0, slots * 40, c->stream)); HIPCHK(c, hipMemsetAsync(c->d_last_gi.p, 0, slots * 40, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_last_sd.p, 0, slots * 60, c->stream));
        c->last_slots = slots;
    }
    // pass 1 writes its debug estimate into a scratch image (the displayed image is pass 3's)
    DevBuf& scratch = c->d_p1scratch; HIPCHK(c, scratch.ensure((size_t)p->width * p->height * 16));     // context-owned: no per-call hipMalloc / hipFree, nothing to leak on an early return
    stats_begin(c);
    struct LaneJoin { rtx_ctx* c; ~LaneJoin() { for (hipStream_t ls : c->lane_stream) if (ls) (void)hipStreamSynchronize(ls); } } lane_join{c};      // no early return leaves another lane running
    HIPCHK(c, hipMemsetAsync(c->d_p1cnt.p, 0, 32, c->stream));
    uint32_t* bufs[6] = {(uint32_t*)c->d_res_di.p, (uint32_t*)c->d_res_gi.p, (uint32_t*)c->d_sdata.p, (uint32_t*)c->d_last_di.p, (uint32_t*)c->d_last_gi.p, (uint32_t*)c->d_last_sd.p};
    if (!c->hist_all) {                       // the history this context holds does not cover the image (a sharded frame came before, and no all-gather since): count reads outside it
        f.hist_x0 = c->hist[0]; f.hist_y0 = c->hist[1]; f.hist_x1 = c->hist[2]; f.hist_y1 = c->hist[3];
        f.hist_stale = (unsigned long long*)c->d_p1cnt.p + 3;
    }
    const uint32_t mbk = (uint32_t)c->num_cus * 8u;
    const CameraGPU* cam = (const CameraGPU*)c->d_cam.p;
    HIPCHK(c, hipEventRecord(c->ev_begin, c->stream));
    for (uint32_t fr = 0; fr < p->spp; fr++) {                       // spp = number of consecutive frames with this camera
        DevFrame ff = f; ff.frame_seed = p->frame_seed + fr;
        HIPCHK(c, hipMemsetAsync(scratch.p, 0, (size_t)p->width * p->height * 16, c->stream));
        if (c->restir_wave) {                                                                                                               // the three DispatchRays of Renderer.cpp:646-673 as wavefront stages
            if ((r = rs_lanes(c, halo, nhalo, [&](const uint32_t* px, uint32_t n, uint32_t lane, hipStream_t st) { return rs_pass1(c, ff, 1u, (F4*)scratch.p, px, n, bufs, lane, st); }))) return r;   // passes 1 + 2
            if ((r = rs_lanes(c, own, nown, [&](const uint32_t* px, uint32_t n, uint32_t lane, hipStream_t st) { return rs_pass3(c, ff, bufs, c->accum_ptr(), px, n, lane, st); }))) return r;
        } else {                                                                                                                            // ... or literally, a thread per pixel
            { Timed t(c, RTX_K_BOUNCE); launch_v6_pass1(c->stream, mbk, c->dsc, ff, cam, 1u, (F4*)scratch.p, bufs[0], bufs[1], bufs[2], (unsigned long long*)c->d_p1cnt.p, sharded ? halo : nullptr, sharded ? nhalo : 0u); }   // Renderer.cpp:651-654
            { Timed t(c, RTX_K_BOUNCE); launch_restir_pass2(c->stream, mbk, c->dsc, ff, cam, bufs, (unsigned long long*)c->d_p1cnt.p, sharded ? halo : nullptr, sharded ? nhalo : 0u); }                                       // :662-664
            { Timed t(c, RTX_K_BOUNCE); launch_restir_pass3(c->stream, mbk, c->dsc, ff, cam, bufs, c->accum_ptr(), (unsigned long long*)c->d_p1cnt.p); }                                    // :671-673
        }
    }
    HIPCHK(c, hipEventRecord(c->ev_end, c->stream));
    HIPCHK(c, hipGetLastError());
    unsigned long long cnt[4] = {0, 0, 0, 0};
    TO_HOST(c, cnt, c->d_p1cnt.p, 32);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    stats_end_restir(c, cnt);
    c->stats.restir_stale_history_reads = cnt[3];
    // pass 3 wrote this frame's history for the pixels it ran on: the whole image, or — on shards — the own tiles (one rectangle in the block deal; the round-robin deal
    // has no rectangle to describe them: every temporal read counts as stale until rtx_restir_unpack_state has brought the other ranks' tiles)
    c->hist_all = !sharded;
    if (sharded) {
        if (f.blk_gx) block_rect(p->width, p->height, f.tile_size, f.tiles_x, f.tiles_y, f.blk_gx, f.blk_gy, f.shard_rank, c->hist);
        else c->hist[0] = c->hist[1] = c->hist[2] = c->hist[3] = 0;
    }
    return RTX_OK;
}

// ---- ReSTIR on shards: exchange of the history (u3 / u5 / u7) of the shard's own tiles, see rtx_render_restir ----
int rtx_restir_state_slab_bytes(const rtx_params* p, size_t* bytes) {
    if (!bytes) return RTX_ERR_INVALID;
    uint32_t ts = 0, cnt = 0; uint64_t npl = 0;
    if (const char* e = validate_tiling(p, ts, cnt, npl)) { g_create_err = e; return RTX_ERR_INVALID; }
    *bytes = (size_t)npl * 140;             // 40 + 40 + 60 bytes per local pixel slot
    return RTX_OK;
}
static int restir_state_bufs(rtx_ctx* c, const rtx_params* p, DevFrame& f, uint32_t* bufs[6]) {
    int r = make_frame(c, p, f); if (r) return r;
    const size_t slots = rtx_pass1_slots(p->width, p->height);
    if (!c->last_slots || c->last_slots != slots) { c->err = "restir state: no ReSTIR history of that image size (render a frame first)"; return RTX_ERR_STATE; }
    bufs[0] = (uint32_t*)c->d_res_di.p; bufs[1] = (uint32_t*)c->d_res_gi.p; bufs[2] = (uint32_t*)c->d_sdata.p;
    bufs[3] = (uint32_t*)c->d_last_di.p; bufs[4] = (uint32_t*)c->d_last_gi.p; bufs[5] = (uint32_t*)c->d_last_sd.p;
    return RTX_OK;
}
int rtx_restir_pack_state(rtx_ctx* c, const rtx_params* p, void* slab) {
    BIND(c);
    DevFrame f; uint32_t* bufs[6];
    int r = restir_state_bufs(c, p, f, bufs); if (r) return r;
    if (!slab) return RTX_ERR_INVALID;
    launch_restir_pack_state(c->stream, (uint32_t)c->num_cus * 8u, f, bufs, (uint32_t*)slab);
    HIPCHK(c, hipGetLastError());
    if (c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));      // on a caller-bound stream the gather that follows is stream-ordered
    return RTX_OK;
}
int rtx_restir_unpack_state(rtx_ctx* c, const rtx_params* p, const void* slabs) {
    BIND(c);
    DevFrame f; uint32_t* bufs[6];
    int r = restir_state_bufs(c, p, f, bufs); if (r) return r;
    if (!slabs) return RTX_ERR_INVALID;
    launch_restir_unpack_state(c->stream, (uint32_t)c->num_cus * 8u, f, f.shard_count, (const uint32_t*)slabs, bufs);
    HIPCHK(c, hipGetLastError());
    if (c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    c->hist_all = true;                     // every rank's tiles are here now
    return RTX_OK;
}

// ---- ... or of its border strips only (rtx.h: rtx_restir_pack_halo) ----
// Peers of rank r in the block deal: every rank q != r whose rectangle comes within halo_px of r's.  send = rect(r) ∩ dilate(rect(q)), recv = rect(q) ∩ dilate(rect(r)).
struct HaloPlan { std::vector<rtx_halo_peer> peers; uint64_t send_total = 0, recv_total = 0; uint32_t own[4] = {0, 0, 0, 0}; };
static const char* halo_plan(const rtx_params* p, uint32_t halo, HaloPlan& P) {
    uint32_t ts = 0, cnt = 0, gx = 0, gy = 0; uint64_t npl = 0;
    if (const char* e = validate_tiling(p, ts, cnt, npl, &gx, &gy)) return e;
    if (cnt < 2 || !gx) return "halo exchange: needs shard_count > 1 and RTX_FLAG_BLOCK_TILES (one rectangle of tiles per rank)";
    if (halo == 0 || halo > 4096) return "halo exchange: halo_px must be in [1, 4096]";
    const uint32_t W = p->width, H = p->height, TX = (W + ts - 1) / ts, TY = (H + ts - 1) / ts;
    block_rect(W, H, ts, TX, TY, gx, gy, p->shard_rank, P.own);
    auto clip = [](const uint32_t a[4], const uint32_t b[4], uint32_t grow, uint32_t W_, uint32_t H_, uint32_t out[4]) {      // a ∩ dilate(b, grow); false: empty
        const uint32_t bx0 = b[0] > grow ? b[0] - grow : 0u, by0 = b[1] > grow ? b[1] - grow : 0u, bx1 = std::min(W_, b[2] + grow), by1 = std::min(H_, b[3] + grow);
        out[0] = std::max(a[0], bx0); out[1] = std::max(a[1], by0); out[2] = std::min(a[2], bx1); out[3] = std::min(a[3], by1);
        return out[0] < out[2] && out[1] < out[3];
    };
    P.peers.clear(); P.send_total = P.recv_total = 0;
    if (P.own[0] >= P.own[2] || P.own[1] >= P.own[3]) return nullptr;                          // a rank without pixels (more ranks than tile columns): no peers
    for (uint32_t q = 0; q < cnt; q++) {
        if (q == p->shard_rank) continue;
        uint32_t rq[4], sr[4], rr[4]; block_rect(W, H, ts, TX, TY, gx, gy, q, rq);
        if (rq[0] >= rq[2] || rq[1] >= rq[3]) continue;
        if (!clip(P.own, rq, halo, W, H, sr) || !clip(rq, P.own, halo, W, H, rr)) continue;         // (both are empty or neither is: the dilation is symmetric)
        rtx_halo_peer e{}; e.rank = q;
        e.send_x0 = sr[0]; e.send_y0 = sr[1]; e.send_x1 = sr[2]; e.send_y1 = sr[3]; e.recv_x0 = rr[0]; e.recv_y0 = rr[1]; e.recv_x1 = rr[2]; e.recv_y1 = rr[3];
        e.send_offset = P.send_total; e.send_bytes = (uint64_t)(sr[2] - sr[0]) * (sr[3] - sr[1]) * 140u; P.send_total += e.send_bytes;
        e.recv_offset = P.recv_total; e.recv_bytes = (uint64_t)(rr[2] - rr[0]) * (rr[3] - rr[1]) * 140u; P.recv_total += e.recv_bytes;
        P.peers.push_back(e);
    }
    if (P.send_total / 140u > 0xFFFFFFFFull || P.recv_total / 140u > 0xFFFFFFFFull) return "halo exchange: regions too large";
    return nullptr;
}
int rtx_restir_halo_plan(const rtx_params* p, uint32_t halo_px, rtx_halo_peer* peers, uint32_t max_peers, uint32_t* npeers, uint64_t* send_total, uint64_t* recv_total) {
    HaloPlan P;
    if (const char* e = halo_plan(p, halo_px, P)) { g_create_err = e; return RTX_ERR_INVALID; }
    if (npeers) *npeers = (uint32_t)P.peers.size();
    if (send_total) *send_total = P.send_total;
    if (recv_total) *recv_total = P.recv_total;
    if (peers) {
        if (P.peers.size() > max_peers) { g_create_err = "halo plan: more peers than the caller's array holds"; return RTX_ERR_INVALID; }
        for (size_t i = 0; i < P.peers.size(); i++) peers[i] = P.peers[i];
    }
    return RTX_OK;
}
static int halo_move(rtx_ctx* c, const rtx_params* p, uint32_t halo_px, void* buf, bool pack) {
    DevFrame f; uint32_t* bufs[6];
    int r = restir_state_bufs(c, p, f, bufs); if (r) return r;
    HaloPlan P;
    if (const char* e = halo_plan(p, halo_px, P)) { c->err = e; return RTX_ERR_INVALID; }
    if (!buf && (pack ? P.send_total : P.recv_total)) return RTX_ERR_INVALID;
    for (size_t i = 0; i < P.peers.size(); i += kHaloPeers) {                                  // (<= 8 peers in practice: one launch)
        uint32_t rects[4 * kHaloPeers]; uint32_t n = 0;
        for (; n < kHaloPeers && i + n < P.peers.size(); n++) {
            const rtx_halo_peer& e = P.peers[i + n];
            rects[4 * n] = pack ? e.send_x0 : e.recv_x0; rects[4 * n + 1] = pack ? e.send_y0 : e.recv_y0;
            rects[4 * n + 2] = pack ? e.send_x1 - e.send_x0 : e.recv_x1 - e.recv_x0; rects[4 * n + 3] = pack ? e.send_y1 - e.send_y0 : e.recv_y1 - e.recv_y0;
        }
        const uint64_t off = pack ? P.peers[i].send_offset : P.peers[i].recv_offset;
        launch_restir_halo(c->stream, (uint32_t)c->num_cus * 8u, p->width, pack, rects, n, bufs, (uint32_t*)((char*)buf + off));
    }
    HIPCHK(c, hipGetLastError());
    if (c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));      // on a caller-bound stream the exchange that follows is stream-ordered
    if (!pack) {                                                        // the history now covers my rectangle + the halo (clipped to the image)
        c->hist_all = false;
        c->hist[0] = P.own[0] > halo_px ? P.own[0] - halo_px : 0u; c->hist[1] = P.own[1] > halo_px ? P.own[1] - halo_px : 0u;
        c->hist[2] = std::min(p->width, P.own[2] + halo_px); c->hist[3] = std::min(p->height, P.own[3] + halo_px);
    }
    return RTX_OK;
}
int rtx_restir_pack_halo(rtx_ctx* c, const rtx_params* p, uint32_t halo_px, void* send) { BIND(c); return halo_move(c, p, halo_px, send, true); }
int rtx_restir_unpack_halo(rtx_ctx* c, const rtx_params* p, uint32_t halo_px, const void* recv) { BIND(c); return halo_move(c, p, halo_px, const_cast<void*>(recv), false); }

int rtx_read_restir_last(rtx_ctx* c, void* di, void* gi, void* sd, size_t slots) {
    BIND(c);
    if (!c->last_slots || slots < c->last_slots) { c->err = "read_restir_last: no ReSTIR state or too few slots"; return RTX_ERR_INVALID; }
    if (di) TO_HOST(c, di, c->d_last_di.p, c->last_slots * 40);
    if (gi) TO_HOST(c, gi, c->d_last_gi.p, c->last_slots * 40);
    if (sd) TO_HOST(c, sd, c->d_last_sd.p, c->last_slots * 60);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

int rtx_read_pass1_buffers(rtx_ctx* c, void* di, void* gi, void* sd, size_t slots) {
    BIND(c);
    if (!c->p1_slots || slots < c->p1_slots) { c->err = "read_pass1_buffers: no pass-1 data or too few slots"; return RTX_ERR_INVALID; }
    if (di) TO_HOST(c, di, c->d_res_di.p, c->p1_slots * 40);
    if (gi) TO_HOST(c, gi, c->d_res_gi.p, c->p1_slots * 40);
    if (sd) TO_HOST(c, sd, c->d_sdata.p, c->p1_slots * 60);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

int rtx_read_accum(rtx_ctx* c, float* out, size_t bytes) {
    BIND(c);
    const size_t need = (size_t)c->acc_w * c->acc_h * 16;
    if (!out || !need || bytes < need || !c->accum_ptr()) { c->err = "read_accum: no image or buffer too small"; return RTX_ERR_INVALID; }
    TO_HOST(c, out, c->accum_ptr(), need);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

int rtx_read_srgb8(rtx_ctx* c, uint8_t* out, size_t bytes) {
    BIND(c);
    const uint32_t npix = c->acc_w * c->acc_h;
    if (!out || !npix || bytes < (size_t)npix * 4 || !c->accum_ptr()) { c->err = "read_srgb8: no image or buffer too small"; return RTX_ERR_INVALID; }
    HIPCHK(c, c->d_srgb.ensure((size_t)npix * 4));
    launch_srgb8(c->stream, c->accum_ptr(), npix, (uint32_t*)c->d_srgb.p);
    TO_HOST(c, out, c->d_srgb.p, (size_t)npix * 4);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

// gOutput layer `layer` of the reference's 30-layer output array (Renderer.h:298-299): 0 = the image (== rtx_read_srgb8); 10-17 = first-hit debug
// attributes (k_debug_layer); every other layer below 30 reads black, as in the reference whose shaders never write them
int rtx_read_layer(rtx_ctx* c, uint32_t layer, uint32_t width, uint32_t height, uint8_t* out, size_t bytes) {
    BIND(c);
    if (layer >= 30u) { c->err = "read_layer: the output array has 30 layers"; return RTX_ERR_INVALID; }
    if (layer == 0u) { if (width != c->acc_w || height != c->acc_h) { c->err = "read_layer: layer 0 has the size of the accumulation buffer"; return RTX_ERR_INVALID; } return rtx_read_srgb8(c, out, bytes); }
    const size_t npix = (size_t)width * height;
    if (!out || !npix || npix > 0x7FFFFFFFull || bytes < npix * 4) { c->err = "read_layer: bad size"; return RTX_ERR_INVALID; }
    if (layer < 10u || layer > 17u) { memset(out, 0, npix * 4); for (size_t i = 0; i < npix; i++) out[i * 4 + 3] = 255; return RTX_OK; }
    if (!c->committed || !c->camera_set) { c->err = "read_layer: scene not committed or camera not set"; return RTX_ERR_STATE; }
    HIPCHK(c, c->d_srgb.ensure(npix * 4));
    launch_debug_layer(c->stream, (uint32_t)c->num_cus * 8u, c->dsc, width, height, (const CameraGPU*)c->d_cam.p, layer, (uint32_t*)c->d_srgb.p);
    HIPCHK(c, hipGetLastError());
    TO_HOST(c, out, c->d_srgb.p, npix * 4);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

int rtx_get_stats(rtx_ctx* c, rtx_stats* out) { if (!out) return RTX_ERR_INVALID; BIND(c); *out = c->stats; return RTX_OK; }     // (joins a frame that RTX_OPT_ASYNC left in flight)

int rtx_get_lights(rtx_ctx* c, void* out80, uint32_t max_count, uint32_t* count_out) {
    if (!c) return RTX_ERR_INVALID;
    if (!c->committed) { c->err = "get_lights: scene not committed"; return RTX_ERR_STATE; }
    const uint32_t n = (uint32_t)(c->built.lights80.size() / 20);
    if (count_out) *count_out = n;
    if (out80) memcpy(out80, c->built.lights80.data(), (size_t)std::min(n, max_count) * 80);
    return RTX_OK;
}

int rtx_shard_slab_bytes(const rtx_params* p, size_t* bytes) {
    if (!bytes) return RTX_ERR_INVALID;
    uint32_t ts = 0, cnt = 0; uint64_t npl = 0;
    if (const char* e = validate_tiling(p, ts, cnt, npl)) { g_create_err = e; return RTX_ERR_INVALID; }   // no context here: message via rtx_last_error(NULL)
    *bytes = (size_t)npl * 16;
    return RTX_OK;
}
int rtx_pack_tiles(rtx_ctx* c, const rtx_params* p, void* slab) {
    BIND_NOWAIT(c);                       // stream-ordered behind an enqueued rtx_render (RTX_OPT_ASYNC): no host join
    DevFrame f; int r = make_frame(c, p, f); if (r) return r;
    if (!slab || !c->accum_ptr() || c->acc_w != p->width || c->acc_h != p->height) { c->err = "pack_tiles: no accumulation buffer of that size"; return RTX_ERR_STATE; }
    launch_pack_tiles(c->stream, (uint32_t)c->num_cus * 8u, f, c->accum_ptr(), (F4*)slab);
    HIPCHK(c, hipGetLastError());
    if (c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));     // on a caller-bound stream the gather that follows is stream-ordered: no host bubble
    return RTX_OK;
}
int rtx_unpack_tiles(rtx_ctx* c, const rtx_params* p, const void* slabs) {
    BIND_NOWAIT(c);                       // stream-ordered behind an enqueued rtx_render (RTX_OPT_ASYNC): no host join
    DevFrame f; int r = make_frame(c, p, f); if (r) return r;
    if (!slabs) return RTX_ERR_INVALID;
    if ((r = ensure_accum(c, p->width, p->height, false))) return r;
    launch_unpack_tiles(c->stream, (uint32_t)c->num_cus * 8u, f, f.shard_count, (const F4*)slabs, c->accum_ptr());
    HIPCHK(c, hipGetLastError());
    if (c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

// ---- kernel-level debug entry points ----

int rtx_debug_primary_rays(rtx_ctx* c, const rtx_params* p, uint32_t sample_id, float* rays8) {
    BIND(c);
    if (!c->camera_set) { c->err = "camera not set"; return RTX_ERR_STATE; }
    DevFrame f; int r = make_frame(c, p, f); if (r) return r;
    Scratch s; const size_t n = (size_t)p->width * p->height;
    HIPCHK(c, s.a.ensure(n * 32));
    launch_dbg_primary(c->stream, f, (const CameraGPU*)c->d_cam.p, sample_id, (F4*)s.a.p);
    TO_HOST(c, rays8, s.a.p, n * 32);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}
static int dbg_trace(rtx_ctx* c, const float* rays8, uint32_t n, int any, float* hits4, uint8_t* occ) {
    BIND(c);
    if (!c->committed) { c->err = "scene not committed"; return RTX_ERR_STATE; }
    if (!n) return RTX_OK;
    Scratch s;
    HIPCHK(c, s.a.ensure((size_t)n * 32)); HIPCHK(c, s.b.ensure((size_t)n * 16));
    TO_DEVICE(c, s.a.p, rays8, (size_t)n * 32);
    launch_dbg_trace(c->stream, c->dsc, (const F4*)s.a.p, n, any, (F4*)s.b.p);
    HIPCHK(c, hipGetLastError());
    std::vector<float> h((size_t)n * 4);
    TO_HOST(c, h.data(), s.b.p, (size_t)n * 16);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hits4) memcpy(hits4, h.data(), (size_t)n * 16);
    if (occ) for (uint32_t i = 0; i < n; i++) { uint32_t prim; memcpy(&prim, &h[(size_t)i * 4 + 3], 4); occ[i] = prim != kMissPrim; }
    return RTX_OK;
}
int rtx_debug_trace_closest(rtx_ctx* c, const float* rays8, uint32_t n, float* hits4) { return dbg_trace(c, rays8, n, 0, hits4, nullptr); }
int rtx_debug_trace_any(rtx_ctx* c, const float* rays8, uint32_t n, uint8_t* occluded) { return dbg_trace(c, rays8, n, 1, nullptr, occluded); }
int rtx_debug_validate_bvh(rtx_ctx* c) {
    BIND(c);
    if (!c->committed) { c->err = "scene not committed"; return RTX_ERR_STATE; }
    std::vector<Node8GPU> nodes(c->dsc.nnodes); std::vector<TriGPU> tris(c->dsc.ntris);
    if (!nodes.empty()) TO_HOST(c, nodes.data(), c->d_nodes.p, nodes.size() * sizeof(Node8GPU));
    if (!tris.empty()) TO_HOST(c, tris.data(), c->d_tris.p, tris.size() * sizeof(TriGPU));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the triangle the kernels intersect: (v0, v0 + e1, v0 + e2), filed under its global id (v0.w): a spatial split references a triangle from several leaf entries
    uint32_t ng = 0;
    for (const TriGPU& T : tris) ng = std::max(ng, f2u(T.v0.w) + 1u);
    std::vector<float> w((size_t)ng * 9, 0.0f); std::vector<uint32_t> ident(tris.size()), gid(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        const TriGPU& T = tris[i]; ident[i] = (uint32_t)i; gid[i] = f2u(T.v0.w); float* o = &w[(size_t)gid[i] * 9];
        o[0] = T.v0.x; o[1] = T.v0.y; o[2] = T.v0.z; o[3] = T.v0.x + T.e1.x; o[4] = T.v0.y + T.e1.y; o[5] = T.v0.z + T.e1.z; o[6] = T.v0.x + T.e2.x; o[7] = T.v0.y + T.e2.y; o[8] = T.v0.z + T.e2.z;
    }
    return validate_bvh8(w, nodes, gid, ident, nullptr);
}
int rtx_debug_tree_hash(rtx_ctx* c, uint64_t out2[2]) {
    BIND(c);
    if (!c->committed || !out2) { if (c) c->err = "scene not committed"; return RTX_ERR_STATE; }
    std::vector<uint8_t> nodes((size_t)c->n_nodes8 * sizeof(Node8GPU)), tris((size_t)c->n_tris8 * sizeof(TriGPU));
    if (!nodes.empty()) TO_HOST(c, nodes.data(), c->d_nodes.p, nodes.size());
    if (!tris.empty()) TO_HOST(c, tris.data(), c->d_tris.p, tris.size());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto fnv = [](const std::vector<uint8_t>& v) { uint64_t h = 1469598103934665603ull; for (uint8_t b : v) { h ^= b; h *= 1099511628211ull; } return h; };
    out2[0] = fnv(nodes); out2[1] = fnv(tris);
    return RTX_OK;
}
int rtx_debug_read_tree(rtx_ctx* c, int which, void* nodes, uint64_t nodes_bytes, void* tris, uint64_t tris_bytes) {
    BIND(c);
    if (!c->committed) { c->err = "scene not committed"; return RTX_ERR_STATE; }
    const size_t nb = (size_t)c->n_nodes8 * sizeof(Node8GPU), tb = (size_t)c->n_tris8 * sizeof(TriGPU);
    if ((nodes && nodes_bytes != nb) || (tris && tris_bytes != tb)) { c->err = "read_tree: buffers must hold rtx_stats.bvh_nodes * 80 and bvh_refs * 48 bytes"; return RTX_ERR_INVALID; }
    if (which == 0) {
        if (nodes) TO_HOST(c, nodes, c->d_nodes.p, nb);
        if (tris) TO_HOST(c, tris, c->d_tris.p, tb);
        return RTX_OK;
    }
    const BuiltScene& B = c->built;
    if (B.nodes8.size() != c->n_nodes8 || B.tris8.size() != c->n_tris8) { c->err = "read_tree: the host holds no mirror of this tree (built on the device, or loaded without one)"; return RTX_ERR_STATE; }
    if (nodes) memcpy(nodes, B.nodes8.data(), nb);
    if (tris) memcpy(tris, B.tris8.data(), tb);
    return RTX_OK;
}
int rtx_debug_read_host_build(rtx_ctx* c, void* nodes2, uint64_t* nodes2_bytes, void* leaf_order, uint64_t* leaf_order_bytes) {
    if (!c || !nodes2_bytes || !leaf_order_bytes) return RTX_ERR_INVALID;
    const BuiltScene& B = c->built;
    const uint64_t nb = (uint64_t)B.nodes.size() * sizeof(NodeGPU), lb = (uint64_t)B.leaf_order.size() * 4u;
    if (nodes2 && *nodes2_bytes >= nb) memcpy(nodes2, B.nodes.data(), (size_t)nb);
    if (leaf_order && *leaf_order_bytes >= lb) memcpy(leaf_order, B.leaf_order.data(), (size_t)lb);
    *nodes2_bytes = nb; *leaf_order_bytes = lb;
    return RTX_OK;
}
int rtx_debug_host_checksums(rtx_ctx* c, uint64_t out8[8]) {
    if (!c || !out8) return RTX_ERR_INVALID;
    auto fnv = [](uint64_t h, const void* d, size_t n) { const uint8_t* q = (const uint8_t*)d; for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; } return h; };
    for (int k = 0; k < 8; k++) out8[k] = 1469598103934665603ull;
    const SceneHost& H = c->host; const BuiltScene& B = c->built;
    for (const MeshHost& m : H.meshes) { out8[0] = fnv(out8[0], m.idx.data(), m.idx.size() * 4); out8[1] = fnv(out8[1], m.verts.data(), m.verts.size() * 4); }
    out8[2] = fnv(fnv(out8[2], H.matids.data(), H.matids.size() * 4), H.mats128.data(), H.mats128.size() * 4);
    out8[3] = fnv(out8[3], H.insts.data(), H.insts.size() * sizeof(InstHost));
    out8[4] = fnv(out8[4], B.leaf_order.data(), B.leaf_order.size() * 4);
    out8[5] = fnv(out8[5], B.nodes.data(), B.nodes.size() * sizeof(NodeGPU));
    out8[6] = fnv(fnv(out8[6], B.nodes8.data(), B.nodes8.size() * sizeof(Node8GPU)), B.tris8.data(), B.tris8.size() * sizeof(TriGPU));
    out8[7] = fnv(fnv(fnv(out8[7], B.shade.data(), B.shade.size() * sizeof(TriShade)), B.objtris.data(), B.objtris.size() * sizeof(F4)), B.tri_slots8.data(), B.tri_slots8.size() * 4);
    return RTX_OK;
}
int rtx_debug_build_info(rtx_ctx* c, double ms5[5], uint32_t counts4[4]) {
    if (!c || !ms5 || !counts4) return RTX_ERR_INVALID;
    const GpuBuildResult& G = c->build_info;
    const bool g = c->dev_built;
    ms5[0] = g ? G.ms_prims : 0; ms5[1] = g ? G.ms_sort : 0; ms5[2] = g ? G.ms_ploc : 0; ms5[3] = g ? G.ms_top_host : 0; ms5[4] = g ? G.ms_layout : 0;
    counts4[0] = c->n_nodes8; counts4[1] = c->n_tris8; counts4[2] = g ? G.ploc_iterations : 0; counts4[3] = g ? G.clusters_top : 0;
    return RTX_OK;
}
// work counters of the persistent traversal kernels since they were last read (RTX_OPT_TRACE_COUNTERS 1): out4 = node steps and triangle tests of closest-hit rays, node steps and
// triangle tests of any-hit rays; reading resets them.  Rays per class: rtx_stats (rays_primary + rays_extension, rays_shadow)
int rtx_debug_trace_counters(rtx_ctx* c, uint64_t out4[4]) {
    BIND(c);
    if (!c->trace_counters || !c->d_trace_cnt.p) { c->err = "trace counters are off (RTX_OPT_TRACE_COUNTERS)"; return RTX_ERR_STATE; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->aux) HIPCHK(c, hipStreamSynchronize(c->aux));
    unsigned long long h[4];
    TO_HOST(c, h, c->d_trace_cnt.p, 32);
    HIPCHK(c, hipMemset(c->d_trace_cnt.p, 0, 32));
    for (int i = 0; i < 4; i++) out4[i] = h[i];
    return RTX_OK;
}
int rtx_debug_trace_stats(rtx_ctx* c, const float* rays8, uint32_t n, float* stats4) { return dbg_trace(c, rays8, n, 2, stats4, nullptr); }

int rtx_debug_surface(rtx_ctx* c, const float* rays8, const float* hits4, uint32_t n, float* out16) {
    BIND(c);
    if (!c->committed) { c->err = "scene not committed"; return RTX_ERR_STATE; }
    if (!n) return RTX_OK;
    Scratch s;
    HIPCHK(c, s.a.ensure((size_t)n * 32)); HIPCHK(c, s.b.ensure((size_t)n * 16)); HIPCHK(c, s.c.ensure((size_t)n * 64));
    TO_DEVICE(c, s.a.p, rays8, (size_t)n * 32);
    TO_DEVICE(c, s.b.p, hits4, (size_t)n * 16);
    launch_dbg_surface(c->stream, c->dsc, (const F4*)s.a.p, (const F4*)s.b.p, n, (F4*)s.c.p);
    HIPCHK(c, hipGetLastError());
    std::vector<float> h((size_t)n * 16);
    TO_HOST(c, h.data(), s.c.p, (size_t)n * 64);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // device layout: pos3,mat | normal3,area | inst,flat3 | 0  ->  API layout: pos3,mat,normal3,area,inst,flat3,pad4
    memcpy(out16, h.data(), (size_t)n * 64);
    return RTX_OK;
}
static int dbg_bsdf(rtx_ctx* c, bool sample, uint32_t mat, uint32_t flags, const float* in, uint32_t stride_in, uint32_t n, float* out8) {
    BIND(c);
    if (!c->committed) { c->err = "scene not committed"; return RTX_ERR_STATE; }
    if (mat >= c->dsc.nmat) { c->err = "material id out of range"; return RTX_ERR_INVALID; }
    if (!n) return RTX_OK;
    Scratch s;
    HIPCHK(c, s.a.ensure((size_t)n * stride_in * 4)); HIPCHK(c, s.b.ensure((size_t)n * 32));
    TO_DEVICE(c, s.a.p, in, (size_t)n * stride_in * 4);
    if (sample) launch_dbg_bsdf_sample(c->stream, c->dsc, mat, flags, (const float*)s.a.p, n, (float*)s.b.p);
    else launch_dbg_bsdf_eval(c->stream, c->dsc, mat, flags, (const float*)s.a.p, n, (float*)s.b.p);
    HIPCHK(c, hipGetLastError());
    TO_HOST(c, out8, s.b.p, (size_t)n * 32);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}
int rtx_debug_bsdf_eval(rtx_ctx* c, uint32_t mat, uint32_t flags, const float* in9, uint32_t n, float* out8) { return dbg_bsdf(c, false, mat, flags, in9, 9, n, out8); }
int rtx_debug_bsdf_sample(rtx_ctx* c, uint32_t mat, uint32_t flags, const float* in8, uint32_t n, float* out8) { return dbg_bsdf(c, true, mat, flags, in8, 8, n, out8); }

int rtx_debug_tea(rtx_ctx* c, uint32_t seed[2], uint32_t n, float* out) {
    BIND(c);
    if (!seed || !out) return RTX_ERR_INVALID;
    Scratch s;
    HIPCHK(c, s.a.ensure((size_t)std::max<uint32_t>(n, 1) * 4)); HIPCHK(c, s.b.ensure(8));
    launch_dbg_tea(c->stream, seed[0], seed[1], n, (float*)s.a.p, (uint32_t*)s.b.p);
    HIPCHK(c, hipGetLastError());
    if (n) TO_HOST(c, out, s.a.p, (size_t)n * 4);
    TO_HOST(c, seed, s.b.p, 8);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTX_OK;
}

}  // extern "C"
