// rtx_kernels.hpp — kernel argument blocks and host-side launchers (implemented in rtx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "rtx_types.hpp"

namespace rtx {

// read-only scene in HBM
struct DevScene {
    const Node8GPU* nodes;  uint32_t nnodes;
    const F4* nodes_f; uint32_t node_v4;      // what the traversal fetches: the same nodes at a stride of node_v4 x 16 B (5: `nodes` itself; 8 = RTX_OPT_NODE_STRIDE 128: one node per 128-B line)
    const TriGPU*   tris;   uint32_t ntris;
    const TriShade* shade;
    const SmallRecPair* small; uint32_t nsmall;   // nsmall != 0: tiny scene: nsmall pre-test records (planar polygons), no BVH
    uint32_t nsmall_occ;                          // records [0, nsmall_occ) can occlude a segment between two scene points; the rest are faces of the scene's convex hull
    const F4* small_poly;                         // 4 polygon corners per record (packet culling of primary rays)
    const TriGPU* small_tris;                     // 2 triangles per record (staged in LDS instead of `tris`)
    float small_cm, small_delta;                  // t-margin coefficient, distance tolerance of the edge planes
    const MatGPU*   mats;   uint32_t nmat;
    const InstGPU*  insts;  uint32_t ninst;
    const LightGPU* lights; uint32_t nlights;
    const float* cdf;               // lights[i].cdf as a dense array: the binary search of the light selection probes 4-byte entries side by side instead of one 80-byte record per probe
    float total_weight;
    uint32_t lds_nodes, lds_tris;   // how many nodes / triangles each workgroup stages in LDS
    uint32_t stack_depth;           // per-lane traversal stack entries IN LDS
    unsigned long long* stack_ovf; uint32_t stack_ovf_stride;   // RTX_OPT_STACK_CAP: entries beyond stack_depth live here, entry k of lane l at [k * stride + (l & (stride - 1))] (nullptr: the tree needs no more than stack_depth)
    uint32_t stack_private;         // queue kernels: 0 = stack in the LDS column, 1 = private (scratch) array
    uint32_t sort_materials;        // 1 = material-sorted shading in k_shade (general path; tuning knob, default 0)
    uint32_t refill_min, trace_sched;   // persistent traversal: idle lanes that trigger a refill; wave schedule (rtx_kernels.hip)
    uint32_t shade_dense;               // general path: k_shade compacts the hits of its sub-queue through an LDS ring before shading them (k_shade_dense)
    uint32_t any_order;                 // any-hit rays: visiting order of a node's hit children, 0 slot order / 1 nearest octant first / 2 farthest first (rtx_traverse.hpp: node8_hits)
    float cell_o[3], cell_s[3]; uint32_t cell_bits;     // RTX_OPT_OCTANT_SORT 3: grid over the scene's box, cell = (pos - cell_o) * cell_s per axis, bits per axis x | y << 4 | z << 8 (8 in all)
    uint32_t any_order_occ;             // ... of the ReSTIR stages' visibility rays (k_trace_shadow<.., SINK 1>): 0 unless RTX_OPT_ANYHIT_ORDER forces an order
    unsigned long long* trace_cnt;      // RTX_OPT_TRACE_COUNTERS: [0] node steps, [1] triangle tests of closest-hit rays, [2], [3] of any-hit rays, summed by the generic traversal instantiations; nullptr = off
    uint32_t occluder_cache;            // any-hit rays: a lane tests the triangle that occluded its previous ray first (rtx_traverse.hpp: ray_begin)
};

// one sample batch of one frame
struct DevFrame {
    uint32_t width, height;
    uint32_t tile_size, tile_shift, tiles_x, tiles_y;   // tile_size = 1 << tile_shift
    uint32_t shard_rank, shard_count;
    uint32_t blk_gx, blk_gy; // RTX_FLAG_BLOCK_TILES: the ranks as a blk_gx x blk_gy grid of tile rectangles (0: tile t -> rank t mod shard_count)
    uint32_t npl;            // local pixel slots = tiles_per_shard * tile_size^2
    uint32_t batch_spp;      // samples in this batch
    uint32_t sample_first;   // sample id of the first one
    uint32_t max_bounces, nee_samples, rr_start, frame_seed, flags;
    // work distribution: `nblocks` workgroups, each owning a private sub-queue of `qcap` entries
    uint32_t nblocks, qcap, chunks_per_sample;   // chunks_per_sample = npl / 256
    uint32_t interleave;     // RTX_OPT_SAMPLE_INTERLEAVE (k_raygen): 0 = a chunk of 256 queue entries is 256 pixel slots of ONE sample; s > 0 = 256 >> s pixel slots x 2^s consecutive samples, a pixel's samples in neighbouring lanes
    uint32_t taper_levels;   // k_raygen: 0 = chunks dealt evenly (chunk c -> sub-queue c mod nblocks); L > 0 = tapered deal with L weight classes (taper_row_width below)
    // ReSTIR on shards: the pixel rectangle [hist_x0, hist_x1) x [hist_y0, hist_y1) in which this context holds last frame's history (rtx_api.hip: rtx_ctx::hist); the temporal
    // pass counts its reads outside it in *hist_stale (nullptr: not counted — unsharded frames hold the whole image)
    uint32_t hist_x0, hist_y0, hist_x1, hist_y1; unsigned long long* hist_stale;
};

// TAPERED sub-queue sizes (RTX_OPT_TAPER).  Workgroups are dispatched in index order and every launch of a bounce ends when its LAST workgroup does; with equal sub-queues
// the last round of workgroups drains over one workgroup lifetime (~1 ms of a 4.8-ms launch at 32 sub-queues per CU: tools/wave_timeline.py shows 96 % of the peak number of
// waves until 83 % of the launch, then a ramp down: 12 % of the launch's wave slots idle).  So the sub-queues get SHORTER towards the end of the dispatch order: with L levels the
// index ranges [0, G/2) | [G/2, 3G/4) | ... (each half of what is left, the last level the rest) carry the weights 2^(L-1) | 2^(L-2) | ... | 1 — the long ones start first,
// the short ones fill the end.  The sizes persist through the bounces (survivors are a near-constant fraction), so every launch of the frame is tapered.
// The deal keeps what the even deal (chunk c -> sub-queue c mod G) has: chunks go out in ROWS of consecutive chunks, one per sub-queue in index order, so that neighbouring
// sub-queues — which run at the same time — hold neighbouring pixels of the same sample and walk the same part of the tree together (a scattered deal of the same sizes measured
// 2.5-4.6 % SLOWER than the even deal).  A sub-queue of weight w takes part in the rows k with (k mod 2^(L-1)) < w; the sub-queues taking part in a row are a prefix [0, n_k).
#if defined(__HIPCC__)
#define RTX_TAPER_HD __host__ __device__ __forceinline__
#else
#define RTX_TAPER_HD inline
#endif
RTX_TAPER_HD uint32_t taper_row_width(uint32_t k, uint32_t G, uint32_t levels) {      // n_k: sub-queues [0, n_k) get one chunk each in row k
    const uint32_t m = k & ((1u << (levels - 1u)) - 1u);
    if (m == 0u) return G;
    uint32_t lg = 0; while ((m >> (lg + 1u)) != 0u) lg++;                              // floor(log2 m)
    return G - (G >> (levels - 1u - lg));                                              // first index of the first class whose weight is <= m
}
// THE rule "k-th tile of a shard -> tile coordinates" (slot_to_pixel on the device, the halo list and the slab sizes on the host).  false: the shard has no such tile
// (its slot range is padded to the same length for every rank, so that the all-gather of the slabs has equal counts).
#if defined(__HIPCC__)
#define RTX_TILE_HD __host__ __device__ __forceinline__
#else
#define RTX_TILE_HD inline
#endif
RTX_TILE_HD bool shard_tile(const DevFrame& f, uint32_t k, uint32_t& tx, uint32_t& ty) {
    if (f.blk_gx) {                                                   // rectangle (bx, by) of the block grid: tile columns [bx TX / gx, (bx + 1) TX / gx), rows alike
        const uint32_t bx = f.shard_rank % f.blk_gx, by = f.shard_rank / f.blk_gx;
        const uint32_t tx0 = bx * f.tiles_x / f.blk_gx, tx1 = (bx + 1u) * f.tiles_x / f.blk_gx, ty0 = by * f.tiles_y / f.blk_gy, ty1 = (by + 1u) * f.tiles_y / f.blk_gy;
        const uint32_t bw = (f.tiles_x + f.blk_gx - 1u) / f.blk_gx;   // widest rectangle: the row length of the local tile index
        tx = tx0 + k % bw; ty = ty0 + k / bw;
        return tx < tx1 && ty < ty1;
    }
    const uint32_t t = f.shard_rank + k * f.shard_count;
    if (t >= f.tiles_x * f.tiles_y) return false;
    ty = t / f.tiles_x; tx = t - ty * f.tiles_x;
    return true;
}

// per-path state, SoA float4 streams indexed by path slot (pid = s_local * npl + pl)
struct DevPaths {
    F4* ray_o;   // origin.xyz, seed.y bits   (tmin is a function of the bounce index)
    F4* ray_d;   // dir.xyz, pdf of the BSDF sample that produced this ray
    F4* thr;     // throughput.xyz, seed.x bits
    F4* rad;     // radiance.xyz, -           (touched only when something is added)
    F4* hit;     // t, u, v, global triangle id bits
    unsigned long long* hitmask;   // fused tiny-scene path: bit (pid & 63) of word pid >> 6 = the primary ray hit something (rad is
                                   // initialised only for those; k_accumulate treats the others as zero).  nullptr: rad is zeroed for all
    // shadow queues: [nee slot j][workgroup b][qcap] entries
    F4* sh_o;    // origin.xyz, tmin
    F4* sh_d;    // dir.xyz, tmax
    F4* sh_c;    // contribution.xyz, pid bits
    // COMPACT path state (separate trace / shade kernels, RTX_OPT_COMPACT_STATE): ray_o / ray_d / thr / hit are indexed by the QUEUE POSITION
    // (sub-queue * qcap + entry) instead of the path id, and k_shade writes the state of a surviving path at its position in the NEXT queue
    // into the other buffer set (out_*).  A wave's state accesses stay one contiguous run however many paths died before (by path id the
    // bounces after Russian roulette touched one 16-B record per 128-B line), and the trace kernels need no queue read before they can
    // fetch a ray.  rad stays indexed by path id.
    F4* out_o; F4* out_d; F4* out_thr;        // nullptr: state indexed by path id, updated in place
    // RTX_OPT_OCTANT_SORT (compact state only): k_shade notes the direction octant of every survivor at its place in the next queue (oct_out); the closest-hit kernel of the
    // next bounce sorts its sub-queue's entries by that byte in a prologue (perm: entry order -> queue position) and fetches its rays through perm, so that a wave's lanes
    // hold rays of one octant for long runs.  Hit records are still written at the entry's own position: k_shade reads its streams in order as before.  nullptr = off
    uint8_t* oct_out; const uint8_t* oct_in; uint32_t* perm; uint32_t key_mode;     // key_mode: RTX_OPT_OCTANT_SORT's value (1 octant, 3 origin cell)
};

// work area of the wavefront ReSTIR stages (rtx_restir_wave.hpp), device pointers; one pass at a time
constexpr uint32_t kMaxMerge = 8;       // most sub-queues one workgroup of a traversal launch takes (MergedQ, rtx_traverse.hpp)
constexpr uint32_t kRsStreams = 7;      // ray_o (origin, seed.y) | ray_d (direction, seed.x) | a0 (normal, material) | a1 (outgoing, item) | a2 | a3 | a4
constexpr uint32_t kRsOcc = 16;         // occlusion bytes per item (pass 3 uses 10)
constexpr uint32_t kRsCand = 10;        // pass 3: candidate record, dwords per item
struct RsQ {
    uint32_t nitems; const uint32_t* pixels;        // work list of this pass (nullptr: the shard's own slots, slot_to_pixel)
    uint32_t G, qcap, rcap;                         // workgroups = private sub-queues; path entries / ray entries per sub-queue
    F4* st[2][kRsStreams];                          // path state by queue position, two sets
    F4* hit;                                        // closest hit of the entry at that position (current set)
    uint32_t* cls;                                  // per item: 0 = pass 1 sampled nothing here (miss / light seen directly), else material id + 1
    F4* fin;                                        // per item: (acc_L, w_sum) when its path ended
    F4* cold;                                       // per item, 5 streams of nitems: xn | nn | x1s | x2s | (L2, selected)
    uint8_t* occ;                                   // per item kRsOcc bytes: ray k of the pass was occluded
    uint32_t* cand;                                 // per item kRsCand dwords (pass 3)
    F4* sh_o; F4* sh_d; uint32_t* sh_pay;           // any-hit ray queue [G][rcap]: (origin, tmin) (direction, tmax) index of the occlusion byte
    unsigned long long* rays;                       // primary, extension, shadow (rtx_stats)
};

size_t trace_lds_bytes(const DevScene& sc);
int trace_workgroups_per_cu(const DevScene& sc);   // occupancy of the persistent traversal kernels for this LDS layout (hipOccupancyMaxActiveBlocksPerMultiprocessor)
void launch_raygen(hipStream_t, const DevFrame&, const DevPaths&, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount, bool compact);
void launch_trace_closest(hipStream_t, const DevFrame&, const DevScene&, const DevPaths&, uint32_t bounce, const uint32_t* queue, const uint32_t* qcount, uint32_t* heads, uint32_t merge = 1);   // merge: consecutive sub-queues per workgroup (MergedQ)
void launch_packet_masks(hipStream_t, const DevScene&, const DevFrame&, const CameraGPU* cam, unsigned long long* masks);   // one 64-bit record mask per 8x8 pixel block of the shard
void launch_raygen_trace_small(hipStream_t, const DevScene&, const DevFrame&, const DevPaths&, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount, uint32_t* gencount, const unsigned long long* masks);
void launch_bounce_small(hipStream_t, const DevScene&, const DevFrame&, const DevPaths&, uint32_t bounce_first, uint32_t bounce_end,
                         uint32_t* queue_a, uint32_t* queue_b, uint32_t* qrows, uint32_t* srows, const uint32_t* order, bool ring = true);   // bounce 0 alone (reads the primary hits), or a range of later bounces; ring: hits go through the LDS ring
// general (BVH) path: trace -> shade -> shadow of a bounce range for every workgroup-private sub-queue in one launch (hitq: G * qcap indices of scratch)
void launch_bounce_bvh(hipStream_t, const DevScene&, const DevFrame&, const DevPaths&, uint32_t bounce_first, uint32_t bounce_end,
                       uint32_t* queue_a, uint32_t* queue_b, uint32_t* hitq, uint32_t* qrows, uint32_t* srows, const uint32_t* order);
void launch_order_queues(hipStream_t, const uint32_t* qcount, uint32_t G, uint32_t* order);   // longest sub-queue first (dispatch order of the fused kernels)
void launch_trace_shadow(hipStream_t, const DevFrame&, const DevScene&, const DevPaths&, uint32_t j, const uint32_t* shcount, uint32_t* heads, uint32_t merge = 1);
void launch_shade(hipStream_t, const DevScene&, const DevFrame&, const DevPaths&, uint32_t bounce,
                  const uint32_t* queue, const uint32_t* qcount, uint32_t* next_queue, uint32_t* next_count, uint32_t* shcounts);
// pixels / npixels (optional): an explicit work list (x | y << 16) instead of the shard's own tiles — ReSTIR on shards runs passes 1 and 2 on the tiles dilated by 20 px
void launch_v6_pass1(hipStream_t, uint32_t max_blocks, const DevScene&, const DevFrame&, const CameraGPU* cam, uint32_t sample_id,
                     F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata, unsigned long long* counters, const uint32_t* pixels = nullptr, uint32_t npixels = 0);
void launch_restir_pass2(hipStream_t, uint32_t max_blocks, const DevScene&, const DevFrame&, const CameraGPU* cam, uint32_t* const bufs[6], unsigned long long* counters,
                         const uint32_t* pixels = nullptr, uint32_t npixels = 0);
// the ReSTIR history (u3 / u5 / u7) of the shard's own tiles <-> [local slot][35 dwords] slab, for the per-frame all-gather of sharded ReSTIR
void launch_restir_pack_state(hipStream_t, uint32_t max_blocks, const DevFrame&, uint32_t* const bufs[6], uint32_t* slab);
void launch_restir_unpack_state(hipStream_t, uint32_t max_blocks, const DevFrame&, uint32_t nshards, const uint32_t* slabs, uint32_t* const bufs[6]);
constexpr uint32_t kHaloPeers = 16;      // rectangles per launch of k_restir_halo (a rank of the block deal has <= 8 neighbours)
void launch_restir_halo(hipStream_t, uint32_t max_blocks, uint32_t width, bool pack, const uint32_t* rects4 /* x0, y0, w, h per rectangle */, uint32_t n /* <= 16 */, uint32_t* const bufs[6], uint32_t* buf);
void launch_restir_pass3(hipStream_t, uint32_t max_blocks, const DevScene&, const DevFrame&, const CameraGPU* cam, uint32_t* const bufs[6], F4* accum, unsigned long long* counters);
// ---- the ReSTIR frame as wavefront stages (rtx_restir_wave.hpp); every launch has q.G workgroups unless noted; cnt_* / shcnt: one entry per workgroup ----
void launch_trace_occ(hipStream_t, const DevScene&, const RsQ&, const uint32_t* shcnt);                  // any-hit rays of the ray queue -> q.occ bytes
void launch_rs_raygen(hipStream_t, const DevFrame&, const RsQ&, const CameraGPU* cam, uint32_t sample_id, uint32_t* cnt_out);
void launch_rs_p1_ris(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, const uint32_t* cnt_in, uint32_t* cnt_out, F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata);
void launch_rs_p1_ris_finish(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, const uint32_t* cnt_in, uint32_t* cnt_out, uint32_t* shcnt, uint32_t* res_di, uint32_t* sdata);
void launch_rs_p1_first(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, const uint32_t* cnt_in, uint32_t* cnt_out);
void launch_rs_p1_loop(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, uint32_t set, uint32_t iter, const uint32_t* cnt_in, uint32_t* cnt_out);
// bufs != nullptr: a ReSTIR frame — the temporal pass (pass 2) of a pixel rides on these two stages (its rays join pass 1's shadow rays, its merge follows the pixel's finish)
void launch_rs_p1_emit_final(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, const CameraGPU* cam, uint32_t* const* bufs, uint32_t* shcnt);
void launch_rs_p1_finish(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata, const CameraGPU* cam, uint32_t* const* bufs);
void launch_rs_p3_keys(hipStream_t, const DevFrame&, const RsQ&, uint32_t* const bufs[6], F4* key_a, F4* key_b);      // the compact neighbour records of the spatial pass (rtx_restir_wave.hpp), written after passes 1 + 2
void launch_rs_p3_select(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, const CameraGPU* cam, uint32_t* const bufs[6], uint32_t* shcnt, F4* key_a = nullptr, F4* key_b = nullptr);   // key_a != nullptr: select on the records
void launch_rs_p3_merge(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, uint32_t* const bufs[6], uint32_t* shcnt);
void launch_rs_p3_shade(hipStream_t, const DevScene&, const DevFrame&, const RsQ&, uint32_t* const bufs[6], F4* accum);
void launch_accumulate(hipStream_t, uint32_t max_blocks, const DevFrame&, const DevPaths&, F4* accum);
void launch_srgb8(hipStream_t, const F4* accum, uint32_t npix, uint32_t* out);
void launch_debug_layer(hipStream_t, uint32_t max_blocks, const DevScene&, uint32_t width, uint32_t height, const CameraGPU* cam, uint32_t layer, uint32_t* out);
void launch_pack_tiles(hipStream_t, uint32_t max_blocks, const DevFrame&, const F4* accum, F4* slab);
void launch_unpack_tiles(hipStream_t, uint32_t max_blocks, const DevFrame&, uint32_t nshards, const F4* slabs, F4* accum);
// transform-only commit: re-derive world triangles and re-quantise the wide nodes on the GPU (level_start: host array, nlevels + 1 entries)
void launch_refit(hipStream_t, Node8GPU* nodes, const uint32_t* level_start, uint32_t nlevels, TriGPU* tris, uint32_t ntris, const TriShade* shade,
                  const InstGPU* insts, const F4* objtris, F4* node_aabb, uint32_t* scale_bits,
                  const uint32_t* moved = nullptr, uint8_t* tri_dirty = nullptr, uint8_t* node_dirty = nullptr);   // moved != nullptr: PARTIAL refit of the instances flagged in it (node_aabb must hold the previous refit's boxes)
void launch_dbg_trace(hipStream_t, const DevScene&, const F4* rays, uint32_t n, int any, F4* hits);
void launch_dbg_surface(hipStream_t, const DevScene&, const F4* rays, const F4* hits, uint32_t n, F4* out);
void launch_dbg_bsdf_eval(hipStream_t, const DevScene&, uint32_t mat, uint32_t flags, const float* in9, uint32_t n, float* out8);
void launch_dbg_bsdf_sample(hipStream_t, const DevScene&, uint32_t mat, uint32_t flags, const float* in8, uint32_t n, float* out8);
void launch_dbg_tea(hipStream_t, uint32_t s0, uint32_t s1, uint32_t n, float* out, uint32_t* seed_out);
void launch_dbg_primary(hipStream_t, const DevFrame&, const CameraGPU* cam, uint32_t sample_id, F4* rays);

}  // namespace rtx
