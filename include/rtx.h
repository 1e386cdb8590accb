/*
 * rtx.h — C-ABI of the MI355X wavefront path tracer (librtx_hip.so).
 *
 * This is the drop-in boundary for the path-tracing inner loop of ML200/RoyalTracer-DX.
 * The reference has no FFI; its de-facto boundary is "Renderer (C++) <-> shader resource
 * bindings" (Pathtracer/rdn/Renderer.cpp:953-976, 983-1008, heap built at :1195-1581).
 * Every entry point below replaces one of those bindings / host steps and cites it.
 * Plain pointers and sizes only; no C++/torch types; no exceptions cross this boundary.
 *
 * Conventions
 *   - 4x4 matrices: 16 floats, element (row r, col c) of the column-vector matrix at m[c*4+r].
 *     These are exactly the bytes the reference memcpy's into its constant/structured buffers
 *     (DirectXMath row-vector matrices stored row-major == glm column-major; Renderer.cpp:1722-1768,
 *     2091-2121) and that HLSL consumes as `mul(M, float4(p,1))`.
 *   - Material = 128 B (Pathtracer/src/Components/Vertex.h:14-23), Vertex = 28 B (Vertex.h:25-35),
 *     LightTriangle = 80 B (Renderer.h:113-124).
 *   - Return value 0 = ok, negative = RTX_ERR_*; message via rtx_last_error().
 *     (reference: ThrowIfFailed -> std::exception, DXSampleHelper.h:17-23)
 *   - A context is bound to ONE GPU and is not thread-safe; different contexts may be driven from
 *     different threads/processes (reference: single thread, one in-flight frame, Renderer.cpp:717-735).
 *   - There is NO CPU fallback: rtx_create fails with RTX_ERR_NO_DEVICE when HIP cannot open the device.
 */
#ifndef RTX_H
#define RTX_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_OK               0
#define RTX_ERR_INVALID     -1   /* bad argument / inconsistent scene arrays */
#define RTX_ERR_NO_DEVICE   -2   /* HIP runtime or device not available */
#define RTX_ERR_HIP         -3   /* a HIP call failed (message has the HIP error string) */
#define RTX_ERR_STATE       -4   /* call order violated (e.g. render before commit/camera) */
#define RTX_ERR_OOM         -5

typedef struct rtx_ctx rtx_ctx;   /* one per GPU; opaque */

/* Compile-time shader #defines of the reference (Common_v6.hlsl:1-28) + hard-wired host values
   (Main.cpp:25, Renderer.cpp:1730) become one POD handed to rtx_render. */
typedef struct rtx_params {
    uint32_t width, height;   /* DispatchRays dims == image size (Renderer.cpp:646-654) */
    uint32_t spp;             /* samples per pixel added by this call */
    uint32_t sample_base;     /* first sample id s in the seed formula (RayGen_v6_pass1.hlsl:76-77 uses 1) */
    uint32_t max_bounces;     /* path segments per sample (`bounces`, RayGen.hlsl:68,99) */
    uint32_t nee_samples;     /* light samples per bounce (`nee_samples`, Common_v6.hlsl:8) */
    uint32_t rr_start;        /* Russian roulette when bounce index > rr_start (`rr_threshold`, RayGen.hlsl:69,118) */
    uint32_t frame_seed;      /* stands in for uint(time) (RayGen_v6_pass1.hlsl:76-77; Renderer.cpp:1754-1760) */
    uint32_t flags;           /* RTX_FLAG_* */
    uint32_t tile_size;       /* shard tile edge in pixels: a power of two in [16, 1024] (0 => 64); anything else is RTX_ERR_INVALID everywhere */
    uint32_t shard_rank;      /* this context renders tiles t (row-major tile index) with t % shard_count == shard_rank (RTX_FLAG_BLOCK_TILES: one rectangle of tiles instead) */
    uint32_t shard_count;     /* 0 or 1 => whole image */
} rtx_params;

#define RTX_FLAG_LAMBERT_ONLY 1u  /* strategy 0 only, p_d = 1 (BRDF_v6.hlsl:7-70 bypassed) */
#define RTX_FLAG_JITTER       2u  /* legacy sub-pixel jitter (RayGen.hlsl:84-87); v6 shoots pixel corners (pass1:80-82) */
#define RTX_FLAG_TRANSMISSION 4u  /* EXTENSION: strategy 3 (rough dielectric transmission) for materials with dissolve Kd.w < 1 and Ni != 1; the reference
                                     has the strategy as a stub only (BRDF_v6.hlsl:5,28-29,44-47,85-87).  rtx_render only; ignored with LAMBERT_ONLY
                                     and by the literal pass-1 / ReSTIR modes.  Off: every result is the reference-faithful one */

#define RTX_FLAG_BLOCK_TILES  8u  /* sharding: the tiles of a shard form ONE rectangle (the shard_count ranks as a gx x gy grid of blocks, gx * gy = shard_count chosen
                                     for the smallest block perimeter; of two grids with the same perimeter the one with fewer columns) instead of tile t -> rank t mod shard_count.  Same pixels, same image; what changes is who owns
                                     which tile: round-robin balances a path-traced frame, blocks keep the 20-px halo of a ReSTIR frame small (rtx_render_restir).
                                     Every call that takes rtx_params (render, pack / unpack, slab sizes) follows the flag */

/* kernel classes for rtx_stats */
enum { RTX_K_RAYGEN = 0, RTX_K_TRACE = 1, RTX_K_SHADE = 2, RTX_K_SHADOW = 3, RTX_K_ACCUM = 4, RTX_K_SORT = 5,
       RTX_K_BOUNCE = 6 /* fused trace+shade+shadow kernels: k_bounce_small (tiny scenes), k_bounce_bvh (general path) */, RTX_K_COUNT = 8 };

typedef struct rtx_stats {
    uint64_t rays_primary, rays_extension, rays_shadow;  /* BVH queries issued by the last rtx_render */
    uint64_t paths;                                       /* pixel-samples started */
    double   kernel_ms[RTX_K_COUNT];                      /* summed hipEvent time per kernel class (timing option on) */
    uint64_t kernel_launches[RTX_K_COUNT];
    uint64_t kernel_items[RTX_K_COUNT];                   /* work items (rays / paths) processed per class */
    double   render_ms;                                   /* hipEvent time of the whole rtx_render on its stream */
    uint32_t bvh_nodes, triangles, lights, materials;
    uint64_t primary_hits;                                /* camera rays that hit the scene (items of the bounce-0 shading launch) */
    uint32_t bvh_refits, bvh_refs;                        /* commits since the last full BVH build that only refitted boxes; leaf entries of the tree (= triangles unless
                                                             spatial splits, RTX_OPT_BVH_SPLIT, reference some from several leaves) */
    uint64_t restir_stale_history_reads;                  /* rtx_render_restir on shards: temporal-pass reads of last frame's history at a pixel where this context does not hold it
                                                             (outside own rectangle + exchanged halo): must be 0, see rtx_restir_pack_halo */
} rtx_stats;

enum { RTX_OPT_KERNEL_TIMING = 1,    /* 0/1: bracket every launch with hipEvents (rtx_stats.kernel_ms) */
       RTX_OPT_PATHS_PER_BATCH = 2,  /* max pixel-samples in flight (queue capacity), default 128 Mi (~17 GB of path state and queues) */
       RTX_OPT_SORT_MATERIALS = 3,   /* 0/1: material-sorted shading (k_shade sorts its sub-queue chunks by material in LDS; default 0: measured slower) */
       RTX_OPT_LDS_NODES = 4,        /* BVH nodes staged in LDS per workgroup (top of tree) */
       RTX_OPT_SMALL_SCENE = 5,      /* 0/1: brute-force pre-test path for scenes of <= 64 triangles (default 1) */
       RTX_OPT_FUSED_BOUNCE = 6,     /* 0/1: with SMALL_SCENE, fuse trace+shade+shadow into one kernel per bounce (default 1) */
       RTX_OPT_BOUNCE_VARIANT = 7,   /* fused tiny-scene kernel, bounces >= 1: 0 (default) = trace phase pushes HITS into an LDS ring, shading runs on ring entries (full waves);
                                        1 = trace and shade the same 256 queue entries (the round-1 form) */
       RTX_OPT_REFILL_MIN = 8,       /* tuning: idle lanes that trigger a refill in the persistent BVH traversal (default 12) */
       RTX_OPT_STACK_PRIVATE = 9,    /* tuning: traversal stack 0 = LDS column, 1 = private (scratch) memory, (2 is accepted and means 0) */
       RTX_OPT_TRACE_SCHED = 10,     /* tuning: wave schedule of the BVH traversal, 0 = while-while, 1-4 = voted node / triangle steps, 5-7 = voted + speculative (default 6) */
       RTX_OPT_GPU_REFIT = 11,       /* 1 (default): a transform-only rtx_commit_scene refits the resident BVH on the GPU; 0: host refit + upload */
       RTX_OPT_RESTIR_WAVEFRONT = 19,/* 1 (default): rtx_render_v6_pass1 / rtx_render_restir run as wavefront stages (csrc/rtx_restir_wave.hpp: stage kernels of <= 128 VGPRs, every
                                        ray traversed by the persistent kernels of the path tracer); 0: the literal form, one thread per pixel and pass like the reference's raygen
                                        shaders (csrc/rtx_restir.hpp).  Byte-identical buffers and images either way */
       RTX_OPT_OCCLUDER_CACHE = 21,  /* 1: in the persistent any-hit traversal a lane first tests the triangle that occluded its previous ray (neighbouring queue entries are neighbouring
                                        pixels aiming at the same light / sample).  Any-hit is existence, so results are identical — and it is MEASURED SLOWER (same box, alternating:
                                        k_trace_shadow 11.63 -> 12.23 ms on C3, 10.21 -> 10.48 on C5, ReSTIR frames +1 %): the extra triangle step per ray costs more than
                                        the early exits save.  Default 0 */
       RTX_OPT_SHADE_DENSE = 23,     /* general path: 1 = k_shade compacts the HITS of its sub-queue through an LDS ring and shades full waves of them (k_shade_dense), 0 (default) = shades
                                        the queue entries in place (lanes whose ray missed idle).  Results identical.  MEASURED, same box alternating: no gain where it was meant to help
                                        (C5, 14-56 % of a bounce's rays miss: k_shade 8.20 vs 8.22 ms — the lanes are lost inside the NEE / BSDF sections, not at their entry) and
                                        slower where nearly every ray hits (C3: 6.51 -> 7.89 ms: two barriers and a second read of the hit record per 256 entries) */
       RTX_OPT_RESTIR_LANES = 22,    /* 2 (default): the pixels of a ReSTIR frame are processed as two independent halves on two streams (passes 1 + 2, then pass 3), so that the
                                        many short dependent launches of one half fill the launch tails of the other; 1: one chain.  Off while RTX_OPT_KERNEL_TIMING is on.
                                        Results identical */
       RTX_OPT_RESTIR_LANE_MIN = 29, /* pixel lists shorter than this many entries run as ONE chain whatever RTX_OPT_RESTIR_LANES says (default 65 536: below that the second stream has
                                        nothing to hide); tests lower it to exercise the cross-stream ordering on small frames */
       RTX_OPT_RESTIR_CHUNKS = 20,   /* tuning: 256-pixel chunks per workgroup (= private sub-queue) of the ReSTIR stages, default 4 */
       RTX_OPT_OVERLAP_SHADOW = 18,  /* 1 (default): general scenes run the shadow-ray kernel of bounce b on a second (internal) stream beside the closest-hit kernel of
                                        bounce b + 1; everything is joined into the context's stream before rtx_render returns.  Off while RTX_OPT_KERNEL_TIMING is on
                                        (overlapping launches have no per-kernel time).  Results identical */
       RTX_OPT_COMPACT_STATE = 16,   /* 1 (default): the separate trace / shade kernels keep ray, throughput and hit records by QUEUE POSITION in two buffer sets (a bounce
                                        writes its survivors densely at their place in the next queue) instead of by path id in place.  Results identical */
       RTX_OPT_WORK_STEALING = 15,   /* 1: a wave of the general-scene trace kernels whose sub-queue is exhausted continues with other sub-queues (global chunk cursors + an
                                        exhausted bitmap, pseudo-random victims) instead of draining.  Bit-identical; wave iterations fall 8 % (busy lanes 50.3 -> 55.7 of 64)
                                        but the frame is SLOWER (C3 56.6 vs 47.4 ms, C5 51.0 vs 44.7 ms): the tail iterations it removes are latency-bound and cost few issue
                                        slots next to the other workgroups' full waves, while with stealing every wave of the launch reaches its tail at the same time.
                                        Default 0 */
       RTX_OPT_FUSED_BVH = 14,       /* 1: general (BVH) scenes run trace -> shade -> shadow of all bounces in ONE launch per batch (k_bounce_bvh, phases separated by workgroup
                                        barriers).  Bit-identical, but MEASURED SLOWER than one launch per phase and bounce (C3 51.3 vs 47.7 ms, C5 46.2 vs 44.4 ms: a wave that has
                                        finished its phase keeps its SIMD slot while it waits for the slowest wave of its workgroup, and the kernel needs 86 VGPRs where the
                                        traversal kernels need 75), so the default is 0.  Needs RTX_OPT_TRACE_SCHED 5-7 */
       RTX_OPT_LPT_ORDER = 13,       /* tuning: 1 (default) = the fused tiny-scene kernels take their sub-queues longest first (shorter launch tails), 0 = in index order */
       RTX_OPT_TAPER = 25,           /* 1 (default) = the sub-queues of a batch get shorter towards the end of the dispatch order (weights 8 | 4 | 2 | 1 over the
                                        index ranges G/2 | G/4 | G/8 | G/8), so that the last workgroups of every launch are short ones; 0 = equal sub-queues; 2-8 = that many weight classes.
                                        Never changes a result */
       RTX_OPT_MERGE_RAYS = 24,      /* tuning, general scenes: a launch of the persistent traversal kernels that is predicted (from the previous rtx_render's counters) to hold fewer than
                                        this many rays per sub-queue gives each workgroup 2 / 4 / 8 consecutive sub-queues, down to one round of resident workgroups (the late bounces of
                                        a frame, after Russian roulette).  Default 1024; 0 = one sub-queue per workgroup always.  Never changes a result */
       RTX_OPT_OCTANT_SORT = 32,     /* 1: general scenes, compact state: the closest-hit kernel of bounce b >= 1 fetches the rays of its sub-queue grouped by direction octant (k_shade notes a survivor's
                                        octant, a prologue of the traversal kernel counting-sorts the sub-queue's entries by it).  3: the key is the cell of the ray's ORIGIN on a 256-cell grid over the scene's box, from bounce 2.
                                        2 and 5 are measurement variants (all keys zero; hashed keys).  Never changes a result; MEASURED slower in every form: profiles/r04_octsort_ab.md.  Default 0 */
       RTX_OPT_PARTIAL_REFIT = 37,   /* 1 (default): a transform-only commit re-derives the triangles of the instances whose transform changed and re-quantises only the nodes above them
                                        (the first refit after a build is a full one); 0: every refit touches the whole tree */
       RTX_OPT_LDS_NODES_CLOSEST = 36, /* BVH nodes staged in LDS by the path tracer's CLOSEST-HIT launches; default -1 = auto: for trees of at most 16 MB the first three levels of the wide tree (73 nodes)
                                        while six workgroups per CU remain (the shadow launches keep RTX_OPT_LDS_NODES' choice: more workgroups, fewer nodes); 0 = as the shadow launches.  Takes effect with the next rtx_commit_scene */
       RTX_OPT_RESTIR_KEYS = 35,     /* 1 (default): after passes 1 + 2 every pixel writes what a NEIGHBOUR's test of the spatial pass reads (x1, n1, material, validity flags, M; the GI sample) into
                                        two 32-B records, and the selection stage of pass 3 (up to 18 random neighbours per pixel) reads those instead of the 60-B + 40-B + 40-B records.
                                        Byte-identical; wavefront form only.  0: the selection reads the full records */
       RTX_OPT_NODE_STRIDE = 34,     /* how the traversal finds a BVH node in HBM.  80: the nodes as built, 80 B apart.  128: a copy with one node per 128-B cache line (no node straddles a line), refreshed
                                        after every build / refit.  0 (default) = auto: the copy is made for trees above 16 MB and fetched by the path tracer's closest-hit launches of bounces >= 1
                                        only (street scene, 3.8 M triangles: k_trace_closest -2.4 %; coherent rays and small trees prefer the dense layout: profiles/r04_node_stride_ab.md).
                                        Never changes a result.  Takes effect with the next rtx_commit_scene */
       RTX_OPT_SAMPLE_INTERLEAVE = 33, /* general scenes (k_raygen): which path sits where in the sub-queues.  0: a chunk of 256 queue entries is 256 pixel slots of ONE sample; 1 (default): 256 / S pixel
                                        slots x S consecutive samples with a pixel's samples in neighbouring lanes, S = the largest power of two <= 16 dividing the batch's sample count: the rays
                                        of a wave start closer together at every bounce (C5 k_trace_closest 18.4 -> 17.5 ms, C3 20.6 -> 20.1: profiles/r04_interleave_ab.md).
                                        Never changes a result: seeds come from pixel and sample id, sums are per path */
       RTX_OPT_ASYNC = 31,           /* 1: on a caller-bound stream (rtx_set_stream) rtx_render returns once the frame is ENQUEUED; rtx_pack_tiles / rtx_unpack_tiles (and the caller's collective)
                                        follow it in stream order with no host join in between.  Statistics (rtx_get_stats) and every other entry point join the frame first.  Default 0:
                                        rtx_render returns with the frame finished.  On the context's own stream the option is ignored */
       RTX_OPT_TRACE_COUNTERS = 30,  /* 1: the persistent traversal kernels count node steps and triangle tests (their generic instantiations; a few per cent slower), read with
                                        rtx_debug_trace_counters: work per ray for the bench record.  Default 0.  Never changes a result */
       RTX_OPT_BVH_REINSERT = 26,    /* BVH builder: passes of the insertion-based topology optimisation after the top-down SAH build (Bittner et al. 2013; default see DESIGN.md section 6c).
                                        Changes the tree, never a result; the next rtx_commit_scene rebuilds */
       RTX_OPT_BVH_SPLIT = 27,       /* BVH builder: spatial splits (Stich et al. 2009) where an object split leaves its two sides overlapping by more than value * 1e-9 of the scene's surface
                                        area; 0 (default) = never.  A split triangle is referenced from several leaves; closest hit = minimum over all triangles and any hit = existence,
                                        so no result changes.  The next rtx_commit_scene rebuilds */
       RTX_OPT_ANYHIT_ORDER = 28,    /* any-hit (shadow / visibility) rays visit the hit children of a node in 0 = slot order, 1 = nearest octant first, 2 = farthest octant first;
                                        -1 (default) = what a commit-time probe of 2 048 NEE-like segments on the host found cheapest for this scene and its lights.  Never changes a result */
       RTX_OPT_GPU_BUILD = 38,       /* 1: a geometry-changing rtx_commit_scene builds the BVH ON THE GPU (csrc/rtx_build.hip: Morton sort, PLOC clustering down to <= 16 384 clusters, the top
                                        of the tree by the host's SAH builder + re-insertion over those clusters, SAH collapse to 8-wide nodes and layout on the device, then the refit
                                        kernels) instead of on the host: what the reference's driver does for it in BottomLevelASGenerator.cpp:178-247 / TopLevelASGenerator.cpp:149-250.
                                        0 (default): host build.  Results never depend on the tree; the host-side mirror of the tree (scene cache save, host refit) is not kept */
       RTX_OPT_STACK_CAP = 39,       /* 11 (default): traversal-stack entries per lane that live in LDS; a tree whose exact stack bound is deeper keeps the rest in per-lane columns in
                                        global memory, so that LDS (the staged top of the tree, workgroups per CU) is sized for what almost every ray needs.  0 = the whole stack in LDS (until round 5).
                                        Never changes a result.  Takes effect with the next rtx_commit_scene */
       RTX_OPT_BLOCKS_PER_CU = 12    /* tuning: workgroups (= private sub-queues) per compute unit; default 0 = auto: 40 (tiny scenes) / 32 at full frame size (8 measured 4-7 % slower there: tail imbalance), fewer — down to 8 — when a batch is so
                                        small (a shard) that a sub-queue would start with fewer than ~16 / ~8 chunks of 256 paths */ };

/* lifetime: replaces LoadPipeline/device creation (Renderer.cpp:106-254) and OnDestroy (:546-552) */
int  rtx_create(int device_ordinal, rtx_ctx** out);
void rtx_destroy(rtx_ctx*);
const char* rtx_last_error(rtx_ctx*);          /* ctx may be NULL: the CALLING THREAD's last error of a context-free call (rtx_create, rtx_shard_slab_bytes, rtx_restir_state_slab_bytes) */
int  rtx_set_option(rtx_ctx*, int option, int64_t value);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream.
   replaces the single m_commandQueue (Renderer.cpp:192-199) */
int  rtx_set_stream(rtx_ctx*, void* hip_stream);   /* a caller-owned stream must outlive the work enqueued on it; rtx_destroy does not touch it */

/* t5 `materials` StructuredBuffer<Material>, 128 B stride (Renderer.cpp:373-389, 1287-1296) */
int  rtx_set_materials(rtx_ctx*, const void* mats128, uint32_t count);
/* t2 `BTriVertex` (28 B stride) + t1 `indices` (u32) per model (CreateVB, Renderer.cpp:1973-2072) and the
   model's slice of t4 `materialIDs` (one id per index, Renderer.cpp:391-407).  Vertex.normal.w must equal the
   number of material ids added before this mesh (ObjLoader.h:466 / Hit_v6.hlsl:17). */
int  rtx_add_mesh(rtx_ctx*, const void* verts28, uint32_t nverts, const uint32_t* indices, uint32_t nidx,
                  const uint32_t* material_ids, uint32_t* mesh_out);
/* m_instances.push_back({BLAS, matrix}) (Renderer.cpp:908-921); instance id = order of insertion (:843-845) */
int  rtx_add_instance(rtx_ctx*, uint32_t mesh, const float o2w[16], uint32_t* inst_out);
/* t3 `instanceProps` update (UpdateInstancePropertiesBuffer, Renderer.cpp:2091-2121; prevObjectToWorld := the old matrix);
   needs rtx_commit_scene again, which then only REFITS the BVH boxes (TLAS refit, Renderer.cpp:594) instead of rebuilding */
int  rtx_set_instance_transform(rtx_ctx*, uint32_t inst, const float o2w[16]);
/* CreateAccelerationStructures (Renderer.cpp:893-946) + CollectEmissiveTriangles (:2123-2213) +
   CreateEmissiveTrianglesBuffer (:2237-2280): BVH build, emissive CDF, upload */
int  rtx_commit_scene(rtx_ctx*);
/* SURVEY 8(f3) binary scene cache.  rtx_save_scene_cache writes the committed scene — what the calls above handed over AND what
   rtx_commit_scene derived for the device (MaterialOptimized table + Ess LUTs, compressed 8-wide BVH, leaf-ordered triangles, shading
   records, emissive CDF, tiny-scene records) — to one versioned, checksummed file.  rtx_load_scene_cache REPLACES the context's scene
   by the file's and uploads it (no BVH build: 3.8 M triangles in a fraction of the 2.2 s a commit takes); it returns RTX_ERR_INVALID and
   leaves the current scene untouched for a missing / truncated / corrupt file, another version or another record layout.  The
   reference has no such file (it rebuilds its BLAS / TLAS at start-up, Renderer.cpp:893-946). */
int  rtx_save_scene_cache(rtx_ctx*, const char* path);
int  rtx_load_scene_cache(rtx_ctx*, const char* path);
/* b0 `CameraParams` (UpdateCameraBuffer, Renderer.cpp:1722-1768): view + projection; inverses computed inside */
int  rtx_set_camera(rtx_ctx*, const float view[16], const float proj[16]);

/* u1 `gPermanentData` RGBA32F W x H (Renderer.cpp:1167-1186): xyz running sum, w sample count.
   rtx_bind_accum lets the caller own the device buffer (W*H*16 bytes, e.g. a torch tensor); NULL = internal. */
int  rtx_bind_accum(rtx_ctx*, void* device_rgba32f, size_t bytes);
int  rtx_clear_accum(rtx_ctx*, uint32_t width, uint32_t height);   /* view-change reset (RayGen_v6_pass3.hlsl:407-423) */
/* 3x DispatchRays (PopulateCommandList, Renderer.cpp:646-673) -> here: the wavefront loop; returns with the frame finished (RTX_OPT_ASYNC on a caller-bound stream: enqueued) */
int  rtx_render(rtx_ctx*, const rtx_params*);
/* The reference's OWN first pass, literally: RayGen of RayGen_v6_pass1.hlsl:48-190 (first DispatchRays,
   Renderer.cpp:651-654): primary hit, SampleRIS (Sampler_v6.hlsl:653-736), visibility, SamplePathSimple
   (Path_Sampler_v6.hlsl:3-286).  params: nee_samples = nee_samples_DI = nee_samples (Common_v6.hlsl:8-9, reference 4),
   max_bounces = `bounces` (:11, reference 3), spp samples are run one after the other; sdata.debug (or L1 on emissive
   primary hits) is added to u1.  The pass-1 outputs u2 `g_Reservoirs_current` / u4 `g_Reservoirs_current_gi` (40 B) and
   u6 `g_sample_current` (60 B) keep the last sample, in MapPixelID order (Common_v6.hlsl:173-198). */
int  rtx_render_v6_pass1(rtx_ctx*, const rtx_params*);
size_t rtx_pass1_slots(uint32_t width, uint32_t height);            /* records per buffer: ceil(w/4)*ceil(h/4)*16 */
int  rtx_read_pass1_buffers(rtx_ctx*, void* reservoirs_di40, void* reservoirs_gi40, void* samples60, size_t slots);
/* The reference's full frame: pass 1, then RayGen2 = temporal reuse (RayGen_v6_pass2.hlsl:46-204, second DispatchRays,
   Renderer.cpp:662-664) and RayGen3 = spatial reuse + final shade (RayGen_v6_pass3.hlsl:46-441, third DispatchRays, :671-673)
   with the pairwise MIS of MIS_v6.hlsl / MIS_GI_v6.hlsl.  `spp` = number of consecutive frames (frame_seed, frame_seed+1, ...)
   rendered with the current camera; each adds ReconnectDI*W + f_gi*W_gi to u1.  u3/u5/u7 (`g_*_last`) persist in the context
   between calls; rtx_set_camera keeps the previous view/projection for the reprojection; rtx_restir_reset zeroes the history.
   ON SHARDS (shard_count > 1; one frame per call: spp = 1): passes 1 and 2 run on the shard's tiles dilated by the 20-px radius of the spatial pass (the
   halo is recomputed, seeds depend on the pixel only), pass 3 and the accumulation on the shard's own tiles; between frames the shards exchange the history of
   their own tiles — rtx_restir_pack_state -> ONE all-gather (140 B per pixel) -> rtx_restir_unpack_state — because the temporal pass reprojects to arbitrary
   pixels.  Images and histories are bit-identical to the unsharded run. */
int  rtx_render_restir(rtx_ctx*, const rtx_params*);
int  rtx_restir_state_slab_bytes(const rtx_params*, size_t* bytes_per_shard);
int  rtx_restir_pack_state(rtx_ctx*, const rtx_params*, void* device_slab);                       /* u3 / u5 / u7 of my tiles -> slab */
int  rtx_restir_unpack_state(rtx_ctx*, const rtx_params*, const void* device_slabs_all_shards);   /* all shards' slabs -> my u3 / u5 / u7 */
/* HALO EXCHANGE of the history ("halo exchange (20 px) if tiles are sharded"; round 5) — instead of all-gathering every rank's 140 B per pixel (326 MB received
   per rank and frame at 1080p on 8 ranks), a rank of the RTX_FLAG_BLOCK_TILES deal sends each neighbour only the part of its own rectangle that lies within `halo_px` of the
   neighbour's rectangle, and receives the mirror image: <= 8 peers (point-to-point, one xGMI link each), ~10 MB sent per rank.  What makes it sufficient: passes 1 + 2 of the
   next frame run on the rank's rectangle dilated by the 20-px radius of the spatial pass, and the temporal pass reads the history at the REPROJECTED pixel — so the history must
   be valid in the rectangle dilated by 20 px + the largest reprojection displacement of a frame, i.e. halo_px >= 20 + that displacement (32 = one tile edge of the sharded deal
   covers 12 px of motion per frame).  The context KNOWS where its history is valid (own rectangle after a frame, + halo_px after rtx_restir_unpack_halo, the whole image after
   rtx_restir_unpack_state / rtx_restir_reset) and COUNTS every temporal read that lands outside it: rtx_stats.restir_stale_history_reads of the frame must be 0 — a caller that
   sees it rise (a camera cut) falls back to the all-gather (rtx_restir_pack_state / unpack_state) and repeats the frame.  With the count at 0, images and the history inside
   the valid region are bit-identical to the unsharded run.
   rtx_restir_halo_plan needs no context (message via rtx_last_error(NULL)): peers in ascending rank order, regions as pixel rectangles [x0, x1) x [y0, y1), records in
   row-major order, 140 B each (Reservoir_DI 40 | Reservoir_GI 40 | SampleData 60), offsets into ONE send and ONE receive buffer per rank.  send region of (me -> q) ==
   receive region of (q <- me) by construction (tests/test_multigpu_gloo.py pins the symmetry). */
typedef struct rtx_halo_peer {
    uint32_t rank;
    uint32_t send_x0, send_y0, send_x1, send_y1;     /* part of MY rectangle within halo_px of the peer's */
    uint32_t recv_x0, recv_y0, recv_x1, recv_y1;     /* part of the PEER's rectangle within halo_px of mine */
    uint64_t send_offset, send_bytes, recv_offset, recv_bytes;
} rtx_halo_peer;
int  rtx_restir_halo_plan(const rtx_params*, uint32_t halo_px, rtx_halo_peer* peers, uint32_t max_peers, uint32_t* npeers, uint64_t* send_bytes_total, uint64_t* recv_bytes_total);
int  rtx_restir_pack_halo(rtx_ctx*, const rtx_params*, uint32_t halo_px, void* device_send_buffer);            /* u3 / u5 / u7 of every send region -> send buffer (enqueued) */
int  rtx_restir_unpack_halo(rtx_ctx*, const rtx_params*, uint32_t halo_px, const void* device_recv_buffer);    /* receive buffer -> my u3 / u5 / u7; the history is valid in my rectangle + halo_px afterwards */
int  rtx_restir_reset(rtx_ctx*);
int  rtx_read_restir_last(rtx_ctx*, void* reservoirs_di40, void* reservoirs_gi40, void* samples60, size_t slots);   /* u3 / u5 / u7 */
int  rtx_read_accum(rtx_ctx*, float* rgba32f, size_t bytes);       /* copy of u1 to the host */
/* u0 `gOutput` layer 0, RGBA8 UNORM after sRGB OETF (RayGen_v6_pass3.hlsl:405,428-441; Common_v6.hlsl:353-376) */
int  rtx_read_srgb8(rtx_ctx*, uint8_t* rgba8, size_t bytes);
/* u0 `gOutput` is a 30-layer RGBA8 array and 'C' cycles the displayed layer through m_displayLevels = {0, 10..17, 20..28} (Renderer.h:298-299,
   Renderer.cpp:690-698, 748-754).  The reference's live shaders only write layer 0.  Here: 0 = the image, 10-17 = first-hit debug attributes of the
   pixel-corner primary ray (10 normal, 11 depth, 12 material id, 13 Kd, 14 instance id, 15 barycentrics, 16 Ke, 17 roughness / metallic / dissolve;
   defined in csrc/rtx_kernels.hip: k_debug_layer), every other layer < 30 opaque black. */
int  rtx_read_layer(rtx_ctx*, uint32_t layer, uint32_t width, uint32_t height, uint8_t* rgba8, size_t bytes);
int  rtx_get_stats(rtx_ctx*, rtx_stats* out);
/* t6 `g_EmissiveTriangles` as built by rtx_commit_scene (80 B records, Renderer.h:113-124) */
int  rtx_get_lights(rtx_ctx*, void* out80, uint32_t max_count, uint32_t* count_out);

/* multi-GPU (new; the reference is single-GPU): owned tiles of the accumulation buffer <-> compact
   [tiles_per_shard][tile_size^2] float4 slab for one RCCL (all)gather.  Pointers are DEVICE pointers.
   On the context's own stream both calls are synchronous; on a caller-bound stream (rtx_set_stream) they only enqueue, so that
   pack -> collective -> unpack runs stream-ordered without a host round trip (a 1/8-shard frame is ~2.7 ms). */
int  rtx_shard_slab_bytes(const rtx_params*, size_t* bytes_per_shard);
int  rtx_pack_tiles(rtx_ctx*, const rtx_params*, void* device_slab);
int  rtx_unpack_tiles(rtx_ctx*, const rtx_params*, const void* device_slabs_all_shards);

/* kernel-level entry points used by the parity tests (host arrays in/out; run the SAME device kernels
   the render loop uses).  rays8 = (ox,oy,oz,tmin, dx,dy,dz,tmax) per ray.
   hits4 = (t,u,v, global triangle id as uint bits; 0xFFFFFFFF = miss) — TraceRay closest hit, a11. */
int  rtx_debug_primary_rays(rtx_ctx*, const rtx_params*, uint32_t sample_id, float* rays8 /* W*H*8 */);
int  rtx_debug_trace_closest(rtx_ctx*, const float* rays8, uint32_t n, float* hits4);
int  rtx_debug_trace_any(rtx_ctx*, const float* rays8, uint32_t n, uint8_t* occluded);
/* closest-hit traversal of the BVH (never the tiny-scene path) that reports its work: stats4 = n * (t, node steps, triangle tests,
   prim bits) — tree-quality measurements for DESIGN.md, not part of the reference boundary */
int  rtx_debug_trace_stats(rtx_ctx*, const float* rays8, uint32_t n, float* stats4);
/* RTX_OPT_TRACE_COUNTERS: node steps / triangle tests of the closest-hit rays and of the any-hit rays traced by the persistent kernels since the last call (which resets them) */
int  rtx_debug_trace_counters(rtx_ctx*, uint64_t out4[4]);
/* downloads the resident wide BVH and checks it on the host: 0 = every triangle is in exactly one leaf slot and inside all the
   decoded boxes above it (what the GPU refit must preserve); > 0 = validator code; < 0 = RTX_ERR_* */
int  rtx_debug_validate_bvh(rtx_ctx*);
/* FNV-1a hashes of the wide tree as the device holds it: out[0] the node records, out[1] the leaf-ordered triangle records — two contexts hold the same tree iff both agree
   (the GPU build against its host twin, a loaded scene cache against the build it was saved from) */
int  rtx_debug_tree_hash(rtx_ctx*, uint64_t out2[2]);
/* the wide tree itself (tooling: diffing two trees that should be equal): which = 0 the device's records, 1 the host builder's mirror of them (RTX_ERR_STATE when the tree was
   built on the device).  nodes: rtx_stats.bvh_nodes x 80 bytes, tris: rtx_stats.bvh_refs x 48 bytes; either may be NULL */
int  rtx_debug_read_tree(rtx_ctx*, int which, void* nodes, uint64_t nodes_bytes, void* tris, uint64_t tris_bytes);
/* what the HOST builder keeps between commits for its refits (tooling): the binary tree (64-byte records) and the leaf order (uint32 per leaf reference).  In: the capacities
   of the two buffers, out: the sizes in bytes; a buffer that is too small (or NULL) is left alone */
int  rtx_debug_read_host_build(rtx_ctx*, void* nodes2, uint64_t* nodes2_bytes, void* leaf_order, uint64_t* leaf_order_bytes);
/* FNV-1a hashes of what the host side holds between commits (tooling: which call changed something it should not have): {mesh indices, mesh vertices, material ids +
   materials, instance records, leaf order, binary tree, wide tree mirror, shade records + object-space triangles + leaf slots} */
int  rtx_debug_host_checksums(rtx_ctx*, uint64_t out8[8]);
/* the last geometry-changing rtx_commit_scene: ms5 = {GPU build: boxes + keys, sort, PLOC rounds, top of the tree on the host, layout} (all 0 after a host build),
   counts4 = {wide nodes, leaf entries, PLOC rounds, clusters handed to the host} */
int  rtx_debug_build_info(rtx_ctx*, double ms5[5], uint32_t counts4[4]);
/* out16 per hit = pos3, matID bits, normal3, area, inst bits, flat3, pad4 (ClosestHit, Hit_v6.hlsl:12-61) */
int  rtx_debug_surface(rtx_ctx*, const float* rays8, const float* hits4, uint32_t n, float* out16);
/* in9 = n(3) wo(3) wi(3); out8 = f(3), pdf, p_d, p_s, 0, 0 */
int  rtx_debug_bsdf_eval(rtx_ctx*, uint32_t mat_id, uint32_t flags, const float* in9, uint32_t n, float* out8);
/* in8 = n(3) wo(3) seed(2 uint bits); out8 = wi(3), strategy bits, seed_out(2), 0, 0 */
int  rtx_debug_bsdf_sample(rtx_ctx*, uint32_t mat_id, uint32_t flags, const float* in8, uint32_t n, float* out8);
/* TEA RNG on the device: n draws from one seed (Common_v6.hlsl:119-138) */
int  rtx_debug_tea(rtx_ctx*, uint32_t seed[2], uint32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif
