/*
 * rt_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the RoyalTracer-DX path-tracing inner loop used only
 * as the checker for the HIP path (tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py).  Nothing under royaltracer-dx_amd/ may include,
 * link or call this file.
 *
 * PARITY UNPINNED (shader math): the reference holds no golden vectors, no tests
 * and no CPU path, and its GPU path is HLSL/DXR + DirectXMath which cannot be
 * built in this image.  What IS pinned against the reference:
 * OBJ/MTL parsing and glm::lookAt, via oracle/ref_probe.cpp, which compiles the
 * vendored tinyobjloader / glm headers where they lie under /root/reference.
 *
 * Conventions shared with the product's C-ABI (include/rtx.h):
 *   - 4x4 matrices are 16 floats, column-major storage of a column-vector matrix
 *     (element (row r, col c) at m[c*4+r]) — the byte layout the reference hands
 *     its shaders (Renderer.cpp:1722-1768, 2091-2121; HLSL reads it column-major).
 *   - Material 128 B, Vertex 28 B, LightTriangle 80 B as in Vertex.h:14-35,
 *     Renderer.h:113-124.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* identical layout to rtx_params (include/rtx.h) */
typedef struct orc_params {
    uint32_t width, height;      /* full image */
    uint32_t spp;                /* samples per pixel rendered by this call */
    uint32_t sample_base;        /* first sample id s (seed formula), reference uses 1 */
    uint32_t max_bounces;        /* path segments traced per sample */
    uint32_t nee_samples;        /* light samples per bounce (0 = BSDF sampling only) */
    uint32_t rr_start;           /* Russian roulette for bounce index > rr_start */
    uint32_t frame_seed;         /* replaces uint(time) in the seed formula */
    uint32_t flags;              /* ORC_FLAG_* */
    uint32_t tile_size;          /* shard tile edge in pixels (0 => 64) */
    uint32_t shard_rank;         /* this shard renders tiles t with t % shard_count == shard_rank */
    uint32_t shard_count;        /* 0 or 1 => whole image */
} orc_params;

#define ORC_FLAG_LAMBERT_ONLY 1u   /* force strategy 0, p_d = 1 */
#define ORC_FLAG_JITTER       2u   /* legacy sub-pixel jitter (RayGen.hlsl:84-87) */
#define ORC_FLAG_TRANSMISSION 4u   /* EXTENSION: strategy 3, rough dielectric transmission (the reference has a stub only: BRDF_v6.hlsl:44-47,85-87) */

orc_ctx* orc_create(void);
void     orc_destroy(orc_ctx*);
int  orc_set_materials(orc_ctx*, const void* mats128, uint32_t count);
int  orc_add_mesh(orc_ctx*, const void* verts28, uint32_t nverts, const uint32_t* indices,
                  uint32_t nidx, const uint32_t* material_ids, uint32_t* mesh_out);
int  orc_add_instance(orc_ctx*, uint32_t mesh, const float* o2w16, uint32_t* inst_out);
int  orc_set_instance_transform(orc_ctx*, uint32_t inst, const float* o2w16);   /* then orc_commit */
int  orc_commit(orc_ctx*);
int  orc_set_camera(orc_ctx*, const float* view16, const float* proj16);
int  orc_set_threads(orc_ctx*, int nthreads);     /* OpenMP threads for orc_render (0 = all) */
int  orc_render(orc_ctx*, const orc_params*, float* accum_rgba /* W*H*4, added into */,
                uint64_t ray_counts[3] /* primary, extension, shadow; may be NULL */);
void orc_srgb8(const float* accum_rgba, uint32_t npix, uint8_t* out_rgba8);
/* the v6 pass-1 estimator (RayGen_v6_pass1.hlsl:48-190): RIS direct light + SamplePathSimple; nee_samples plays
   nee_samples_DI and nee_samples, max_bounces plays `bounces` (reference: 4, 4, 3).  Buffers use MapPixelID order and
   hold orc_pass1_slots(w,h) records of 40 / 40 / 60 bytes (Reservoir_DI, Reservoir_GI, SampleData). */
int  orc_render_v6_pass1(orc_ctx*, const orc_params*, float* accum_rgba, void* res_di40, void* res_gi40, void* sample60, uint64_t ray_counts[3]);
/* one full ReSTIR frame of the reference: pass 1 + temporal reuse (RayGen_v6_pass2.hlsl:46-204) + spatial reuse and final
   shade (RayGen_v6_pass3.hlsl:46-441).  cur_* / last_* are the u2..u7 buffers (40/40/60-byte records, orc_pass1_slots
   of them, MapPixelID order); last_* carry the state between frames (zero them for the first frame).  orc_set_camera
   keeps the previous view / projection like UpdateCameraBuffer does. */
int  orc_restir_frame(orc_ctx*, const orc_params*, float* accum_rgba, void* cur_di, void* cur_gi, void* cur_sd,
                      void* last_di, void* last_gi, void* last_sd, uint64_t ray_counts[3]);
uint32_t orc_map_pixel_id(uint32_t width, uint32_t x, uint32_t y);
size_t orc_pass1_slots(uint32_t width, uint32_t height);

/* unit-level entry points (golden vectors / GPU parity of the individual kernels) */
void orc_tea(uint32_t seed[2], uint32_t n, float* out);
void orc_seed_init(uint32_t x, uint32_t y, uint32_t s, uint32_t frame_seed, uint32_t out[2]);
void orc_sincos(float x, float* s, float* c);
void orc_rsqrt(const float* x, uint32_t n, float* out);   /* the deterministic rsqrt behind normalize() */
float orc_pow(float x, float y);
float orc_half_round(float x);
void orc_mat4_inverse(const float* m16, float* out16);
int  orc_primary_rays(orc_ctx*, const orc_params*, uint32_t s, float* rays8 /* W*H*8: o,tmin,d,tmax */);
/* rays8: n * (ox,oy,oz,tmin,dx,dy,dz,tmax); hits4: n * (t,u,v, prim-as-uint-bits); prim = 0xFFFFFFFF on miss.
   mode 0 = brute force over all triangles, 1 = the oracle's own BVH */
int  orc_trace_closest(orc_ctx*, const float* rays8, uint32_t n, int mode, float* hits4);
int  orc_trace_any(orc_ctx*, const float* rays8, uint32_t n, int mode, uint8_t* occluded);
/* surface record for hits: out16 per hit = pos3, matID(bits), normal3, area, inst(bits), flat3, pad4 */
int  orc_surface(orc_ctx*, const float* rays8, const float* hits4, uint32_t n, float* out16);
uint32_t orc_num_triangles(orc_ctx*);
uint32_t orc_num_lights(orc_ctx*);
int  orc_get_lights(orc_ctx*, void* out80, uint32_t max_count);
/* BSDF leaf math. in: n(3) wo(3) wi(3) per item, material id, flags; out8: f(3), pdf, p_d, p_s, pad2 */
int  orc_bsdf_eval(orc_ctx*, uint32_t mat_id, uint32_t flags, const float* n_wo_wi9, uint32_t n, float* out8);
/* in: n(3) wo(3) seed(2 as uint bits) per item; out8: wi(3), strategy, seed_out(2), pad2 */
int  orc_bsdf_sample(orc_ctx*, uint32_t mat_id, uint32_t flags, const float* n_wo_seed8, uint32_t n, float* out8);

#ifdef __cplusplus
}
#endif
#endif
