/*
 * rtx_host.h — C entry points of the HOST layer that sits above include/rtx.h: the reference's scene-loader /
 * material / camera surface (ObjLoader::loadObjFile, Material, Vertex, Manipulator, Renderer; see
 * the headers under royaltracer-dx_amd/host/ for the C++ classes with the reference's own names) and the synthetic scenes
 * BASELINE.json names.  These exist so that Python (tests, bench.py) and other FFI callers can build the same
 * scenes the C++ facade builds; none of them touches the GPU except rtxh_scene_upload.
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H
#include "rtx.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtxh_scene rtxh_scene;

rtxh_scene* rtxh_scene_cornell(void);                                      /* SURVEY §8d Cornell Box, 32 triangles */
rtxh_scene* rtxh_scene_sponza_class(uint32_t target_tris, uint32_t seed);  /* C3/C4 */
rtxh_scene* rtxh_scene_bistro_class(uint32_t target_tris, uint32_t seed);  /* C5 */
/* the same two scenes with the triangle-size distribution of the real assets (walls of a few large triangles beside millimetre ornament, long thin trims, overlapping
   cloth, foliage; host/Scenes.h): what BVH-builder quality is measured on */
rtxh_scene* rtxh_scene_sponza_class_hard(uint32_t target_tris, uint32_t seed);
rtxh_scene* rtxh_scene_bistro_class_hard(uint32_t target_tris, uint32_t seed);
/* files: nfiles OBJ paths, each loaded through ObjLoader::loadObjFile (ObjLoader.h:393-495) as the reference's
   Renderer does (Renderer.cpp:363-407); returns NULL on parse error (message via rtxh_last_error) */
rtxh_scene* rtxh_scene_from_obj(const char* const* files, uint32_t nfiles, const char* mtl_dir);
/* SURVEY 8(f3) binary scene cache at the host level: the scene (materials, meshes, instances, camera) AND everything rtx_commit_scene
   derives from it (BVH, shading records, LUTs, light CDF); versioned + checksummed (csrc/rtx_scene_cache.cpp).  Saving needs no GPU.
   A loaded scene answers the accessors below like any other and rtxh_scene_upload hands the file's prebuilt arrays to the context
   (rtx_load_scene_cache) instead of rebuilding.  NULL / RTX_ERR_INVALID + rtxh_last_error on a missing, corrupt or foreign file. */
int         rtxh_scene_save(const rtxh_scene*, const char* path);
rtxh_scene* rtxh_scene_load(const char* path);
void        rtxh_scene_free(rtxh_scene*);
const char* rtxh_last_error(void);

uint32_t    rtxh_scene_num_materials(const rtxh_scene*);
const void* rtxh_scene_materials(const rtxh_scene*);                       /* 128-byte Material records */
/* SURVEY 8(f3) "full MTL PBR extension coverage": what tinyobj's material_t holds beyond the 128-byte Material (ObjLoader.h:428-435 copies
   Kd, d, Ks, Ke, Pr, Pm, Ps, Pc only; Vertex.h:21 "ADD MAP IDs LATER"): Ni, Ns, illum, Ka, Tf / Kt, Pcr, aniso, anisor and one texture id per map
   statement (-1 = none; ids index rtxh_scene_texture).  Only scenes loaded from OBJ / MTL files carry it (index-aligned with the
   material table, default materials included); no kernel reads it — the reference's shaders sample no textures. */
#define RTXH_NUM_MAP_SLOTS 13
enum { RTXH_MAP_KA = 0, RTXH_MAP_KD, RTXH_MAP_KS, RTXH_MAP_KE, RTXH_MAP_NS, RTXH_MAP_BUMP, RTXH_MAP_D, RTXH_MAP_DISP, RTXH_MAP_REFL, RTXH_MAP_PR, RTXH_MAP_PM, RTXH_MAP_PS, RTXH_MAP_NORM };
typedef struct rtxh_material_ext { float Ni, Ns, Pcr, aniso, anisor; int32_t illum; float Ka[3], Tf[3]; int32_t map[RTXH_NUM_MAP_SLOTS]; } rtxh_material_ext;
int         rtxh_scene_material_ext(const rtxh_scene*, uint32_t material, rtxh_material_ext* out);   /* RTX_ERR_INVALID: no such record */
uint32_t    rtxh_scene_num_textures(const rtxh_scene*);
const char* rtxh_scene_texture(const rtxh_scene*, uint32_t i);                                      /* file name as written in the .mtl */
uint32_t    rtxh_scene_num_meshes(const rtxh_scene*);
int         rtxh_scene_mesh(const rtxh_scene*, uint32_t i, const void** verts28, uint32_t* nverts,
                            const uint32_t** indices, uint32_t* nidx, const uint32_t** material_ids);
uint32_t    rtxh_scene_num_instances(const rtxh_scene*);
int         rtxh_scene_instance(const rtxh_scene*, uint32_t i, uint32_t* mesh, float o2w[16]);
uint64_t    rtxh_scene_num_triangles(const rtxh_scene*);
/* eye[3], center[3], up[3], fovY in degrees, znear, zfar (Renderer.cpp:46-48, 1730-1731) */
int         rtxh_scene_camera(const rtxh_scene*, float eye[3], float center[3], float up[3], float* fovy_deg, float* zn, float* zf);
int         rtxh_scene_set_camera(rtxh_scene*, const float eye[3], const float center[3], const float up[3]);   /* CameraManip.setLookat of the scene (Renderer.cpp:46-48); an OBJ file carries no camera */
/* view (Manipulator::setLookat -> getMatrix) and projection (XMMatrixPerspectiveFovRH) for an aspect ratio */
int         rtxh_scene_view_proj(const rtxh_scene*, float aspect, float view[16], float proj[16]);
/* rtx_set_materials / rtx_add_mesh / rtx_add_instance / rtx_commit_scene / rtx_set_camera */
int         rtxh_scene_upload(const rtxh_scene*, rtx_ctx*, float aspect);

/* leaf helpers of the host layer */
void rtxh_lookat(const float eye[3], const float center[3], const float up[3], float view16[16]);      /* manipulator.cpp:305-314 */
void rtxh_perspective_fov_rh(float fovy_rad, float aspect, float zn, float zf, float proj16[16]);      /* Renderer.cpp:1730-1731 */
void rtxh_generate_ess_lut(float roughness, float lut16[16]);                                          /* ObjLoader.h:351-387 */
void rtxh_mat4_inverse(const float m16[16], float out16[16]);
float rtxh_half_round(float x);
/* BVH builder invariants for tests: returns 0 when every triangle is in exactly one leaf and every child box
   contains its subtree; fills nodes / depth / max leaf size */
int  rtxh_bvh_check(const float* world_tris9, uint32_t ntris, uint32_t* nodes_out, uint32_t* depth_out, uint32_t* max_leaf_out);
/* the compressed 8-wide collapse of that tree (the device traversal form): same coverage invariants on the DECODED byte-grid
   boxes of the wide nodes, children stored after their parents, <= 4 triangles per leaf slot, and *stack_out = the exact
   number of sibling-group entries the deepest root-to-leaf path can push */
int  rtxh_bvh8_check(const float* world_tris9, uint32_t ntris, uint32_t* nodes8_out, uint32_t* stack_out);
/* shape of that tree: hist[0..4] = leaf slots with 0 (unused) / 1 / 2 / 3 / 4 triangles, hist[5] = internal child slots */
int  rtxh_bvh8_stats(const float* world_tris9, uint32_t ntris, uint32_t hist[6], uint32_t* nodes8_out);
/* builder knobs of the checks above and of every context created afterwards (process-wide defaults; csrc/rtx_scene_host.hpp BvhBuildOptions): "bins", "sweep",
   "leaf_stop", "split" (spatial splits, overlap threshold as a fraction of the scene's surface area), "split_budget", "reinsert" (passes), "reinsert_frac", "slot_assign",
   "tri_cost".  Returns 0, or RTX_ERR_INVALID for an unknown key */
int  rtxh_bvh_option(const char* key, double value);
/* the device traversal REPLAYED on the host over the wide tree the current options build for these triangles (node steps and triangle tests per ray: tools/bvh_lab.cpp
   judges builder changes by them; tests compare its hits with brute force): out4 = (t, node steps, triangle tests, global triangle id bits / 0xffffffff) per ray
   (origin, tmin, direction, tmax); any != 0: first hit found in visiting order any_order (RTX_OPT_ANYHIT_ORDER).  *refs_out = leaf entries (> ntris after spatial splits) */
int  rtxh_bvh_replay(const float* world_tris9, uint32_t ntris, const float* rays8, uint32_t nrays, int any, uint32_t any_order, float* out4, uint32_t* refs_out);
/* same invariants after building on `before` and REFITTING (topology kept) to `after` (TLAS refit, Renderer.cpp:594) */
int  rtxh_bvh_refit_check(const float* before_tris9, const float* after_tris9, uint32_t ntris);

/* the tiny-scene pre-test records rtx_commit_scene would build for this scene: per record 20 floats (plane xyz d,
   4 edge planes xyz c) and the global ids of its 1-2 triangles (-1 = none); for host-side conservativeness tests */
int  rtxh_scene_small_records(const rtxh_scene*, float* recs20, int32_t* tri_ids2, uint32_t max_recs, uint32_t* nrec_out, float* delta_out, float* cm_out);
/* of those records, [0, *nocc_out) can lie between two scene points; the rest are faces of the scene's convex hull, which NEE
   shadow segments skip */
int  rtxh_scene_small_occluders(const rtxh_scene*, uint32_t* nocc_out);
/* the visiting order of any-hit rays rtx_commit_scene's probe picks for this scene (RTX_OPT_ANYHIT_ORDER -1): 0 slot order, 1 nearest octant first, 2 farthest first; no GPU needed */
int  rtxh_scene_anyhit_order(const rtxh_scene*, uint32_t* order_out);

/* the headless Renderer facade (Renderer.h:46-51) for FFI callers */
typedef struct rtxh_renderer rtxh_renderer;
rtxh_renderer* rtxh_renderer_create(uint32_t width, uint32_t height, const char* name, int device);
int  rtxh_renderer_set_scene(rtxh_renderer*, const rtxh_scene*);
rtx_params* rtxh_renderer_params(rtxh_renderer*);
/* what OnRender issues: 0 (default) = rtx_render with params(), 1 = the reference's shipping frame — its three DispatchRays (Renderer.cpp:646-673), rtx_render_restir with
   restir_params() (one frame per OnRender; nee_samples 4, max_bounces 3 as in Common_v6.hlsl:8-12) */
int  rtxh_renderer_set_mode(rtxh_renderer*, int mode);
rtx_params* rtxh_renderer_restir_params(rtxh_renderer*);
rtx_ctx* rtxh_renderer_context(rtxh_renderer*);                           /* the context behind the facade (options, statistics); NULL before on_init */
int  rtxh_renderer_on_init(rtxh_renderer*);
int  rtxh_renderer_on_update(rtxh_renderer*);
/* Renderer::SetInstanceTransform: m_instances[i].second of the reference's OnUpdate (Renderer.cpp:444-449); the next on_update hands the matrix to the context and refits the
   resident tree on the GPU (transform-only rtx_commit_scene; the reference refits its TLAS every frame, Renderer.cpp:594) */
int  rtxh_renderer_set_instance_transform(rtxh_renderer*, uint32_t instance, const float o2w[16]);
int  rtxh_renderer_on_render(rtxh_renderer*);
int  rtxh_renderer_read_accum(rtxh_renderer*, float* rgba32f, size_t bytes);
int  rtxh_renderer_read_output(rtxh_renderer*, uint8_t* rgba8, size_t bytes);
int  rtxh_renderer_on_key_up(rtxh_renderer*, uint8_t key);          /* 'C': next entry of m_displayLevels (Renderer.cpp:748-754); read_output returns that layer */
uint32_t rtxh_renderer_display_layer(const rtxh_renderer*);
void rtxh_renderer_destroy(rtxh_renderer*);

/* image writers for the headless display path (the reference presents gOutput through a swap chain, Renderer.cpp:554-735, and
   writes no files): 8-bit RGBA PNG (stored deflate), binary PPM, and OpenEXR (uncompressed scanlines, FLOAT B/G/R =
   accumulation xyz / max(w, 1)).  0 on success. */
int  rtxh_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
int  rtxh_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
int  rtxh_write_exr(const char* path, const float* rgba32f_accum, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif
