// rtx_scene_host.hpp — host-side scene state behind the C-ABI: what the reference's Renderer keeps in
// m_materials / m_materialIDs / m_VB / m_IB / m_instances / m_emissiveTriangles (Renderer.h:100-141),
// plus the BVH build that replaces the driver's BLAS/TLAS (Renderer.cpp:772-946).
#pragma once
#include <vector>
#include <string>
#include <stdint.h>
#include "rtx_types.hpp"

namespace rtx {

float half_round(float x);                               // binary16 round trip (MaterialOptimized)
void  mat4_inverse(const float* m16, float* out16);      // XMMatrixInverse stand-in
void  normal_matrix(const float* o2w16, float* out16);   // Renderer.cpp:2104-2116

struct MeshHost { std::vector<float> verts; std::vector<uint32_t> idx; uint32_t matid_base = 0; };
struct InstHost { uint32_t mesh; float o2w[16]; float nrm[16]; float o2w_inv[16]; float prev_o2w[16]; uint32_t tri_base; };

struct BuiltScene {
    std::vector<MatGPU>   mats;
    std::vector<NodeGPU>  nodes;      // binary tree (build / refit form, host only)
    std::vector<Node8GPU> nodes8;     // compressed 8-wide collapse of `nodes` (device traversal form)
    std::vector<uint32_t> tri_slots8; // leaf-order slot of each triangle in the wide tree's order
    std::vector<TriGPU>   tris8;      // `tris` permuted into that order (device copy)
    uint32_t stack8 = 0;              // traversal stack entries (sibling groups) the wide tree can need
    std::vector<uint32_t> level_start8; // breadth-first levels of nodes8: level l = [level_start8[l], level_start8[l+1]) (GPU refit sweeps them bottom-up)
    std::vector<F4>       objtris;    // object-space vertex positions, 3 per GLOBAL triangle id (input of the GPU refit)
    float bvh_pad = 0.0f;             // absolute padding of the leaf boxes used by the last host build
    std::vector<TriGPU>   tris;       // leaf order
    // tiny-scene path (only when the scene has <= kSmallSceneMaxTris triangles): pre-test records + their triangles
    std::vector<SmallRecPair> small_recs; std::vector<TriGPU> small_tris; uint32_t small_nrec = 0;
    std::vector<F4> small_poly;       // 4 corners per record (world space; a triangle repeats its last corner)
    uint32_t small_nocc = 0;          // records [0, small_nocc) can lie between two scene points; [small_nocc, small_nrec) are faces of the scene's convex hull
    float small_cm = 0.0f, small_delta = 0.0f;   // margin coefficient for t, distance tolerance of the edge planes
    float small_hull_margin = 0.0f;              // NEE origins must lie this far inside every hull-face plane to use the hull-face shortcut (TriShade::guard_tau)
    std::vector<TriShade> shade;      // global triangle id order
    std::vector<InstGPU>  insts;
    std::vector<LightGPU> lights;
    std::vector<float>    lights80;   // reference-layout LightTriangle records (20 floats each)
    std::vector<uint32_t> leaf_order;   // BVH leaf order (kept for refits)
    float total_weight = 0.0f;
    uint32_t max_depth = 0;
    uint32_t refit_count = 0;           // commits since the last full build that only refitted the boxes
};

struct SceneHost {
    std::vector<float> mats128;                 // count * 32 floats
    std::vector<MeshHost> meshes;
    std::vector<uint32_t> matids;               // global materialIDs[]
    std::vector<InstHost> insts;
    std::string err;
    bool topo_dirty = true;                     // meshes / instances added since the last build (a transform change alone refits)
    bool mats_dirty = true;                     // rtx_set_materials since the material table was last derived

    bool set_materials(const void* mats, uint32_t count);
    bool add_mesh(const void* verts28, uint32_t nverts, const uint32_t* idx, uint32_t nidx, const uint32_t* matids, uint32_t* out);
    bool add_instance(uint32_t mesh, const float* o2w, uint32_t* out);
    bool set_instance_transform(uint32_t inst, const float* o2w);
    bool build(BuiltScene& out);
    void build_materials(BuiltScene& out);      // mats128 -> MatGPU table (clears mats_dirty)
    // transform-only update of the records the GPU refit does not derive itself: instance matrices and the light list
    bool refresh_transforms(BuiltScene& out);
    void build_lights(BuiltScene& out) const;
    void fill_objtris(BuiltScene& out) const;   // object-space triangles for the GPU refit (rtx_scene_cache.cpp: not stored in a cache file)
};

// binary scene cache (rtx_scene_cache.cpp): the host scene + everything build() derived that the device needs; versioned, checksummed
// cam12 (optional): eye, center, up, fovY degrees, znear, zfar of the host layer's scene (rtxh_scene_save / rtxh_scene_load)
// aux (optional): data the host layer keeps beside the scene and this layer only carries — fixed-size records (MaterialExt, index-aligned with the material
// table) and a text blob (the texture file names, each NUL-terminated); both empty for a context-level save
struct CacheAux { uint32_t rec_bytes = 0; std::vector<uint8_t> records; std::vector<char> text; };
bool save_scene_cache(const SceneHost& H, const BuiltScene& B, const char* path, std::string& err, const float* cam12 = nullptr, const CacheAux* aux = nullptr);
bool load_scene_cache(const char* path, SceneHost& H, BuiltScene& B, std::string& err, float* cam12 = nullptr, CacheAux* aux = nullptr);

// binned-SAH BVH2 over world-space triangles (9 floats each); fills nodes (breadth-first, children boxes in
// parent) and the leaf-ordered triangle permutation.
void refit_bvh(const std::vector<float>& wtri, float pad_abs, std::vector<NodeGPU>& nodes, const std::vector<uint32_t>& leaf_order);
void build_bvh(const std::vector<float>& wtri, float pad_abs, std::vector<NodeGPU>& nodes,
               std::vector<uint32_t>& leaf_order, uint32_t& max_depth);
// collapse the binary tree into the compressed 8-wide device form (largest-area internal child opened first, octant-ordered
// slots, outward-rounded byte quantisation); tri_slots = leaf-order slots in the wide tree's triangle order; max_stack =
// bound on the sibling-group entries a traversal can hold (one per level).  Returns false on a malformed input tree.
bool collapse_bvh8(const std::vector<NodeGPU>& nodes2, std::vector<Node8GPU>& nodes8, std::vector<uint32_t>& tri_slots, uint32_t& max_stack,
                   std::vector<uint32_t>* level_start = nullptr);
// coverage check of a wide tree on its DECODED boxes (tests, rtx_debug_validate_bvh): 0 = every triangle order[tri_slots[i]] is in
// exactly one leaf slot and inside all boxes above it, children follow parents; otherwise a small positive code
int validate_bvh8(const std::vector<float>& world_tris9, const std::vector<Node8GPU>& nodes, const std::vector<uint32_t>& order,
                  const std::vector<uint32_t>& tri_slots, uint32_t* max_stack_seen);

}  // namespace rtx
