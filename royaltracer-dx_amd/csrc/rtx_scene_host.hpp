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

// Knobs of the BVH builder (defaults = what the product builds; tools/bvh_lab.cpp and the A/B tools change them by key).
struct BvhBuildOptions {
    int      bins = 16;           // binned SAH: bins per axis for nodes above `sweep_below`
    uint32_t sweep_below = 0;     // nodes with at most this many references use the full-sweep SAH (every centroid position) instead of bins
    uint32_t leaf_stop = 1;       // nodes with at most this many references are not split further (the wide collapse merges small subtrees into leaf slots anyway).  1 since round 5:
                                  // chosen on the HARD stand-ins (profiles/r05_bvh_lab.md: closest-hit cost -2.1 % / -2.8 %, any-hit -1.1 % / -0.7 %; nothing on the uniform ones)
    double   split_alpha = 0.0;   // spatial splits where the object split's two sides overlap by more than this fraction of the scene's surface area (0 = never)
    double   split_budget = 0.3;  // ... and at most this many extra references, as a fraction of the triangle count
    int      reinsert_passes = 2; // passes of the insertion-based topology optimisation
    double   reinsert_frac = 1.0; // share of the nodes (largest boxes first) a pass tries to re-insert ...
    uint32_t reinsert_cap = 200000; // ... and at most this many of them
    int      slot_assign = 0;     // collapse_bvh8: children to octant slots greedily (0) or by the exact maximum of the summed diagonal projections (1)
    double   tri_cost = 0.7;      // collapse_bvh8: cost of a triangle test relative to a node step
    int      threads = 0;         // build_bvh: threads of the top-down phase (0 = up to 16 of the machine's; 1 = serial).  The tree does not depend on it
    int      ploc_radius = 0;     // > 0: the bottom-up PLOC builder with this search radius below the top-down SAH builder (the host twin of the GPU build, csrc/rtx_build.hip)
    uint32_t ploc_top = 16384;    // ... which stops at this many clusters; the SAH builder (+ re-insertion) then builds the top of the tree over them (1: PLOC to the root)
};
BvhBuildOptions& bvh_build_options();                      // process-wide defaults: what a new SceneHost starts with (RTX_BVH="key=value,..." in the environment edits them once)
bool bvh_build_option(const char* key, double value);      // edits the defaults; false: unknown key
bool bvh_build_option(BvhBuildOptions& o, const char* key, double value);

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
    std::vector<uint32_t> leaf_order;   // BVH leaf order (kept for refits): the triangle behind every leaf REFERENCE — a permutation unless spatial splits duplicated some
    uint32_t any_order = 0;             // visiting order of any-hit rays chosen by probe_anyhit_order (0 slot order, 1 nearest octant first, 2 farthest first)
    uint32_t built_tris = 0;            // triangle count the topology was built for
    float total_weight = 0.0f;
    uint32_t max_depth = 0;
    uint32_t refit_count = 0;           // commits since the last full build that only refitted the boxes
    std::vector<uint32_t> inst_moved;   // refresh_transforms: 1 = the instance's objectToWorld differs from the last commit's (the GPU refit touches the triangles and nodes of these only)
};

struct SceneHost {
    std::vector<float> mats128;                 // count * 32 floats
    std::vector<MeshHost> meshes;
    std::vector<uint32_t> matids;               // global materialIDs[]
    std::vector<InstHost> insts;
    std::string err;
    bool topo_dirty = true;                     // meshes / instances added since the last build (a transform change alone refits)
    bool mats_dirty = true;                     // rtx_set_materials since the material table was last derived
    BvhBuildOptions bvh = bvh_build_options();  // builder knobs of this scene (rtx_set_option RTX_OPT_BVH_*)

    bool set_materials(const void* mats, uint32_t count);
    bool add_mesh(const void* verts28, uint32_t nverts, const uint32_t* idx, uint32_t nidx, const uint32_t* matids, uint32_t* out);
    bool add_instance(uint32_t mesh, const float* o2w, uint32_t* out);
    bool set_instance_transform(uint32_t inst, const float* o2w);
    bool build(BuiltScene& out, bool host_bvh = true);      // host_bvh = false (RTX_OPT_GPU_BUILD): everything but the tree — materials, shade records, object-space triangles, lights
    void build_materials(BuiltScene& out);      // mats128 -> MatGPU table (clears mats_dirty)
    // transform-only update of the records the GPU refit does not derive itself: instance matrices and the light list
    bool refresh_transforms(BuiltScene& out);
    void build_lights(BuiltScene& out) const;
    void refresh_lights(BuiltScene& out) const; // the world-space half of the light records from lights80 (transform-only commits)
    // RTX_OPT_GPU_BUILD: everything of a geometry-changing commit EXCEPT the per-triangle work (flatten, shade records, tree), which the device does from the meshes themselves
    // (csrc/rtx_build.hip: k_flatten): materials, instance records and triangle ranges, lights.  out.shade / objtris / trees are left empty, out.built_tris = the triangle count
    bool prepare_device_build(BuiltScene& out);
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
               std::vector<uint32_t>& leaf_order, uint32_t& max_depth, const BvhBuildOptions& opt = bvh_build_options());
// the top of a PLOC tree over m cluster boxes (mn.xyz, mx.xyz each): top-down SAH + re-insertion, root first; child >= 0: node index, < 0: ~cluster (host twin and GPU build share it)
struct ClusterTopNode { float mn[3], mx[3]; int32_t left, right; };
void build_cluster_top(const float* boxes6, uint32_t m, const BvhBuildOptions& opt, std::vector<ClusterTopNode>& out);
// collapse the binary tree into the compressed 8-wide device form (largest-area internal child opened first, octant-ordered
// slots, outward-rounded byte quantisation); tri_slots = leaf-order slots in the wide tree's triangle order; max_stack =
// bound on the sibling-group entries a traversal can hold (one per level).  Returns false on a malformed input tree.
bool collapse_bvh8(const std::vector<NodeGPU>& nodes2, std::vector<Node8GPU>& nodes8, std::vector<uint32_t>& tri_slots, uint32_t& max_stack,
                   std::vector<uint32_t>* level_start = nullptr, const BvhBuildOptions& opt = bvh_build_options());
// coverage check of a wide tree on its DECODED boxes (tests, rtx_debug_validate_bvh): 0 = children follow parents, every leaf slot entry order[tri_slots[i]] is a
// triangle, and every triangle is COVERED: referenced once and inside all boxes above that reference, or — a triangle a spatial split handed to several leaves —
// each of a fixed set of 28 points on it (corners, edge thirds, interior lattice) lies inside all boxes above one of its references; otherwise a small positive code
// host-side replay of the device traversal on B.nodes8 / B.tris8 (counts for tools/bvh_lab.cpp and the any-hit probe; rtx_scene_host.cpp)
struct ReplayHit { float t; uint32_t slot, prim; uint32_t steps, tris; };     // prim = global triangle id or 0xffffffff; steps = node steps, tris = triangle tests
ReplayHit replay_trace(const BuiltScene& B, const float o[3], const float d[3], float tmin, float tmax, bool any, uint32_t any_order = 0, float t_known = -1.0f,
                       std::vector<uint8_t>* seq = nullptr);      // seq: per node step, the number of triangles it queued (tools/bvh_lab: wave-schedule simulation)
bool replay_tri_test(const float o[3], const float d[3], const TriGPU& Tg, float tmin, float tmax, float& t);     // the replay's triangle test alone (tools/soup_lab.cpp: brute force in the same arithmetic)
uint32_t probe_anyhit_order(const BuiltScene& B);

struct CoverCheck {                                         // coverage bookkeeping of the tree validators (rtx_scene_host.cpp)
    struct Part { uint32_t tri; double b[6]; };
    const std::vector<float>& w; std::vector<uint32_t> refs; std::vector<Part> boxes;
    explicit CoverCheck(const std::vector<float>& world_tris9);
    void count(uint32_t tri);                                // first pass: one call per reference
    int add(uint32_t tri, const double mn[3], const double mx[3]);   // second pass: the box chain above a reference (intersection of all boxes above it)
    int finish();
};
int validate_bvh8(const std::vector<float>& world_tris9, const std::vector<Node8GPU>& nodes, const std::vector<uint32_t>& order,
                  const std::vector<uint32_t>& tri_slots, uint32_t* max_stack_seen);

}  // namespace rtx
