// rtx_build.hpp — BVH build ON THE GPU (RTX_OPT_GPU_BUILD; rtx_build.hip).  The reference gets its acceleration structures from the device in milliseconds
// (nv_helpers_dx12/BottomLevelASGenerator.cpp:178-247, TopLevelASGenerator.cpp:149-250, Renderer.cpp:893-946); this is that capability for geometry changes here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "rtx_types.hpp"
#include "rtx_scene_host.hpp"

namespace rtx {

struct GpuBuildResult {
    uint32_t nnodes8 = 0, ntris8 = 0, stack8 = 0;
    std::vector<uint32_t> level_start8;            // breadth-first levels of the wide tree (the refit sweeps them bottom-up)
    uint32_t ploc_iterations = 0, clusters_top = 0;
    float scale = 1.0f;                            // max(1, largest |world coordinate|): the host build's `scale` (box padding 2e-6 x scale)
    double ms_prims = 0, ms_sort = 0, ms_ploc = 0, ms_top_host = 0, ms_layout = 0;      // wall time of the phases (host clock around stream synchronises)
};

// Builds the TOPOLOGY of the compressed 8-wide tree over triangles [0, ntri): internal masks, child / triangle bases, triangle-valid nibbles, the leaf-slot order of the
// triangles (tris_out[s].v0.w = global id) and the level table.  Boxes are not its business: the caller runs the refit kernels (launch_refit, a full refit) afterwards, which
// derive the world triangles and quantise every node bottom-up exactly as they do after a transform-only commit.
class GpuBvhBuilder {
public:
    GpuBvhBuilder();
    ~GpuBvhBuilder();
    GpuBvhBuilder(const GpuBvhBuilder&) = delete; GpuBvhBuilder& operator=(const GpuBvhBuilder&) = delete;
    // "" on success, else what failed.  Synchronises `st` several times (it reads cluster and level counts back).
    std::string build(hipStream_t st, const F4* d_objtris, const TriShade* d_shade, const InstGPU* d_insts, uint32_t ntri, const BvhBuildOptions& opt, TriGPU* d_tris_out, GpuBuildResult& R);
    const Node8GPU* nodes() const;                 // device: R.nnodes8 records of the last build (topology only), valid until the next build
    void release();                                // frees the scratch memory (~130 B per triangle)
private:
    struct Impl; Impl* m;
};

// ---- the flatten of a geometry-changing commit ON THE DEVICE (RTX_OPT_GPU_BUILD): object-space triangles and shade records (Hit_v6.hlsl:12-61) of every instanced triangle from
//      the meshes as they were handed over.  The host twin is SceneHost::build's flatten loop (csrc/rtx_scene_host.cpp) — same functions of rtx_math.hpp, same order. ----
struct FlatInst { uint32_t tri_base, ntri, vert_base, idx_base, matid_base, pad_[3]; };       // one per instance, in instance order (tri_base ascending)
void launch_flatten(hipStream_t st, const float* verts7, const uint32_t* idx, const uint32_t* matids, uint32_t nmatids, const FlatInst* insts, uint32_t ninst, uint32_t ntri,
                    F4* objtris_out, TriShade* shade_out);

}  // namespace rtx
