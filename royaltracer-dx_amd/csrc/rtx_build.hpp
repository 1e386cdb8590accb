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

}  // namespace rtx
