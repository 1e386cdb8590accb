// rtx_dev_common.hpp — device-side helpers shared by all kernels: block size, wave helpers, sub-queue compaction, path slot -> pixel, primary ray
// (included by rtx_kernels.hip only; see its header comment for the overall design)
#pragma once
#include "rtx_kernels.hpp"

namespace rtx {

constexpr int kBlock = 256;
constexpr uint32_t kMaxNee = 16;

// ---------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Stream compaction into a WORKGROUP-PRIVATE sub-queue: every lane of the wave must call this (convergent).
// The counter lives in LDS (one ds_add per wave); there are no global atomics anywhere in the render loop —
// a single global counter saturates at ~88 returning atomics/us on MI355X and was the first bottleneck found
// (profiles/r01_cornell_c2_v1.md).
__device__ __forceinline__ uint32_t block_push(bool pred, uint32_t* lds_counter) {
    const unsigned long long mask = __ballot(pred);
    const uint32_t cnt = (uint32_t)__popcll(mask);
    if (cnt == 0) return 0xFFFFFFFFu;                    // wave-uniform
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(lds_counter, cnt);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    return base + prefix;
}

// ---------------------------------------------------------------------------------------------
// pixel <-> local path-slot mapping (shard tiles, 8x8 pixel blocks inside a tile so that one wave
// covers a compact screen region)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slot_to_pixel(const DevFrame& f, uint32_t pl, uint32_t& x, uint32_t& y) {
    const uint32_t ts2 = 2u * f.tile_shift;                  // tile_size is a power of two
    const uint32_t k = pl >> ts2, r = pl & ((1u << ts2) - 1u);
    uint32_t tx, ty;
    if (!shard_tile(f, k, tx, ty)) return false;
    const uint32_t bshift = f.tile_shift - 3u;               // 8x8 pixel blocks per tile row = 2^bshift
    const uint32_t blk = r >> 6, ln = r & 63u;
    const uint32_t bx = blk & ((1u << bshift) - 1u), by = blk >> bshift;
    x = (tx << f.tile_shift) + bx * 8u + (ln & 7u);
    y = (ty << f.tile_shift) + by * 8u + (ln >> 3);
    return x < f.width && y < f.height;
}

// primary ray, RayGen_v6_pass1.hlsl:51-95
__device__ __forceinline__ void primary_ray(const CameraGPU& cam, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float jx, float jy, f3& o, f3& d) {
    const float dx = (((float)x + jx) / (float)W) * 2.0f - 1.0f;
    const float dy = (((float)y + jy) / (float)H) * 2.0f - 1.0f;
    const float* P = cam.projI; const float* Vi = cam.viewI;
    const float ndy = -dy;
    f3 tg = mk3(P[0] * dx + P[4] * ndy + P[8] + P[12], P[1] * dx + P[5] * ndy + P[9] + P[13], P[2] * dx + P[6] * ndy + P[10] + P[14]);
    d = normalize(xform_dir(Vi, tg));
    o = mk3(Vi[12], Vi[13], Vi[14]);
}

}  // namespace rtx
