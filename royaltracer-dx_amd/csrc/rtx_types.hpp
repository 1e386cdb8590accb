// rtx_types.hpp — device-resident scene layouts (HBM) shared by host build code and kernels.
#pragma once
#include <stdint.h>
#include "rtx_bsdf.hpp"

namespace rtx {

struct alignas(16) F4 { float x, y, z, w; };

// BVH2 node with BOTH children's boxes stored in the parent (one 64-B fetch per traversal step).
//   a = (c0.min.xyz, c0.max.x)   b = (c0.max.yz, c1.min.xy)   c = (c1.min.z, c1.max.xyz)
//   d = (child0 bits, child1 bits, -, -)
// child >= 0: index of an internal node.  child < 0: leaf, v = ~child, first triangle slot = v >> 3,
// triangle count = (v & 7) + 1.  Nodes are laid out breadth-first so that [0, K) is the top of the
// tree (the part staged in LDS).
struct alignas(16) NodeGPU { F4 a, b, c, d; };
constexpr int32_t kEmptyChild = 0x7FFFFFFF;   // never visited (box is inverted)

// World-space triangle in leaf order: v0 (w = global triangle id bits), e1 = v1 - v0, e2 = v2 - v0.
struct alignas(16) TriGPU { F4 v0, e1, e2; };

// Tiny scenes (<= 64 triangles, e.g. the Cornell Box): a CONSERVATIVE plane-form pre-test per triangle, evaluated
// for all triangles in a wave-uniform loop (no BVH, no divergence); survivors go through the exact test.
//   t = (pl.w - pl.xyz . o) / (pl.xyz . d),  P = o + t d,  u = pu.xyz . P + pu.w,  v = pv.xyz . P + pv.w
// eps = barycentric tolerances (u, v, u+v) for a distance tolerance delta; eps.w = delta.
struct alignas(16) SmallTri { F4 pl, pu, pv, eps; };
constexpr uint32_t kSmallSceneMaxTris = 64;
// Device form: two consecutive triangles transposed into 16 float2 rows so that the pre-test of both runs on
// packed-FP32 instructions (v_pk_fma_f32) with wave-uniform (scalar-loaded) operands.  Rows:
//  0-3 pl.xyzw | 4-7 pu.xyzw | 8-11 pv.xyzw | 12 eps_u | 13 eps_v | 14 1+eps_w | 15 grazing threshold (-1 = padding)
struct alignas(16) SmallPair { float r[16][2]; };
static_assert(sizeof(SmallPair) == 128, "SmallPair must be 128 bytes");

// What ClosestHit (Hit_v6.hlsl:12-61) needs about a triangle, pre-gathered per GLOBAL triangle id:
// object-space flat normal + area, and the three per-vertex normals with the "all(n != 0) else flat"
// substitution (Hit_v6.hlsl:40-46) already applied.  One 64-B record instead of ~12 dependent gathers.
struct alignas(16) TriShade {
    uint32_t mat, inst;
    float flat[3];
    float n0[3], n1[3], n2[3];
    float area;
    float pad;
};
static_assert(sizeof(TriShade) == 64, "TriShade must be 64 bytes");

struct alignas(16) InstGPU { float o2w[16]; float nrm[16]; };   // a5: objectToWorld, objectToWorldNormal

// LightTriangle (Renderer.h:113-124) with the sample-independent part of SampleLightNEE_GI
// (Sampler_v6.hlsl:540-575) hoisted to the host: world-space vertices, light normal, clamped area pdf.
struct alignas(16) LightGPU {
    float xv[3]; float cdf;
    float yv[3]; float pdf_l;      // max(EPS, weight / max(area, EPS))
    float zv[3]; float pad0;
    float em[3]; float pad1;
    float nl[3]; float pad2;       // normalize(cross(y - x, z - x)), before the per-sample flip
};
static_assert(sizeof(LightGPU) == 80, "LightGPU must be 80 bytes");

struct CameraGPU { float viewI[16]; float projI[16]; };

}  // namespace rtx
