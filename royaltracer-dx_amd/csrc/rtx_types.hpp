// rtx_types.hpp — device-resident scene layouts (HBM) shared by host build code and kernels.
#pragma once
#include <stdint.h>
#include "rtx_bsdf.hpp"

namespace rtx {

struct alignas(16) F4 { float x, y, z, w; };

// BVH2 node with BOTH children's boxes stored in the parent: the HOST-side build / refit form (rtx_scene_host.cpp); the kernels
// traverse the compressed 8-wide collapse below.
//   a = (c0.min.xyz, c0.max.x)   b = (c0.max.yz, c1.min.xy)   c = (c1.min.z, c1.max.xyz)
//   d = (child0 bits, child1 bits, -, -)
// child >= 0: index of an internal node.  child < 0: leaf, v = ~child, first triangle slot = v >> 3,
// triangle count = (v & 7) + 1.  Nodes are laid out breadth-first so that [0, K) is the top of the
// tree (the part staged in LDS).
struct alignas(16) NodeGPU { F4 a, b, c, d; };
constexpr int32_t kEmptyChild = 0x7FFFFFFF;   // never visited (box is inverted)

// Device traversal format: compressed 8-wide node, 80 bytes = five 16-B loads per traversal step (the traversal kernels are
// bound by the number of per-lane loads, profiles/r01_pmc_bvh.md).  After Ylitie, Karras, Laine, "Efficient Incoherent Ray
// Traversal on GPUs Through Compressed Wide BVHs" (HPG 2017), re-derived for this renderer's exactness contract:
//   child box plane = p + q * 2^e per axis, q an unsigned byte, rounded OUTWARD from the binary tree's padded boxes
//   e_imask   = (ex + 127) | (ey + 127) << 8 | (ez + 127) << 16 | imask << 24      imask bit s: slot s is an internal node
//   child_base: index of the first internal child; internal children are consecutive in slot order (rank = popcount)
//   tri_base  : first triangle of this node's leaf children, consecutive in slot order, <= 4 per child
//   trivalid  : nibble s = (1 << count_s) - 1 for a leaf child in slot s, 0 otherwise
//   q[12]     : qlo_x, qlo_y, qlo_z, qhi_x, qhi_y, qhi_z as two dwords each (byte k of the pair = slot k)
// Slots are assigned so that visiting hit children in increasing (slot ^ octant) — octant bit a = ray goes negative
// along axis a — approximates front-to-back order without sorting.  Breadth-first: children follow their parents.
constexpr uint32_t kStackEntryBytes = 6;      // LDS traversal stack: bytes per entry (one per tree level) and lane (rtx_traverse.hpp: StackLds)
struct alignas(16) Node8GPU { float px, py, pz; uint32_t e_imask; uint32_t child_base, tri_base, trivalid, pad; uint32_t q[12]; };
static_assert(sizeof(Node8GPU) == 80, "Node8GPU must be 80 bytes");

// World-space triangle (device copy: in the wide tree's order, rtx_scene_host.cpp): v0 (w = global triangle id bits), e1 = v1 - v0, e2 = v2 - v0.
struct alignas(16) TriGPU { F4 v0, e1, e2; };

// Tiny scenes (<= 64 triangles, e.g. the Cornell Box): no BVH.  Triangles are merged, where possible, into planar
// convex quads; each record is a CONSERVATIVE pre-test "ray hits the supporting plane inside the polygon":
//   t = (pl.w - pl.xyz . o) / (pl.xyz . d),  P = o + t d,  inside <=> m_i . P + c_i >= -delta for the 4 edge planes
// (unit in-plane edge normals, so one world-space distance tolerance delta serves all).  Records are evaluated for
// all rays in a wave-uniform loop, two records per iteration on packed-FP32 instructions, coefficients scalar-loaded.
// Device form: two records transposed into 20 float2 rows: 0-3 plane | 4-7 edge 0 | 8-11 edge 1 | 12-15 edge 2 | 16-19 edge 3.
// Record r owns triangles 2r and 2r+1 of `small_tris` (a missing partner is a zero-area triangle).
struct alignas(16) SmallRecPair { float r[20][2]; };
static_assert(sizeof(SmallRecPair) == 160, "SmallRecPair must be 160 bytes");
constexpr uint32_t kSmallSceneMaxTris = 64;

// What ClosestHit (Hit_v6.hlsl:12-61) needs about a triangle, pre-gathered per GLOBAL triangle id:
// object-space flat normal + area, and the three per-vertex normals with the "all(n != 0) else flat"
// substitution (Hit_v6.hlsl:40-46) already applied.  One 64-B record instead of ~12 dependent gathers.
struct alignas(16) TriShade {
    uint32_t mat, inst;
    float flat[3];
    float n0[3], n1[3], n2[3];
    float area;
    float guard_tau;   // tiny scenes: a hit whose smallest barycentric is below this may lie within the guard margin of a hull plane (rtx_scene_host.cpp); 0 = never
};
static_assert(sizeof(TriShade) == 64, "TriShade must be 64 bytes");

struct alignas(16) InstGPU { float o2w[16]; float nrm[16]; float o2w_inv[16]; float prev_o2w[16]; };   // a5: objectToWorld, objectToWorldNormal, objectToWorldInverse, prevObjectToWorld

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

struct CameraGPU { float viewI[16]; float projI[16]; float prev_view[16]; float prev_proj[16]; };   // a7 (+ prevView / prevProjection for the temporal pass)

}  // namespace rtx
