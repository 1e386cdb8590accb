// rtx_math.hpp — scalar float math shared by the HIP kernels and the host-side scene code.
//
// Everything here is evaluated in IEEE-754 binary32 with a FIXED operation order: the library is
// compiled with -ffp-contract=off and without fast-math, division and sqrt are the correctly rounded
// forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  sin/cos/pow are written out with
// + - * / only (the HLSL intrinsics they replace are implementation-defined to a few ULP), so the
// result of every function is a pure function of its input bits on any IEEE machine.
//
// Reference constants: Pathtracer/include/Common_v6.hlsl:1-3.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RTX_HD __host__ __device__ __forceinline__
#else
#define RTX_HD inline
#endif

namespace rtx {

constexpr float kPI      = 3.1415f;      // Common_v6.hlsl:1 (sic)
constexpr float kInvPI   = 1.0f / kPI;   // x / PI is evaluated as x * (1 / PI), the reciprocal rounded once (rt_oracle.c: INV_PI_REF)
constexpr float kSBias   = 0.00002f;     // Common_v6.hlsl:2
constexpr float kEps     = 0.000001f;    // Common_v6.hlsl:3
constexpr float kTwoPi   = 6.28318548202514648f;  // float(2.0 * 3.14159265358979323846), Lambertian_v6.hlsl:10
constexpr float kTMinCam = 0.0001f;      // RayGen_v6_pass1.hlsl:94
constexpr float kTMax    = 10000.0f;     // RayGen_v6_pass1.hlsl:95
constexpr uint32_t kMissPrim = 0xFFFFFFFFu;
constexpr uint32_t kMissMat  = 0xFFFFFFFEu;  // Miss_v6.hlsl:6

struct f3 { float x, y, z; };

RTX_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RTX_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// dot, cross and the matrix-vector products are FUSED (one rounding per a*b + c, fixed nesting; oracle/rt_oracle.c does the same with
// fmaf): HLSL lets mul/add chains become mad/FMA, so this is as much "the reference result" as the unfused form, and it is what keeps
// the VALU-bound kernels short (tri_test 41 -> 27 instructions).  Everything else stays unfused (-ffp-contract=off).
RTX_HD float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
RTX_HD f3 cross(f3 a, f3 b) { return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))); }
// fused linear combinations (same rule as dot / cross; rt_oracle.c: madd3, lincomb3)
RTX_HD f3 madd3(f3 a, float s, f3 b) { return mk3(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)); }
RTX_HD f3 lincomb3(f3 x, float a, f3 y, float b, f3 z, float c) {
    return mk3(__builtin_fmaf(z.x, c, __builtin_fmaf(y.x, b, x.x * a)), __builtin_fmaf(z.y, c, __builtin_fmaf(y.y, b, x.y * a)), __builtin_fmaf(z.z, c, __builtin_fmaf(y.z, b, x.z * a)));
}
RTX_HD float length(f3 a) { return sqrtf(dot(a, a)); }
// THE HIT DEFINITION'S GUARD AGAINST 0 / 0 (a11; oracle/rt_oracle.c: tri_det_floor).  Moeller-Trumbore divides by det = e1 . (d x e2) = -d . (e1 x e2).  For a ray in (or
// within rounding of) the triangle's plane the exact value is 0 and the float value is rounding noise — at most ~7 ulp(|e1| |e2|) = 2^-21 |e1| |e2| for a unit direction
// (one rounding per product of the cross product, three of the fused dot) — and the quotients are then 0 / 0: u = v = -0 is accepted and t is arbitrary, a "hit" outside
// every bounding volume, which a culling structure reports or not depending on the boxes it happens to visit.  That hole was known since round 2
// (test_wide_bvh_equals_brute_force_on_hostile_soups counted and excluded such rays) and it is what made round 4's one-in-20 000 mismatch non-reproducible: the four waves of
// a workgroup of the persistent kernels draw their rays from one LDS cursor, so which rays share a wave — hence WHEN the speculative schedule lets a lane's pending triangles
// shorten its ray, hence which boxes it still visits — depends on timing.  So a triangle is hit only if |det| > kDetRel |e1| |e2| (32 x that noise bound): a ray closer
// than 1.5e-5 rad / sin(corner angle) to the triangle's plane passes it — 2e-10 of cosine-distributed directions; a sliver with a corner angle below 1.5e-5 rad is never hit.
// The floor is a per-triangle constant kept in TriGPU::e1.w, so the test costs what `det != 0` cost.  Measured on the host replay of the device traversal against brute
// force (tools/soup_lab.cpp, 600 000 hostile rays per soup: aimed at vertices / edges, lying in triangle planes, axis-parallel): needles (aspect 750) 7 closest-hit and
// 6 any-hit mismatches without the floor, 0 / 0 with it; slivers of aspect 10 .. 1e5: 8 / 7 -> 0 / 0 (at 2^-18: 0 / 1 — a legitimate but inaccurate t next to tmax).
// What the floor does NOT bound is the error of an accepted t for slivers thinner than that test's (relative error ~ 2^-23 / sin(corner angle)); the slab margins carry it.
#ifndef RTX_DET_REL                 // (tooling: `make VARIANT=nofloor VARFLAGS=-DRTX_DET_REL=0.0f` builds the definition of rounds 1-4, det != 0, for tools/order_fuzz.py's A/B)
#define RTX_DET_REL 1.52587890625e-05f
#endif
constexpr float kDetRel = RTX_DET_REL;      // 2^-16
RTX_HD float tri_det_floor(f3 e1, f3 e2) { return kDetRel * (sqrtf(dot(e1, e1)) * sqrtf(dot(e2, e2))); }
RTX_HD uint32_t f2u(float f);
RTX_HD float u2f(uint32_t u);
// rsqrt(x), x > 0 normal.  HLSL's rsqrt is a 1-ULP implementation-defined approximation; this is a deterministic one of the same quality
// (integer seed + three Newton steps in fma arithmetic, < 1 ulp) that the oracle executes operation for operation (rt_oracle.c:rsqrt_det):
// 13 instructions instead of the 33 issue slots of an IEEE sqrt followed by an IEEE divide, six normalizations per path vertex.
RTX_HD float rsqrt_det(float x) {
    float y = u2f(0x5f375a86u - (f2u(x) >> 1));
    const float h = 0.5f * x;
    y = y * __builtin_fmaf(-h, y * y, 1.5f);
    y = y * __builtin_fmaf(-h, y * y, 1.5f);
    return __builtin_fmaf(0.5f * y, __builtin_fmaf(-x, y * y, 1.0f), y);
}
RTX_HD f3 normalize(f3 a) { return a * rsqrt_det(dot(a, a)); }   // HLSL normalize(v) = v * rsqrt(dot(v, v))
RTX_HD float saturate(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
RTX_HD float maxf_(float a, float b) { return a > b ? a : b; }
RTX_HD float minf_(float a, float b) { return a < b ? a : b; }
RTX_HD bool is_nan(float x) { return x != x; }
RTX_HD bool is_inf(float x) { return fabsf(x) == INFINITY; }
RTX_HD bool finite3(f3 a) { return !(is_nan(a.x) || is_nan(a.y) || is_nan(a.z) || is_inf(a.x) || is_inf(a.y) || is_inf(a.z)); }
RTX_HD bool is_zero3(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// SafeMultiply, Common_v6.hlsl:151-160
RTX_HD f3 safe_mul(float s, f3 v) { f3 r = v * s; return finite3(r) ? r : mk3(0.0f, 0.0f, 0.0f); }
RTX_HD float safe_mul(float s, float v) { float r = s * v; return (is_nan(r) || is_inf(r)) ? 0.0f : r; }

RTX_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
RTX_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

// RandomFloat (TEA, 4 rounds), Common_v6.hlsl:119-138.  float(v0)/2^32 rounds to nearest, so 1.0f is reachable.
RTX_HD float tea_next(uint32_t& s0, uint32_t& s1) {
    uint32_t v0 = s0, v1 = s1, sum = 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    s0 = v0; s1 = v1;
    return (float)v0 * (1.0f / 4294967296.0f);
}
// per-sample seed, RayGen_v6_pass1.hlsl:63-77 (uint(time) := frame_seed, sample id := s)
RTX_HD void seed_init(uint32_t x, uint32_t y, uint32_t s, uint32_t frame_seed, uint32_t& s0, uint32_t& s1) {
    s0 = (y * 73856093u) ^ (x * 19349663u) ^ (s * 83492791u) ^ (frame_seed * 293803u);
    s1 = (x * 37623481u) ^ (y * 51964263u) ^ (s * 68250729u) ^ (frame_seed * 423977u);
}

// sin and cos of x in [0, 8): octant reduction (three-constant Cody-Waite split of pi/4) and
// degree-7 / degree-8 polynomials on [-pi/4, pi/4].  Stands in for HLSL sin()/cos() in
// Lambertian_v6.hlsl:13-14 and GGX_v6.hlsl:134-135.
RTX_HD void sincos_(float x, float& sn, float& cs) {
    int j = (int)(x * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    float r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int q = (j >> 1) & 3;
    float a = (q & 1) ? pc : ps;      // sin candidate
    float b = (q & 1) ? ps : pc;      // cos candidate
    sn = (q & 2) ? -a : a;
    cs = (q == 1 || q == 2) ? -b : b;
}

// pow(x, y), x > 0, as exp2(y * log2 x); ~1e-6 relative.  Only the sRGB OETF uses it (Common_v6.hlsl:353-376).
RTX_HD float pow_(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t ux = f2u(x);
    int e = (int)((ux >> 23) & 0xFF) - 127;
    float m = u2f((ux & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float t = (m - 1.0f) / (m + 1.0f);
    float t2 = t * t;
    float ln = 2.0f * t * (1.0f + t2 * (0.333333333f + t2 * (0.2f + t2 * (0.142857143f + t2 * 0.111111111f))));
    float l2 = (float)e + ln * 1.44269504089f;
    float p = y * l2;
    if (p < -126.0f) return 0.0f;
    if (p > 127.0f) return INFINITY;
    float fl = floorf(p + 0.5f);
    float f = p - fl;
    float g = f * 0.693147180560f;
    float ex = 1.0f + g * (1.0f + g * (0.5f + g * (0.166666667f + g * (0.0416666667f + g * (0.00833333333f + g * 0.00138888889f)))));
    return ex * u2f((uint32_t)((int)fl + 127) << 23);
}

// mul(M, float4(p,1)).xyz / mul(M, float4(v,0)).xyz for the column-major 16-float layout of rtx.h
RTX_HD f3 xform_point(const float* m, f3 p) {
    return mk3(__builtin_fmaf(m[8], p.z, __builtin_fmaf(m[4], p.y, __builtin_fmaf(m[0], p.x, m[12]))),
               __builtin_fmaf(m[9], p.z, __builtin_fmaf(m[5], p.y, __builtin_fmaf(m[1], p.x, m[13]))),
               __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[6], p.y, __builtin_fmaf(m[2], p.x, m[14]))));
}
RTX_HD f3 xform_dir(const float* m, f3 p) {
    return mk3(__builtin_fmaf(m[8], p.z, __builtin_fmaf(m[4], p.y, m[0] * p.x)),
               __builtin_fmaf(m[9], p.z, __builtin_fmaf(m[5], p.y, m[1] * p.x)),
               __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[6], p.y, m[2] * p.x)));
}

}  // namespace rtx
