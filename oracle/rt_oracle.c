/*
 * rt_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see rt_oracle.h).
 *
 * PARITY UNPINNED for the shader math: the reference has no golden vectors and
 * cannot be built here.  This file restates, in scalar C with a fixed IEEE-754
 * binary32 operation order (compile with -ffp-contract=off, no fast-math):
 *
 *   leaf math      = the live v6 shader set  (Pathtracer/include/ *_v6.hlsl under /root/reference)
 *   loop structure = the legacy bounce loop  (include/RayGen.hlsl:99-133, include/Hit.hlsl:126-174,340-369)
 *
 * as SURVEY.md §8(a) prescribes.  Every function cites the reference lines it
 * follows; deliberate deviations are tagged DEVIATION and listed in DESIGN.md.
 *
 * Transcendentals (sin/cos/pow) are implemented here with +,-,*,/ only so that
 * the HIP path can execute the identical operation sequence (the HLSL intrinsics
 * are implementation-defined to a few ULP anyway).
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: Common_v6.hlsl:1-3 ---- */
#define PI_REF   3.1415f          /* Common_v6.hlsl:1 (sic) */
#define S_BIAS   0.00002f         /* Common_v6.hlsl:2 */
#define EPSILON_ 0.000001f        /* Common_v6.hlsl:3 */
#define TWO_PI_F 6.28318548202514648f /* float(2.0 * 3.14159265358979323846): Lambertian_v6.hlsl:10 */
#define MISS_PRIM 0xFFFFFFFFu

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* ORC_LITERAL (make librt_oracle_literal.so; tests/test_ggx_pins.py): the three speed-motivated arithmetic DEVIATIONs switched OFF — mul / add chains
   unfused (two roundings), normalize(v) = v / sqrt(dot(v,v)) with IEEE sqrt and divide, x / PI as a division.  Not what the GPU is compared
   with; it exists to show that the default build's images agree with the literal reading of the HLSL within Monte-Carlo noise. */
#ifdef ORC_LITERAL
#define ORC_LIT_FMA
#define ORC_LIT_NORM
#define ORC_LIT_PI
#endif
#ifdef ORC_LIT_FMA
#define FMAF(a, b, c) ((a) * (b) + (c))
#else
#define FMAF(a, b, c) fmaf((a), (b), (c))
#endif
/* DEVIATION (fused multiply-add): dot, cross and the matrix-vector products contract a*b + c into ONE rounding, fmaf, in a fixed
   nesting.  HLSL leaves mul/add chains free to become `mad`/FMA (no `precise`), and every GPU driver compiler does so, so neither
   form is "the" reference result; the HIP kernels issue v_fma_f32 at exactly these sites (csrc/rtx_math.hpp).  fmaf is correctly
   rounded with or without hardware FMA, so the oracle's results do not depend on the build flags (see Makefile). */
static inline float dot3(v3 a, v3 b) { return FMAF(a.z, b.z, FMAF(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(FMAF(a.y, b.z, -(a.z * b.y)), FMAF(a.z, b.x, -(a.x * b.z)), FMAF(a.x, b.y, -(a.y * b.x)));
}
/* fused linear combinations (same DEVIATION as dot3 / cross3): a*s + b, and x*a + y*b + z*c as fma(z, c, fma(y, b, x*a)) */
static inline v3 madd3(v3 a, float s, v3 b) { return V3(FMAF(a.x, s, b.x), FMAF(a.y, s, b.y), FMAF(a.z, s, b.z)); }
static inline v3 lincomb3(v3 x, float a, v3 y, float b, v3 z, float c) {
    return V3(FMAF(z.x, c, FMAF(y.x, b, x.x * a)), FMAF(z.y, c, FMAF(y.y, b, x.y * a)), FMAF(z.z, c, FMAF(y.z, b, x.z * a)));
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline uint32_t f2u(float f);
static inline float u2f(uint32_t u);
/* rsqrt(x), x > 0 normal: HLSL's rsqrt is a 1-ULP implementation-defined hardware approximation; this is a DETERMINISTIC one of the
   same quality that both backends execute operation for operation: integer seed (max relative error 1.75e-3 after the first step)
   and three Newton steps in fmaf arithmetic, the last one in residual form; error < 1 ulp (tests/test_oracle_golden.py).
   13 instructions on the GPU against 33 issue slots for an IEEE sqrt followed by an IEEE divide.
   Degenerate input: rsqrt_det(0) is a large finite number, so normalize(0) = 0 (with IEEE 1/sqrt it was NaN). */
static inline float rsqrt_det(float x) {
    float y = u2f(0x5f375a86u - (f2u(x) >> 1));
    const float h = 0.5f * x;
    y = y * fmaf(-h, y * y, 1.5f);
    y = y * fmaf(-h, y * y, 1.5f);
    return fmaf(0.5f * y, fmaf(-x, y * y, 1.0f), y);
}
/* HLSL normalize(v) = v * rsqrt(dot(v,v)) */
#ifdef ORC_LIT_NORM
static inline v3 normalize3(v3 a) { const float l = sqrtf(dot3(a, a)); return V3(a.x / l, a.y / l, a.z / l); }
#else
static inline v3 normalize3(v3 a) { return scale3(a, rsqrt_det(dot3(a, a))); }
#endif
static inline float saturatef(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline int is_nan(float x) { return x != x; }
static inline int is_inf(float x) { return fabsf(x) == INFINITY; }
static inline int finite3(v3 a) { return !(is_nan(a.x) || is_nan(a.y) || is_nan(a.z) || is_inf(a.x) || is_inf(a.y) || is_inf(a.z)); }

/* Common_v6.hlsl:151-160 */
static inline v3 safe_mul3(float s, v3 v) { v3 r = scale3(v, s); return finite3(r) ? r : V3(0, 0, 0); }
static inline float safe_mul1(float s, float v) { float r = s * v; return (is_nan(r) || is_inf(r)) ? 0.0f : r; }

/* ---- TEA-4 RNG: Common_v6.hlsl:119-138 ---- */
static inline float rnd(uint32_t seed[2]) {
    uint32_t v0 = seed[0], v1 = seed[1], sum = 0u;
    for (int i = 0; i < 4; i++) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xA341316Cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xC8013EA4u);
        v1 += ((v0 << 4) + 0xAD90777Du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7E95761Eu);
    }
    seed[0] = v0; seed[1] = v1;
    return (float)v0 * (1.0f / 4294967296.0f);   /* float(v0)/2^32; may return exactly 1.0f */
}
void orc_tea(uint32_t seed[2], uint32_t n, float* out) { for (uint32_t i = 0; i < n; i++) out[i] = rnd(seed); }

/* RayGen_v6_pass1.hlsl:63-77 (s = sample id; uint(time) := frame_seed) */
void orc_seed_init(uint32_t x, uint32_t y, uint32_t s, uint32_t frame_seed, uint32_t out[2]) {
    out[0] = (y * 73856093u) ^ (x * 19349663u) ^ (s * 83492791u) ^ (frame_seed * 293803u);
    out[1] = (x * 37623481u) ^ (y * 51964263u) ^ (s * 68250729u) ^ (frame_seed * 423977u);
}

/* ---- own transcendentals (replace HLSL sin/cos/pow; see file header) ---- */
/* sin & cos for x in [0, 8): Cody-Waite reduction by pi/4 octants + degree-7/8 minimax polynomials */
void orc_sincos(float x, float* sn, float* cs) {
    int j = (int)(x * 1.27323954473516f);       /* x * 4/pi, truncation (x >= 0) */
    j = (j + 1) & ~1;                           /* round up to even octant */
    float y = (float)j;
    float r = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    switch ((j >> 1) & 3) {
    case 0: *sn = ps;  *cs = pc;  break;
    case 1: *sn = pc;  *cs = -ps; break;
    case 2: *sn = -ps; *cs = -pc; break;
    default: *sn = -pc; *cs = ps; break;
    }
}
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
void orc_rsqrt(const float* x, uint32_t n, float* out) { for (uint32_t i = 0; i < n; i++) out[i] = rsqrt_det(x[i]); }   /* KAT hook */
/* pow(x,y) for x > 0 via exp2(y*log2(x)); ~1e-6 relative; only used for the sRGB OETF (Common_v6.hlsl:353-376) */
float orc_pow(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    uint32_t ux = f2u(x);
    int e = (int)((ux >> 23) & 0xFF) - 127;
    float m = u2f((ux & 0x007FFFFFu) | 0x3F800000u);         /* [1,2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }           /* [0.707,1.414) */
    float t = (m - 1.0f) / (m + 1.0f);
    float t2 = t * t;
    float ln = 2.0f * t * (1.0f + t2 * (0.333333333f + t2 * (0.2f + t2 * (0.142857143f + t2 * 0.111111111f))));
    float l2 = (float)e + ln * 1.44269504089f;
    float p = y * l2;
    if (p < -126.0f) return 0.0f;
    if (p > 127.0f) return INFINITY;
    float fl = floorf(p + 0.5f);
    float f = p - fl;                                        /* [-0.5,0.5] */
    float g = f * 0.693147180560f;
    float ex = 1.0f + g * (1.0f + g * (0.5f + g * (0.166666667f + g * (0.0416666667f + g * (0.00833333333f + g * 0.00138888889f)))));
    return ex * u2f((uint32_t)((int)fl + 127) << 23);
}
/* float -> binary16 (round to nearest even) -> float: half4(mat.Kd) etc., Sampler_v6.hlsl:71-83 */
float orc_half_round(float x) {
    uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;                            /* inf / nan */
    if (a >= 0x477FF000u) return u2f(sign | 0x7F800000u);      /* >= 65520 -> inf */
    if (a < 0x33000001u) return u2f(sign);                     /* <= 2^-25 -> 0 */
    if (a < 0x38800000u) {                                     /* subnormal half: quantum 2^-24 */
        float q = u2f(a) * 16777216.0f;                        /* exact scaling */
        float r = nearbyintf(q);                               /* default rounding mode: RNE */
        return u2f(sign | f2u(r * (1.0f / 16777216.0f)));
    }
    uint32_t rem = a & 0x1FFFu, base = a & ~0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}

/* ---- matrices: column-major storage of column-vector matrices ---- */
/* general 4x4 inverse by cofactors, evaluated in double then rounded (stands in for XMMatrixInverse,
   Renderer.cpp:1735-1736, 2101-2118) */
void orc_mat4_inverse(const float* mf, float* out) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = (double)mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double id = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * id);
}
/* mul(M, float4(p,1)).xyz */
static inline v3 xform_point(const float* m, v3 p) {
    return V3(FMAF(m[8], p.z, FMAF(m[4], p.y, FMAF(m[0], p.x, m[12]))),
              FMAF(m[9], p.z, FMAF(m[5], p.y, FMAF(m[1], p.x, m[13]))),
              FMAF(m[10], p.z, FMAF(m[6], p.y, FMAF(m[2], p.x, m[14]))));
}
/* mul(M, float4(v,0)).xyz */
static inline v3 xform_dir(const float* m, v3 p) {
    return V3(FMAF(m[8], p.z, FMAF(m[4], p.y, m[0] * p.x)),
              FMAF(m[9], p.z, FMAF(m[5], p.y, m[1] * p.x)),
              FMAF(m[10], p.z, FMAF(m[6], p.y, m[2] * p.x)));
}

/* ---- scene ---- */
typedef struct {            /* fp16-rounded working copy: MaterialOptimized, Common_v6.hlsl:62-74 */
    v3 Kd; float alpha;
    float Pr, Pm, Ps, Pc;
    float Ni;               /* full-precision Material.Ni (strategy-3 EXTENSION only; MaterialOptimized has no such member) */
    v3 Ks; v3 Ke;
    float Ke_len;           /* length(Ke) of the rounded copy */
} matopt_t;

typedef struct { float* verts; uint32_t nverts; uint32_t* idx; uint32_t nidx; uint32_t matid_base; } mesh_t;
typedef struct { uint32_t mesh; float o2w[16]; float nrm[16]; float o2w_inv[16]; float prev_o2w[16]; uint32_t tri_base; } inst_t;
typedef struct { float bmin[3], bmax[3]; uint32_t left, right, first, count; } node_t;   /* count>0 => leaf */

struct orc_ctx {
    float* mats; matopt_t* mopt; uint32_t nmat;
    mesh_t* meshes; uint32_t nmesh;
    uint32_t* matids; uint32_t nmatids;
    inst_t* insts; uint32_t ninst;
    /* flattened world-space triangles */
    uint32_t ntri; float* wtri; float* tri_floor; uint32_t* tri_inst; uint32_t* tri_prim;   /* tri_floor: the hit definition's determinant floor per triangle (tri_det_floor) */
    /* lights (80-byte records as 20 floats) */
    float* lights; uint32_t nlights; float total_weight;
    /* bvh */
    node_t* nodes; uint32_t nnodes; uint32_t* order;
    float view[16], proj[16], viewI[16], projI[16];
    float prev_view[16], prev_proj[16]; int have_cam;      /* m_prevViewMatrix / m_prevProjMatrix, Renderer.cpp:1766-1767 */
    int nthreads;
};

orc_ctx* orc_create(void) { return (orc_ctx*)calloc(1, sizeof(orc_ctx)); }
void orc_destroy(orc_ctx* c) {
    if (!c) return;
    free(c->mats); free(c->mopt);
    for (uint32_t i = 0; i < c->nmesh; i++) { free(c->meshes[i].verts); free(c->meshes[i].idx); }
    free(c->meshes); free(c->matids); free(c->insts);
    free(c->wtri); free(c->tri_floor); free(c->tri_inst); free(c->tri_prim); free(c->lights); free(c->nodes); free(c->order);
    free(c);
}
int orc_set_threads(orc_ctx* c, int n) { c->nthreads = n; return 0; }

int orc_set_materials(orc_ctx* c, const void* mats128, uint32_t count) {
    free(c->mats); free(c->mopt);
    c->mats = (float*)malloc((size_t)count * 128);
    c->mopt = (matopt_t*)malloc((size_t)count * sizeof(matopt_t));
    memcpy(c->mats, mats128, (size_t)count * 128);
    c->nmat = count;
    for (uint32_t i = 0; i < count; i++) {
        const float* m = c->mats + i * 32;  /* Kd[4] Ks[3] Ni Ke[3] pad Pr_Pm_Ps_Pc[4] LUT[16]: Vertex.h:14-23 */
        matopt_t* o = &c->mopt[i];
        o->Kd = V3(orc_half_round(m[0]), orc_half_round(m[1]), orc_half_round(m[2])); o->alpha = orc_half_round(m[3]); o->Ni = m[7];
        o->Ks = V3(orc_half_round(m[4]), orc_half_round(m[5]), orc_half_round(m[6]));
        o->Ke = V3(orc_half_round(m[8]), orc_half_round(m[9]), orc_half_round(m[10]));
        o->Pr = orc_half_round(m[12]); o->Pm = orc_half_round(m[13]); o->Ps = orc_half_round(m[14]); o->Pc = orc_half_round(m[15]);
        o->Ke_len = length3(o->Ke);
    }
    return 0;
}

int orc_add_mesh(orc_ctx* c, const void* verts28, uint32_t nverts, const uint32_t* indices, uint32_t nidx,
                 const uint32_t* material_ids, uint32_t* mesh_out) {
    if (nidx % 3) return -1;
    const float* v = (const float*)verts28;
    for (uint32_t i = 0; i < nidx; i++) if (indices[i] >= nverts) return -2;
    /* Vertex.normal.w is the base of this model inside the global materialIDs[] (ObjLoader.h:466, Hit_v6.hlsl:17) */
    for (uint32_t i = 0; i < nverts; i++) if ((uint32_t)v[i * 7 + 6] != c->nmatids) return -3;
    c->meshes = (mesh_t*)realloc(c->meshes, (c->nmesh + 1) * sizeof(mesh_t));
    mesh_t* m = &c->meshes[c->nmesh];
    m->verts = (float*)malloc((size_t)nverts * 28); memcpy(m->verts, v, (size_t)nverts * 28);
    m->idx = (uint32_t*)malloc((size_t)nidx * 4); memcpy(m->idx, indices, (size_t)nidx * 4);
    m->nverts = nverts; m->nidx = nidx; m->matid_base = c->nmatids;
    c->matids = (uint32_t*)realloc(c->matids, (size_t)(c->nmatids + nidx) * 4);
    memcpy(c->matids + c->nmatids, material_ids, (size_t)nidx * 4);
    c->nmatids += nidx;
    if (mesh_out) *mesh_out = c->nmesh;
    c->nmesh++;
    return 0;
}

/* Renderer.cpp:2091-2121: normal matrix = transpose(inverse(upper 3x3, rest identity)) */
static void normal_matrix(const float* o2w, float* out) {
    float u[16], inv[16];
    memcpy(u, o2w, 64);
    u[3] = u[7] = u[11] = 0.0f; u[12] = u[13] = u[14] = 0.0f; u[15] = 1.0f;
    orc_mat4_inverse(u, inv);
    for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) out[cc * 4 + r] = inv[r * 4 + cc];
}

int orc_add_instance(orc_ctx* c, uint32_t mesh, const float* o2w16, uint32_t* inst_out) {
    if (mesh >= c->nmesh) return -1;
    c->insts = (inst_t*)realloc(c->insts, (c->ninst + 1) * sizeof(inst_t));
    inst_t* in = &c->insts[c->ninst];
    in->mesh = mesh; memcpy(in->o2w, o2w16, 64); normal_matrix(o2w16, in->nrm); in->tri_base = 0;
    orc_mat4_inverse(o2w16, in->o2w_inv); memcpy(in->prev_o2w, o2w16, 64);      /* Renderer.cpp:2098-2102 (static scene: prev = current) */
    if (inst_out) *inst_out = c->ninst;
    c->ninst++;
    return 0;
}

/* UpdateInstancePropertiesBuffer, Renderer.cpp:2091-2121: prev := current, then the new matrix; call orc_commit afterwards */
int orc_set_instance_transform(orc_ctx* c, uint32_t inst, const float* o2w16) {
    if (inst >= c->ninst) return -1;
    inst_t* in = &c->insts[inst];
    memcpy(in->prev_o2w, in->o2w, 64);
    memcpy(in->o2w, o2w16, 64); normal_matrix(o2w16, in->nrm); orc_mat4_inverse(o2w16, in->o2w_inv);
    return 0;
}

static inline v3 mesh_pos(const mesh_t* m, uint32_t vi) { const float* p = m->verts + (size_t)vi * 7; return V3(p[0], p[1], p[2]); }
static inline v3 mesh_nrm(const mesh_t* m, uint32_t vi) { const float* p = m->verts + (size_t)vi * 7; return V3(p[3], p[4], p[5]); }

/* ---- light list: Renderer.cpp:2123-2233, 2237-2243 ---- */
typedef struct { float w; uint32_t order; float rec[20]; } lt_tmp;
static int lt_cmp(const void* a, const void* b) {
    const lt_tmp* x = (const lt_tmp*)a; const lt_tmp* y = (const lt_tmp*)b;
    if (x->w > y->w) return -1;
    if (x->w < y->w) return 1;
    return (x->order > y->order) - (x->order < y->order);   /* DEVIATION: std::sort is unstable; ties broken by collection order */
}
static void build_lights(orc_ctx* c) {
    free(c->lights); c->lights = NULL; c->nlights = 0; c->total_weight = 0.0f;
    uint32_t cap = 0, n = 0; lt_tmp* tmp = NULL;
    for (uint32_t ii = 0; ii < c->ninst; ii++) {
        const mesh_t* m = &c->meshes[c->insts[ii].mesh];
        for (uint32_t t = 0; t < m->nidx / 3; t++) {
            uint32_t m0 = c->matids[m->matid_base + t * 3], m1 = c->matids[m->matid_base + t * 3 + 1], m2 = c->matids[m->matid_base + t * 3 + 2];
            if (m0 != m1 || m0 != m2) continue;                     /* Renderer.cpp:2153-2156 */
            if (m0 >= c->nmat) continue;
            const float* mat = c->mats + (size_t)m0 * 32;
            if (!(mat[8] + mat[9] + mat[10] > 0.0f)) continue;      /* :2162 */
            v3 p0 = mesh_pos(m, m->idx[t * 3]), p1 = mesh_pos(m, m->idx[t * 3 + 1]), p2 = mesh_pos(m, m->idx[t * 3 + 2]);
            /* ComputeTriangleWeight :2217-2233 */
            float area = 0.5f * length3(cross3(sub3(p1, p0), sub3(p2, p0)));
            float inten = (mat[8] + mat[9] + mat[10]) / 3.0f;
            if (n == cap) { cap = cap ? cap * 2 : 64; tmp = (lt_tmp*)realloc(tmp, cap * sizeof(lt_tmp)); }
            lt_tmp* L = &tmp[n];
            memset(L, 0, sizeof(*L));
            L->w = area * inten; L->order = n;
            L->rec[0] = p0.x; L->rec[1] = p0.y; L->rec[2] = p0.z;
            L->rec[4] = p1.x; L->rec[5] = p1.y; L->rec[6] = p1.z; memcpy(&L->rec[7], &ii, 4);
            L->rec[8] = p2.x; L->rec[9] = p2.y; L->rec[10] = p2.z; L->rec[11] = L->w;
            L->rec[12] = mat[8]; L->rec[13] = mat[9]; L->rec[14] = mat[10];
            n++;
        }
    }
    if (!n) { free(tmp); return; }
    qsort(tmp, n, sizeof(lt_tmp), lt_cmp);
    float total = 0.0f;
    for (uint32_t i = 0; i < n; i++) total += tmp[i].rec[11];
    float cum = 0.0f;
    c->lights = (float*)malloc((size_t)n * 80);
    for (uint32_t i = 0; i < n; i++) {
        float* r = tmp[i].rec;
        r[11] = r[11] / total; cum += r[11]; r[3] = cum; r[16] = total;
        memcpy(&r[15], &n, 4);                                         /* triCount :2240-2243 */
        memcpy(c->lights + (size_t)i * 20, r, 80);
    }
    c->lights[(size_t)(n - 1) * 20 + 3] = 1.0f;                        /* :2208-2210 */
    c->nlights = n; c->total_weight = total;
    free(tmp);
}

/* ---- the oracle's own BVH (median split); purely an accelerator, results must equal brute force ---- */
static void tri_bounds(const float* t, float* mn, float* mx) {
    for (int a = 0; a < 3; a++) {
        mn[a] = minf(t[a], minf(t[3 + a], t[6 + a]));
        mx[a] = maxf(t[a], maxf(t[3 + a], t[6 + a]));
    }
}
typedef struct { const orc_ctx* c; int axis; } sortctx;
static const float* g_sort_wtri; static int g_sort_axis;
static int cen_cmp(const void* a, const void* b) {
    uint32_t i = *(const uint32_t*)a, j = *(const uint32_t*)b;
    const float* ti = g_sort_wtri + (size_t)i * 9; const float* tj = g_sort_wtri + (size_t)j * 9;
    float ci = ti[g_sort_axis] + ti[3 + g_sort_axis] + ti[6 + g_sort_axis];
    float cj = tj[g_sort_axis] + tj[3 + g_sort_axis] + tj[6 + g_sort_axis];
    if (ci < cj) return -1;
    if (ci > cj) return 1;
    return (i > j) - (i < j);
}
static uint32_t build_node(orc_ctx* c, uint32_t first, uint32_t count, float pad) {
    uint32_t id = c->nnodes++;
    node_t* nd = &c->nodes[id];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < count; i++) {
        float a[3], b[3]; tri_bounds(c->wtri + (size_t)c->order[first + i] * 9, a, b);
        for (int k = 0; k < 3; k++) { mn[k] = minf(mn[k], a[k]); mx[k] = maxf(mx[k], b[k]); }
    }
    for (int k = 0; k < 3; k++) { nd->bmin[k] = mn[k] - pad; nd->bmax[k] = mx[k] + pad; }
    if (count <= 4) { nd->first = first; nd->count = count; nd->left = nd->right = 0; return id; }
    int axis = 0; float ext = mx[0] - mn[0];
    if (mx[1] - mn[1] > ext) { axis = 1; ext = mx[1] - mn[1]; }
    if (mx[2] - mn[2] > ext) axis = 2;
    g_sort_wtri = c->wtri; g_sort_axis = axis;
    qsort(c->order + first, count, 4, cen_cmp);
    uint32_t half = count / 2;
    nd->count = 0; nd->first = 0;
    uint32_t l = build_node(c, first, half, pad);
    uint32_t r = build_node(c, first + half, count - half, pad);
    c->nodes[id].left = l; c->nodes[id].right = r;
    return id;
}

static inline float tri_det_floor(const float* t9);
int orc_commit(orc_ctx* c) {
    uint32_t nt = 0;
    for (uint32_t i = 0; i < c->ninst; i++) { c->insts[i].tri_base = nt; nt += c->meshes[c->insts[i].mesh].nidx / 3; }
    free(c->wtri); free(c->tri_floor); free(c->tri_inst); free(c->tri_prim); free(c->nodes); free(c->order);
    c->ntri = nt;
    c->wtri = (float*)malloc((size_t)(nt ? nt : 1) * 36);
    c->tri_floor = (float*)malloc((size_t)(nt ? nt : 1) * 4);
    c->tri_inst = (uint32_t*)malloc((size_t)(nt ? nt : 1) * 4);
    c->tri_prim = (uint32_t*)malloc((size_t)(nt ? nt : 1) * 4);
    float scale = 1.0f;
    for (uint32_t i = 0; i < c->ninst; i++) {
        const inst_t* in = &c->insts[i]; const mesh_t* m = &c->meshes[in->mesh];
        for (uint32_t t = 0; t < m->nidx / 3; t++) {
            uint32_t g = in->tri_base + t;
            for (int k = 0; k < 3; k++) {
                v3 w = xform_point(in->o2w, mesh_pos(m, m->idx[t * 3 + k]));
                c->wtri[(size_t)g * 9 + k * 3] = w.x; c->wtri[(size_t)g * 9 + k * 3 + 1] = w.y; c->wtri[(size_t)g * 9 + k * 3 + 2] = w.z;
                scale = maxf(scale, maxf(fabsf(w.x), maxf(fabsf(w.y), fabsf(w.z))));
            }
            c->tri_inst[g] = i; c->tri_prim[g] = t;
            c->tri_floor[g] = tri_det_floor(c->wtri + (size_t)g * 9);
        }
    }
    build_lights(c);
    c->order = (uint32_t*)malloc((size_t)(nt ? nt : 1) * 4);
    for (uint32_t i = 0; i < nt; i++) c->order[i] = i;
    c->nodes = (node_t*)malloc((size_t)(2 * (nt ? nt : 1)) * sizeof(node_t));
    c->nnodes = 0;
    if (nt) build_node(c, 0, nt, 1e-5f * scale);
    return 0;
}

int orc_set_camera(orc_ctx* c, const float* view16, const float* proj16) {
    if (c->have_cam) { memcpy(c->prev_view, c->view, 64); memcpy(c->prev_proj, c->proj, 64); }
    else { memcpy(c->prev_view, view16, 64); memcpy(c->prev_proj, proj16, 64); c->have_cam = 1; }
    memcpy(c->view, view16, 64); memcpy(c->proj, proj16, 64);
    orc_mat4_inverse(view16, c->viewI); orc_mat4_inverse(proj16, c->projI);   /* Renderer.cpp:1735-1736 */
    return 0;
}
uint32_t orc_num_triangles(orc_ctx* c) { return c->ntri; }
uint32_t orc_num_lights(orc_ctx* c) { return c->nlights; }
int orc_get_lights(orc_ctx* c, void* out80, uint32_t max_count) {
    uint32_t n = c->nlights < max_count ? c->nlights : max_count;
    memcpy(out80, c->lights, (size_t)n * 80);
    return (int)n;
}

/* ---- ray / triangle (a11: TraceRay closest hit; DXR range is exclusive TMin < t < TMax) ---- */
typedef struct { float t, u, v; uint32_t prim; } hit_t;

/* Moeller-Trumbore, no culling (geometry opaque, RAY_FLAG_NONE).  The BVH format and the intersection
   arithmetic of DXR are driver-opaque; this fixed operation order IS the definition both backends share.
   det_floor (DEVIATION from plain Moeller-Trumbore, which only excludes det == 0; csrc/rtx_math.hpp: tri_det_floor): det = -d . (e1 x e2) is rounding noise — at most
   ~7 ulp(|e1| |e2|) for a unit d — when the ray lies in the triangle's plane, and u, v, t are then 0 / 0: u = v = -0 is accepted with an arbitrary t, a "hit" outside every
   bounding volume that brute force reports and a BVH culls or not depending on its boxes.  A triangle is hit only if |det| > 2^-16 |e1| |e2| (32 x that bound of 2^-21 |e1| |e2|). */
#define DET_REL 1.52587890625e-05f
static inline float tri_det_floor(const float* t9) {
    v3 v0 = V3(t9[0], t9[1], t9[2]);
    v3 e1 = sub3(V3(t9[3], t9[4], t9[5]), v0), e2 = sub3(V3(t9[6], t9[7], t9[8]), v0);
    return DET_REL * (sqrtf(dot3(e1, e1)) * sqrtf(dot3(e2, e2)));
}
static inline int tri_hit(v3 o, v3 d, const float* t9, float det_floor, float tmin, float tmax, float* to, float* uo, float* vo) {
    v3 v0 = V3(t9[0], t9[1], t9[2]);
    v3 e1 = sub3(V3(t9[3], t9[4], t9[5]), v0), e2 = sub3(V3(t9[6], t9[7], t9[8]), v0);
    v3 p = cross3(d, e2);
    float det = dot3(e1, p);
    if (!(fabsf(det) > det_floor)) return 0;
    float inv = 1.0f / det;
    v3 s = sub3(o, v0);
    float u = dot3(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    v3 q = cross3(s, e1);
    float v = dot3(d, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    float t = dot3(e2, q) * inv;
    if (!(t > tmin && t < tmax)) return 0;
    *to = t; *uo = u; *vo = v;
    return 1;
}
/* closest = min t, ties -> lowest global triangle id: order-independent */
static inline void closest_update(hit_t* h, float t, float u, float v, uint32_t prim) {
    if (t < h->t || (t == h->t && prim < h->prim)) { h->t = t; h->u = u; h->v = v; h->prim = prim; }
}
static hit_t closest_brute(const orc_ctx* c, v3 o, v3 d, float tmin, float tmax) {
    hit_t h = {tmax, 0, 0, MISS_PRIM};
    for (uint32_t i = 0; i < c->ntri; i++) {
        float t, u, v;
        if (tri_hit(o, d, c->wtri + (size_t)i * 9, c->tri_floor[i], tmin, tmax, &t, &u, &v)) closest_update(&h, t, u, v, i);
    }
    return h;
}
/* conservative slab test: relative padding of the interval so that no accepted triangle hit is culled */
static inline int box_hit(const node_t* n, v3 o, v3 d, v3 inv, float tmin, float tbest) {
    float te = -INFINITY, tx = INFINITY;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ii[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; a++) {
        if (dd[a] == 0.0f) { if (oo[a] < n->bmin[a] || oo[a] > n->bmax[a]) return 0; continue; }
        float t1 = (n->bmin[a] - oo[a]) * ii[a], t2 = (n->bmax[a] - oo[a]) * ii[a];
        float lo = minf(t1, t2), hi = maxf(t1, t2);
        te = maxf(te, lo); tx = minf(tx, hi);
    }
    /* generous: the margins must cover the error of tri_hit's t, not only the slab arithmetic: ~1e-5 relative next to a vertex of a
       small far triangle, and 1e-3 for a ray almost parallel to the triangle (tests/test_gpu_parity.py, "duplicates" soup: a
       coincident duplicate lost its lowest-id tie at 1e-4).  The oracle's BVH only has to be conservative, not fast. */
    te = te - fabsf(te) * 4e-3f; tx = tx + fabsf(tx) * 4e-3f;
    return te <= tx && tx >= tmin && te <= tbest;
}
static hit_t closest_bvh(const orc_ctx* c, v3 o, v3 d, float tmin, float tmax) {
    hit_t h = {tmax, 0, 0, MISS_PRIM};
    if (!c->ntri) return h;
    v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const node_t* n = &c->nodes[stack[--sp]];
        if (!box_hit(n, o, d, inv, tmin, h.t)) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; i++) {
                uint32_t g = c->order[n->first + i]; float t, u, v;
                if (tri_hit(o, d, c->wtri + (size_t)g * 9, c->tri_floor[g], tmin, tmax, &t, &u, &v)) closest_update(&h, t, u, v, g);
            }
        } else { stack[sp++] = n->left; stack[sp++] = n->right; }
    }
    return h;
}
static int any_brute(const orc_ctx* c, v3 o, v3 d, float tmin, float tmax) {
    for (uint32_t i = 0; i < c->ntri; i++) { float t, u, v; if (tri_hit(o, d, c->wtri + (size_t)i * 9, c->tri_floor[i], tmin, tmax, &t, &u, &v)) return 1; }
    return 0;
}
static int any_bvh(const orc_ctx* c, v3 o, v3 d, float tmin, float tmax) {
    if (!c->ntri) return 0;
    v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp) {
        const node_t* n = &c->nodes[stack[--sp]];
        if (!box_hit(n, o, d, inv, tmin, tmax)) continue;
        if (n->count) {
            for (uint32_t i = 0; i < n->count; i++) {
                float t, u, v;
                if (tri_hit(o, d, c->wtri + (size_t)c->order[n->first + i] * 9, c->tri_floor[c->order[n->first + i]], tmin, tmax, &t, &u, &v)) return 1;
            }
        } else { stack[sp++] = n->left; stack[sp++] = n->right; }
    }
    return 0;
}

int orc_trace_closest(orc_ctx* c, const float* r, uint32_t n, int mode, float* hits4) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float* q = r + i * 8;
        hit_t h = mode ? closest_bvh(c, V3(q[0], q[1], q[2]), V3(q[4], q[5], q[6]), q[3], q[7])
                       : closest_brute(c, V3(q[0], q[1], q[2]), V3(q[4], q[5], q[6]), q[3], q[7]);
        hits4[i * 4] = h.t; hits4[i * 4 + 1] = h.u; hits4[i * 4 + 2] = h.v; memcpy(&hits4[i * 4 + 3], &h.prim, 4);
    }
    return 0;
}
int orc_trace_any(orc_ctx* c, const float* r, uint32_t n, int mode, uint8_t* occ) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float* q = r + i * 8;
        occ[i] = (uint8_t)(mode ? any_bvh(c, V3(q[0], q[1], q[2]), V3(q[4], q[5], q[6]), q[3], q[7])
                                : any_brute(c, V3(q[0], q[1], q[2]), V3(q[4], q[5], q[6]), q[3], q[7]));
    }
    return 0;
}

/* ---- ClosestHit: Hit_v6.hlsl:12-61 ---- */
typedef struct { v3 pos; uint32_t mat; v3 normal; float area; uint32_t inst; v3 flat; } surf_t;
static surf_t surface(const orc_ctx* c, v3 o, v3 d, hit_t h) {
    surf_t s;
    uint32_t ii = c->tri_inst[h.prim], prim = c->tri_prim[h.prim];
    const inst_t* in = &c->insts[ii]; const mesh_t* m = &c->meshes[in->mesh];
    s.inst = ii;
    s.pos = madd3(d, h.t, o);                                                 /* :15,60 */
    uint32_t vert = 3 * prim;
    uint32_t i0 = m->idx[vert], i1 = m->idx[vert + 1], i2 = m->idx[vert + 2];
    s.mat = c->matids[vert + (uint32_t)m->verts[(size_t)i0 * 7 + 6]];          /* :17 */
    float bary[3] = {1.0f - h.u - h.v, h.u, h.v};                              /* :18 */
    v3 p0 = mesh_pos(m, i0);
    v3 e1 = sub3(mesh_pos(m, i1), p0), e2 = sub3(mesh_pos(m, i2), p0);         /* :28-29 */
    v3 cr = cross3(e1, e2);
    s.area = fabsf(length3(cr) * 0.5f);                                        /* :31 */
    v3 flat = normalize3(cr);                                                  /* :32 */
    s.flat = flat;
    const uint32_t vi[3] = {i0, i1, i2};
    v3 use[3];
    for (int k = 0; k < 3; k++) {                                              /* :40-46, all(n != 0) is per component */
        v3 nk = mesh_nrm(m, vi[k]);
        use[k] = (nk.x != 0.0f && nk.y != 0.0f && nk.z != 0.0f) ? nk : flat;
    }
    v3 smooth = lincomb3(use[0], bary[0], use[1], bary[1], use[2], bary[2]);
    v3 n = (length3(smooth) > 0.0001f) ? normalize3(smooth) : flat;            /* :49-54 */
    s.normal = normalize3(xform_dir(in->nrm, n));                              /* :56 */
    return s;
}
int orc_surface(orc_ctx* c, const float* rays8, const float* hits4, uint32_t n, float* out16) {
    for (uint32_t i = 0; i < n; i++) {
        hit_t h; h.t = hits4[i * 4]; h.u = hits4[i * 4 + 1]; h.v = hits4[i * 4 + 2]; memcpy(&h.prim, &hits4[i * 4 + 3], 4);
        float* o = out16 + (size_t)i * 16; memset(o, 0, 64);
        if (h.prim == MISS_PRIM) { uint32_t m = 0xFFFFFFFEu; memcpy(&o[3], &m, 4); continue; }   /* Miss_v6.hlsl:3-7 */
        const float* q = rays8 + (size_t)i * 8;
        surf_t s = surface(c, V3(q[0], q[1], q[2]), V3(q[4], q[5], q[6]), h);
        o[0] = s.pos.x; o[1] = s.pos.y; o[2] = s.pos.z; memcpy(&o[3], &s.mat, 4);
        o[4] = s.normal.x; o[5] = s.normal.y; o[6] = s.normal.z; o[7] = s.area; memcpy(&o[8], &s.inst, 4);
        o[9] = s.flat.x; o[10] = s.flat.y; o[11] = s.flat.z;
    }
    return 0;
}

/* ---- BSDF leaf math ---- */
/* GGX_v6.hlsl:26-29; pow(abs(1-c),5) restated as repeated multiplication */
static inline v3 schlick(v3 F0, float cosT) {
    float x = fabsf(1.0f - cosT);
    float x2 = x * x; float x5 = x2 * x2 * x;
    return V3(saturatef(F0.x + (1.0f - F0.x) * x5), saturatef(F0.y + (1.0f - F0.y) * x5), saturatef(F0.z + (1.0f - F0.z) * x5));
}
/* GGX_v6.hlsl:31-40 */
static inline float d_ggx(float NdotH, float rough) {
    float alpha = rough * rough, alpha2 = alpha * alpha, nh2 = NdotH * NdotH;
    float den = nh2 * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI_REF * den * den);
}
/* GGX_v6.hlsl:43-52 */
static inline float g2_smith(float NdotV, float NdotL, float alpha) {
    float a2 = alpha * alpha;
    float dA = NdotV * sqrtf(a2 + (1.0f - a2) * NdotL * NdotL);
    float dB = NdotL * sqrtf(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * NdotL * NdotV / (dA + dB);
}
/* GGX_v6.hlsl:55-61 */
static inline float g1_smith(float NdotV, float alpha) {
    float a2 = alpha * alpha;
    float dC = sqrtf(a2 + (1.0f - a2) * NdotV * NdotV) + NdotV;
    return 2.0f * NdotV / dC;
}
/* GGX_v6.hlsl:1-23 (LUT from the full-precision Material record) */
static inline float ess_lut(const float* mat, float NdotV) {
    NdotV = saturatef(NdotV);
    float f = NdotV * 15.0f;
    int i0 = (int)floorf(f);
    int i1 = i0 + 1 < 15 ? i0 + 1 : 15;
    float w = f - (float)i0;
    float v0 = mat[16 + i0], v1 = mat[16 + i1];
    return v0 + w * (v1 - v0);                                   /* lerp */
}
/* Lambertian_v6.hlsl:54-58 */
static inline v3 lambert_eval(const matopt_t* m) { return V3(m->Kd.x / PI_REF, m->Kd.y / PI_REF, m->Kd.z / PI_REF); }
/* Lambertian_v6.hlsl:61-64: max(dot(n, -incoming), EPS)/PI, with L = -incoming */
/* x / PI as x * (1 / PI), the reciprocal rounded once at compile time: what shader compilers do with a division by a constant */
#define INV_PI_REF (1.0f / PI_REF)
#ifdef ORC_LIT_PI
static inline float lambert_pdf(v3 n, v3 L) { return maxf(dot3(n, L), EPSILON_) / PI_REF; }
#else
static inline float lambert_pdf(v3 n, v3 L) { return maxf(dot3(n, L), EPSILON_) * INV_PI_REF; }
#endif
/* GGX_v6.hlsl:174-206; V = outgoing, L = -incoming (dots NOT clamped) */
static v3 ggx_eval(const matopt_t* m, const float* mat, v3 normal, v3 Lin, v3 Vin) {
    v3 N = normalize3(normal), V = normalize3(Vin), L = normalize3(Lin);
    v3 H = normalize3(add3(V, L));
    float NdotV = dot3(N, V), NdotL = dot3(N, L), NdotH = dot3(N, H), VdotH = dot3(V, H);
    v3 F = schlick(m->Ks, VdotH);
    float D = d_ggx(NdotH, m->Pr);
    float G = g2_smith(NdotV, NdotL, m->Pr * m->Pr);
    float den = 4.0f * NdotV * NdotL;
    if (den < EPSILON_) return V3(0, 0, 0);
    v3 spec = V3(F.x * D * G / den, F.y * D * G / den, F.z * D * G / den);
    float Ess = ess_lut(mat, NdotV);
    float kms = (1.0f - Ess) / Ess;
    v3 r = V3(spec.x * (1.0f + m->Ks.x * kms), spec.y * (1.0f + m->Ks.y * kms), spec.z * (1.0f + m->Ks.z * kms));
    return finite3(r) ? r : V3(0, 0, 0);
}
/* GGX_v6.hlsl:209-224 */
static float ggx_pdf(const matopt_t* m, v3 normal, v3 Lin, v3 Vin) {
    v3 N = normalize3(normal), V = normalize3(Vin), L = normalize3(Lin);
    v3 H = normalize3(add3(V, L));
    float NdotH = dot3(N, H), NdotV = dot3(N, V);
    float alpha = m->Pr * m->Pr;
    return g1_smith(NdotV, alpha) * d_ggx(NdotH, m->Pr) / (NdotV * 4.0f);
}
/* ---- EXTENSION: strategy 3, rough dielectric transmission (ORC_FLAG_TRANSMISSION).  The reference has it as a stub only ("3 - Refraction",
   `//p_d *= alpha;`, `// Refraction, currently replaced by diffuse (later 3)`, `//SampleBTDF_GGX` ...: BRDF_v6.hlsl:5,28-29,44-47,85-87,102-104,
   120-122); there is nothing to restate, so this part of the oracle is UNPINNED BY DEFINITION and pinned by its own properties
   (tests/test_dielectric.py).  Same operation order as royaltracer-dx_amd/csrc/rtx_bsdf.hpp, which documents the model. */
static inline float transmission_eta(const matopt_t* m, uint32_t flags, v3 outgoing, v3* normal) {
    if (!(flags & ORC_FLAG_TRANSMISSION) || (flags & ORC_FLAG_LAMBERT_ONLY) || !(m->alpha < 1.0f) || fabsf(m->Ni - 1.0f) < 0.01f) return 0.0f;
    if (dot3(*normal, outgoing) < 0.0f) *normal = neg3(*normal);     /* thin-pane model: every crossing is air -> Ni, from either side (rtx_bsdf.hpp) */
    return m->Ni;
}
static v3 btdf_eval(const matopt_t* m, v3 normal, v3 Lin, v3 Vin, float eta_p, float* pdf) {
    *pdf = 0.0f;
    const v3 zero = V3(0, 0, 0);
    v3 N = normalize3(normal), V = normalize3(Vin), L = normalize3(Lin);
    float NdotV = dot3(N, V), NdotL = dot3(N, L);
    if (!(NdotV > 0.0f) || !(NdotL < 0.0f)) return zero;
    v3 H = normalize3(madd3(L, eta_p, V));
    if (dot3(N, H) < 0.0f) H = neg3(H);
    float VdotH = dot3(V, H), LdotH = dot3(L, H);
    if (!(VdotH > 0.0f) || !(LdotH < 0.0f)) return zero;
    float sq = VdotH + eta_p * LdotH;
    float den = sq * sq;
    if (den < EPSILON_) return zero;
    float alpha = m->Pr * m->Pr;
    float D = d_ggx(dot3(N, H), m->Pr);
    float G = g2_smith(NdotV, -NdotL, alpha);
    float e2 = eta_p * eta_p;
    float c = D * G * e2 * (-LdotH) * VdotH / (NdotV * (-NdotL) * den);
    v3 Fr = schlick(m->Ks, VdotH);
    float q = g1_smith(NdotV, alpha) * VdotH * D / NdotV * (e2 * (-LdotH) / den);
    v3 f = V3((1.0f - Fr.x) * c, (1.0f - Fr.y) * c, (1.0f - Fr.z) * c);
    if (!finite3(f) || is_nan(q) || is_inf(q)) return zero;
    *pdf = q;
    return f;
}
/* BRDF_v6.hlsl:50-70 -> (p_d, p_s); pt: the transmitted share of the diffuse part (extension; 0 when eta_p = 0) */
static inline void strategy_probs3(const matopt_t* m, v3 outgoing, v3 normal, uint32_t flags, float* pd, float* ps, float eta_p, float* pt) {
    *pt = 0.0f;
    if (flags & ORC_FLAG_LAMBERT_ONLY) { *pd = 1.0f; *ps = 0.0f; return; }
    v3 fr = schlick(m->Ks, dot3(normal, outgoing));
    float p_s = minf(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m->Pm);
    *ps = p_s; *pd = 1.0f - p_s;
    if (eta_p != 0.0f) { *pt = *pd * (1.0f - m->alpha); *pd = *pd * m->alpha; }      /* BRDF_v6.hlsl:28-29 `p_d *= alpha` */
}
static inline void strategy_probs(const matopt_t* m, v3 outgoing, v3 normal, uint32_t flags, float* pd, float* ps) { float pt; strategy_probs3(m, outgoing, normal, flags, pd, ps, 0.0f, &pt); }
/* BRDF_v6.hlsl:7-48; LAMBERT_ONLY consumes no random number */
static inline uint32_t select_strategy3(const matopt_t* m, v3 outgoing, v3 normal, uint32_t flags, uint32_t seed[2], float eta_p) {
    if (flags & ORC_FLAG_LAMBERT_ONLY) return 0;
    float r = rnd(seed);
    v3 fr = schlick(m->Ks, dot3(normal, outgoing));
    float p_s = minf(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m->Pm);
    if (r <= p_s) return m->Pr < 0.04f ? 0u : 1u;
    if (eta_p != 0.0f) {                                                         /* :41-47 with `p_d *= alpha` un-commented */
        float p_d = (1.0f - p_s) * m->alpha;
        return r <= p_s + p_d ? 0u : 3u;
    }
    return 0;
}
static inline uint32_t select_strategy(const matopt_t* m, v3 outgoing, v3 normal, uint32_t flags, uint32_t seed[2]) { return select_strategy3(m, outgoing, normal, flags, seed, 0.0f); }
/* mixture F = p_d f_lambert + p_s f_ggx and P likewise: Sampler_v6.hlsl:443-457, Path_Sampler_v6.hlsl:66-80; with eta_p != 0 a direction on the far
   side of the interface gets p_t f_t, p_t pdf_t alone */
static inline void bsdf_mixture3(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 normal, v3 L, v3 outgoing, v3* F, float* P, float* pd_o, float* ps_o, float eta_p) {
    const matopt_t* m = &c->mopt[mid];
    float pd, ps, pt; strategy_probs3(m, outgoing, normal, flags, &pd, &ps, eta_p, &pt);
    v3 f0 = lambert_eval(m); float q0 = lambert_pdf(normal, L);
    if (pd_o) *pd_o = pd;
    if (ps_o) *ps_o = ps;
    if (flags & ORC_FLAG_LAMBERT_ONLY) { *F = safe_mul3(pd, f0); *P = safe_mul1(pd, q0); return; }
    if (eta_p != 0.0f && dot3(normal, L) < 0.0f) {
        float q3; v3 f3 = btdf_eval(m, normal, L, outgoing, eta_p, &q3);
        *F = safe_mul3(pt, f3); *P = safe_mul1(pt, q3);
        return;
    }
    v3 f1 = ggx_eval(m, c->mats + (size_t)mid * 32, normal, L, outgoing);
    float q1 = ggx_pdf(m, normal, L, outgoing);
    *F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
    *P = safe_mul1(pd, q0) + safe_mul1(ps, q1);
}
static inline void bsdf_mixture(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 normal, v3 L, v3 outgoing, v3* F, float* P, float* pd_o, float* ps_o) {
    bsdf_mixture3(c, mid, flags, normal, L, outgoing, F, P, pd_o, ps_o, 0.0f);
}
/* Lambertian_v6.hlsl:2-38 */
static v3 sample_lambert(v3 normal, uint32_t seed[2]) {
    float u1 = rnd(seed), u2 = rnd(seed);
    float r = sqrtf(u1);
    float theta = TWO_PI_F * u2;
    float sn, cs; orc_sincos(theta, &sn, &cs);
    float x = r * cs, y = r * sn;
    float z = sqrtf(maxf(0.0f, 1.0f - x * x - y * y));
    v3 h = normal;
    v3 up = fabsf(normal.z) < 0.999f ? V3(0, 0, 1) : V3(1, 0, 0);
    v3 right = normalize3(cross3(up, h));
    v3 fwd = cross3(h, right);
    v3 s = lincomb3(right, x, fwd, y, h, z);
    s = normalize3(s);
    if (dot3(s, normal) < 0.0f) s = neg3(s);
    return s;
}
/* GGX_v6.hlsl:65-76 */
static inline void coord_system(v3 N, v3* T, v3* B) {
    if (fabsf(N.z) < 0.999f) *T = normalize3(cross3(V3(0, 0, 1), N));
    else *T = normalize3(cross3(V3(1, 0, 0), N));
    *B = cross3(N, *T);
}
/* GGX_v6.hlsl:93-169: the visible half vector H (:104-157) ... */
static v3 sample_ggx_h(const matopt_t* m, v3 outgoing, v3 normal, uint32_t seed[2], v3* Vout) {
    float alpha = m->Pr * m->Pr;
    v3 N = normalize3(normal), V = normalize3(outgoing), T1, T2;
    *Vout = V;
    coord_system(N, &T1, &T2);
    float vx = dot3(T1, V), vy = dot3(T2, V), vz = dot3(N, V);
    v3 Ve = normalize3(V3(alpha * vx, alpha * vy, vz));
    float lensq = Ve.x * Ve.x + Ve.y * Ve.y;
    v3 T1h;
    if (lensq > 0.0f) { float rs = 1.0f / sqrtf(lensq); T1h = V3(-Ve.y * rs, Ve.x * rs, 0.0f * rs); }
    else T1h = V3(1, 0, 0);
    v3 T2h = cross3(Ve, T1h);
    float U1 = rnd(seed), U2 = rnd(seed);
    float r = sqrtf(U1);
    float phi = 2.0f * PI_REF * U2;
    float sn, cs; orc_sincos(phi, &sn, &cs);
    float t1 = r * cs, t2 = r * sn;
    float s = 0.5f * (1.0f + Ve.z);
    t2 = (1.0f - s) * sqrtf(saturatef(1.0f - t1 * t1)) + s * t2;
    float w = sqrtf(saturatef(1.0f - t1 * t1 - t2 * t2));
    v3 Nh = V3(t1 * T1h.x + t2 * T2h.x + w * Ve.x, t1 * T1h.y + t2 * T2h.y + w * Ve.y, t1 * T1h.z + t2 * T2h.z + w * Ve.z);
    v3 Ne = normalize3(V3(alpha * Nh.x, alpha * Nh.y, maxf(0.0f, Nh.z)));
    return V3(Ne.x * T1.x + Ne.y * T2.x + Ne.z * N.x, Ne.x * T1.y + Ne.y * T2.y + Ne.z * N.y, Ne.x * T1.z + Ne.y * T2.z + Ne.z * N.z);
}
/* ... and the direction reflected about it (:159-165) */
static v3 sample_ggx(const matopt_t* m, v3 outgoing, v3 normal, uint32_t seed[2]) {
    v3 V; v3 H = sample_ggx_h(m, outgoing, normal, seed, &V);
    v3 I = neg3(V);                                   /* reflect(-V, H) = I - 2 dot(H,I) H */
    float k = 2.0f * dot3(H, I);
    v3 smp = V3(I.x - k * H.x, I.y - k * H.y, I.z - k * H.z);
    if (dot3(smp, normal) < 0.0f) smp = neg3(smp);    /* :164-165 flipped, not rejected */
    return smp;
}
/* EXTENSION, strategy 3: refract about the same visible half vector; total internal reflection ends the path (zero vector) */
static v3 sample_btdf(const matopt_t* m, v3 outgoing, v3 normal, float eta_p, uint32_t seed[2]) {
    v3 V; v3 H = sample_ggx_h(m, outgoing, normal, seed, &V);
    float eta = 1.0f / eta_p;
    float c = dot3(V, H);
    float s2 = eta * eta * (1.0f - c * c);
    if (!(s2 < 1.0f)) return V3(0, 0, 0);
    float k = eta * c - sqrtf(1.0f - s2);
    return normalize3(V3(k * H.x - eta * V.x, k * H.y - eta * V.y, k * H.z - eta * V.z));
}
static inline v3 sample_bsdf3(const orc_ctx* c, uint32_t mid, uint32_t strategy, v3 outgoing, v3 normal, uint32_t seed[2], float eta_p) {
    if (strategy == 3) return sample_btdf(&c->mopt[mid], outgoing, normal, eta_p, seed);
    return strategy == 1 ? sample_ggx(&c->mopt[mid], outgoing, normal, seed) : sample_lambert(normal, seed);   /* BRDF_v6.hlsl:74-88 */
}
static inline v3 sample_bsdf(const orc_ctx* c, uint32_t mid, uint32_t strategy, v3 outgoing, v3 normal, uint32_t seed[2]) { return sample_bsdf3(c, mid, strategy, outgoing, normal, seed, 0.0f); }

int orc_bsdf_eval(orc_ctx* c, uint32_t mid, uint32_t flags, const float* in9, uint32_t n, float* out8) {
    if (mid >= c->nmat) return -1;
    for (uint32_t i = 0; i < n; i++) {
        const float* q = in9 + (size_t)i * 9; float* o = out8 + (size_t)i * 8;
        v3 F; float P, pd, ps;
        v3 nrm = V3(q[0], q[1], q[2]), wo = V3(q[3], q[4], q[5]);
        float eta_p = transmission_eta(&c->mopt[mid], flags, wo, &nrm);
        bsdf_mixture3(c, mid, flags, nrm, V3(q[6], q[7], q[8]), wo, &F, &P, &pd, &ps, eta_p);
        o[0] = F.x; o[1] = F.y; o[2] = F.z; o[3] = P; o[4] = pd; o[5] = ps; o[6] = eta_p; o[7] = 0.0f;
    }
    return 0;
}
int orc_bsdf_sample(orc_ctx* c, uint32_t mid, uint32_t flags, const float* in8, uint32_t n, float* out8) {
    if (mid >= c->nmat) return -1;
    for (uint32_t i = 0; i < n; i++) {
        const float* q = in8 + (size_t)i * 8; float* o = out8 + (size_t)i * 8;
        uint32_t seed[2]; memcpy(seed, &q[6], 8);
        v3 nrm = V3(q[0], q[1], q[2]), wo = V3(q[3], q[4], q[5]);
        float eta_p = transmission_eta(&c->mopt[mid], flags, wo, &nrm);
        uint32_t st = select_strategy3(&c->mopt[mid], wo, nrm, flags, seed, eta_p);
        v3 wi = sample_bsdf3(c, mid, st, wo, nrm, seed, eta_p);
        o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; memcpy(&o[3], &st, 4); memcpy(&o[4], seed, 8); o[6] = o[7] = 0.0f;
    }
    return 0;
}

/* ---- primary rays: RayGen_v6_pass1.hlsl:51-95 ---- */
static inline void primary_ray(const orc_ctx* c, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float jx, float jy, v3* o, v3* d) {
    float dx = (((float)x + jx) / (float)W) * 2.0f - 1.0f;
    float dy = (((float)y + jy) / (float)H) * 2.0f - 1.0f;
    const float* P = c->projI; const float* Vi = c->viewI;
    float ndy = -dy;
    v3 tg = V3(P[0] * dx + P[4] * ndy + P[8] + P[12], P[1] * dx + P[5] * ndy + P[9] + P[13], P[2] * dx + P[6] * ndy + P[10] + P[14]);
    *d = normalize3(xform_dir(Vi, tg));
    *o = V3(Vi[12], Vi[13], Vi[14]);
}
static inline int owns_pixel(const orc_params* p, uint32_t x, uint32_t y) {
    if (p->shard_count <= 1) return 1;
    uint32_t ts = p->tile_size ? p->tile_size : 64;
    uint32_t tiles_x = (p->width + ts - 1) / ts;
    uint32_t t = (y / ts) * tiles_x + (x / ts);
    return (t % p->shard_count) == p->shard_rank;
}
int orc_primary_rays(orc_ctx* c, const orc_params* p, uint32_t s, float* rays8) {
    for (uint32_t y = 0; y < p->height; y++) for (uint32_t x = 0; x < p->width; x++) {
        uint32_t seed[2]; orc_seed_init(x, y, s, p->frame_seed, seed);
        float jx = 0.0f, jy = 0.0f;
        if (p->flags & ORC_FLAG_JITTER) { jx = rnd(seed); jy = rnd(seed); }
        v3 o, d; primary_ray(c, p->width, p->height, x, y, jx, jy, &o, &d);
        float* r = rays8 + ((size_t)y * p->width + x) * 8;
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = 0.0001f; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = 10000.0f;
    }
    return 0;
}

/* ---- one path sample: loop of RayGen.hlsl:99-133 with the v6 leaf math ---- */
static v3 trace_path(const orc_ctx* c, const orc_params* p, uint32_t x, uint32_t y, uint32_t s, uint64_t cnt[3]) {
    uint32_t seed[2]; orc_seed_init(x, y, s, p->frame_seed, seed);
    const uint32_t flags = p->flags, nee = c->nlights ? p->nee_samples : 0;
    float jx = 0.0f, jy = 0.0f;
    if (flags & ORC_FLAG_JITTER) { jx = rnd(seed); jy = rnd(seed); }          /* RayGen.hlsl:84-85 */
    v3 origin, dir; primary_ray(c, p->width, p->height, x, y, jx, jy, &origin, &dir);
    float tmin = 0.0001f;                                                      /* pass1:95 */
    v3 thr = V3(1, 1, 1), rad = V3(0, 0, 0);
    float prev_pdf = 1.0f;
    for (uint32_t b = 0; b < p->max_bounces; b++) {
        hit_t h = closest_bvh(c, origin, dir, tmin, 10000.0f);
        cnt[b == 0 ? 0 : 1]++;
        if (h.prim == MISS_PRIM) break;                                        /* Miss.hlsl:3-11: black, terminate */
        surf_t sf = surface(c, origin, dir, h);
        if (sf.mat >= c->nmat) break;
        const matopt_t* m = &c->mopt[sf.mat];
        if (m->Ke_len > 0.0f) {                                                /* Hit.hlsl:126, Sampler_v6.hlsl:457 */
            if (b == 0) rad = add3(rad, m->Ke);                                /* Hit.hlsl:128-131 */
            else {
                float mi = 1.0f;
                if (nee) {                                                     /* Sampler_v6.hlsl:459-465, Path_Sampler_v6.hlsl:241 */
                    v3 Lv = sub3(sf.pos, origin);
                    float dist = length3(Lv), dist2 = dist * dist;
                    float cos_t = fabsf(dot3(sf.normal, neg3(dir)));           /* DEVIATION: abs (two-sided lights, as NEE does) */
                    float pdf_light = (((m->Ke.x + m->Ke.y + m->Ke.z) / 3.0f) / c->total_weight) * dist2 / maxf(cos_t, EPSILON_);
                    mi = prev_pdf / ((float)nee * pdf_light + prev_pdf);
                    if (prev_pdf < 0.0f) mi = 1.0f;                            /* (extension) the ray came through a transmission lobe: NEE never samples through an interface */
                }
                v3 e = V3(m->Ke.x * thr.x * mi, m->Ke.y * thr.y * mi, m->Ke.z * thr.z * mi);   /* Hit.hlsl:173 */
                if (finite3(e)) rad = add3(rad, e);
            }
            break;
        }
        v3 outgoing = neg3(dir);
        v3 normal = sf.normal;
        const float eta_p = transmission_eta(m, flags, outgoing, &normal);    /* (extension) hits from behind a dielectric flip the shading normal */
        /* ---- NEE: SampleLightNEE_GI, Sampler_v6.hlsl:508-647, with its visibility ray enabled ---- */
        for (uint32_t j = 0; j < nee; j++) {
            float rv = rnd(seed);
            int left = 0, right = (int)c->nlights - 1, sel = 0;
            while (left <= right) {                                            /* :523-537 */
                int mid = left + (right - left) / 2;
                if (rv < c->lights[(size_t)mid * 20 + 3]) { sel = mid; right = mid - 1; } else left = mid + 1;
            }
            const float* lt = c->lights + (size_t)sel * 20;
            uint32_t li; memcpy(&li, &lt[7], 4);
            const float* M = c->insts[li].o2w;
            v3 xv = xform_point(M, V3(lt[0], lt[1], lt[2])), yv = xform_point(M, V3(lt[4], lt[5], lt[6])), zv = xform_point(M, V3(lt[8], lt[9], lt[10]));
            float xi1 = rnd(seed), xi2 = rnd(seed);
            if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
            float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
            v3 sp = lincomb3(xv, u, yv, v, zv, w);
            v3 Lv = sub3(sp, sf.pos);
            float dist2 = dot3(Lv, Lv);
            float dist = sqrtf(maxf(dist2, EPSILON_));
            v3 Ln = normalize3(Lv);
            v3 cl = cross3(sub3(yv, xv), sub3(zv, xv));
            v3 nl = normalize3(cl);
            if (dot3(nl, neg3(Ln)) < 0.0f) nl = neg3(nl);
            float area_l = fabsf(length3(cl) * 0.5f);
            float pdf_l = lt[11] / maxf(area_l, EPSILON_);
            float cos_x = dot3(normal, Ln);                                    /* DEVIATION: clamped like ReconnectDI (:117), v6 GI uses abs (:579) */
            float cos_y = fabsf(dot3(nl, neg3(Ln)));
            if (cos_x < EPSILON_ || cos_y < EPSILON_) continue;                /* :580-585; no shadow ray is traced */
            float pdf_light = maxf(EPSILON_, pdf_l) * dist2 / cos_y;           /* :629-630 */
            v3 F; float P; bsdf_mixture3(c, sf.mat, flags, normal, Ln, outgoing, &F, &P, NULL, NULL, eta_p);
            float mi = pdf_light / ((float)nee * pdf_light + P);               /* Path_Sampler_v6.hlsl:164 */
            float g = cos_x / pdf_light * mi;
            v3 con = V3(lt[12] * (thr.x * F.x) * g, lt[13] * (thr.y * F.y) * g, lt[14] * (thr.z * F.z) * g);
            if (!finite3(con) || (con.x == 0.0f && con.y == 0.0f && con.z == 0.0f)) continue;
            /* visibility: Sampler_v6.hlsl:616-628 */
            v3 so = madd3(normalize3(normal), S_BIAS, sf.pos);
            float smax = maxf(S_BIAS, dist - S_BIAS * 5.0f);
            cnt[2]++;
            if (!any_bvh(c, so, Ln, 0.5f * S_BIAS, smax)) rad = add3(rad, con);
        }
        if (b + 1 == p->max_bounces) break;
        /* ---- BSDF sampling: Path_Sampler_v6.hlsl:205-229, Sampler_v6.hlsl:423-457,482-497 ---- */
        uint32_t st = select_strategy3(m, outgoing, normal, flags, seed, eta_p);
        v3 smp = sample_bsdf3(c, sf.mat, st, outgoing, normal, seed, eta_p);
        if (st == 3 && smp.x == 0.0f && smp.y == 0.0f && smp.z == 0.0f) break;    /* (extension) total internal reflection ends the path */
        v3 F; float P; bsdf_mixture3(c, sf.mat, flags, normal, smp, outgoing, &F, &P, NULL, NULL, eta_p);
        float NdotL = dot3(normal, smp);                                       /* unclamped: Sampler_v6.hlsl:455 */
        if (st == 3) NdotL = fabsf(NdotL);                                     /* (extension) the transmitted direction lies on the far side */
        if (!(P > 0.0f)) break;
        float wgt = NdotL / P;                                                 /* Hit.hlsl:366 */
        thr = V3(thr.x * (F.x * wgt), thr.y * (F.y * wgt), thr.z * (F.z * wgt));
        if (!finite3(thr) || (thr.x == 0.0f && thr.y == 0.0f && thr.z == 0.0f)) break;
        prev_pdf = st == 3 ? -P : P;                                           /* Hit.hlsl:369; the sign carries "transmitted" to the next emissive hit */
        if (b > p->rr_start) {                                                 /* RayGen.hlsl:118-130 */
            float mx = maxf(thr.x, maxf(thr.y, thr.z));
            float q = minf(maxf(mx, 0.05f), 1.0f);
            float r = rnd(seed);
            if (r > q) break;
            float iq = 1.0f / q;
            thr = scale3(thr, iq);
        }
        origin = sf.pos; dir = smp; tmin = S_BIAS;                             /* Sampler_v6.hlsl:224-227 */
    }
    return rad;
}

int orc_render(orc_ctx* c, const orc_params* p, float* accum, uint64_t ray_counts[3]) {
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    int nth = c->nthreads;
#ifdef _OPENMP
    if (nth <= 0) nth = omp_get_max_threads();
#else
    nth = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth) reduction(+ : c0, c1, c2)
    for (int64_t y = 0; y < (int64_t)p->height; y++) {
        uint64_t cnt[3] = {0, 0, 0};
        for (uint32_t x = 0; x < p->width; x++) {
            if (!owns_pixel(p, x, (uint32_t)y)) continue;
            float* a = accum + ((size_t)y * p->width + x) * 4;
            for (uint32_t s = 0; s < p->spp; s++) {
                v3 r = trace_path(c, p, x, (uint32_t)y, p->sample_base + s, cnt);
                if (finite3(r)) { a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += 1.0f; }   /* pass3:383-405 */
            }
        }
        c0 += cnt[0]; c1 += cnt[1]; c2 += cnt[2];
    }
    if (ray_counts) { ray_counts[0] = c0; ray_counts[1] = c1; ray_counts[2] = c2; }
    return 0;
}

/* =====================================================================================================
 * The v6 PASS-1 estimator, restated literally: RayGen_v6_pass1.hlsl:48-190 = primary hit, RIS for direct
 * light (SampleRIS, Sampler_v6.hlsl:653-736, a21), its visibility ray, and SamplePathSimple
 * (Path_Sampler_v6.hlsl:3-286, a22) with its quirks kept: abs cosines and unshadowed NEE inside the loop,
 * two SelectSamplingStrategy draws per bounce, reservoir updates that consume random numbers, half-precision
 * L2 / E3 / L1, the `pdf_light = 1` initial value.  Outputs are the reference's own buffers:
 * Reservoir_DI / Reservoir_GI (40 B) and SampleData (60 B) at MapPixelID order (Reservoir_v6.hlsl:1-33,
 * Common_v6.hlsl:173-198); the accumulated radiance is sdata.debug (pass1:173-174), or L1 for emissive
 * primary hits (pass3:457-462).
 * DEVIATIONS: a ray that misses ends the estimator at that point with zero contribution (v6 reads an
 * uninitialised payload); uint(time) := frame_seed and the literal sample id 1 := s.
 * ===================================================================================================== */
static inline uint16_t half_bits(float x) {          /* x is already half-representable after orc_half_round */
    float r = orc_half_round(x);
    uint32_t u = f2u(r), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((a & 0x007FFFFFu) ? 0x200u : 0u));
    if (a == 0) return (uint16_t)sign;
    int e = (int)(a >> 23) - 127;
    if (e < -14) { float q = u2f(a) * 16777216.0f; return (uint16_t)(sign | (uint32_t)q); }    /* subnormal half */
    return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | ((a >> 13) & 0x3FFu));
}
static inline v3 half3(v3 a) { return V3(orc_half_round(a.x), orc_half_round(a.y), orc_half_round(a.z)); }

typedef struct { v3 x2; float w_sum; v3 n2; float W; v3 L2; uint32_t M; } res_t;     /* L2 holds half-rounded values */

/* raw lobes for a direction L (= -incidence): f0,f1 / q0,q1 / (p_d,p_s) */
static inline void lobes(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 normal, v3 L, v3 outgoing_eval, v3 outgoing_pdf,
                         v3* f0, v3* f1, float* q0, float* q1, float* pd, float* ps) {
    const matopt_t* m = &c->mopt[mid];
    strategy_probs(m, outgoing_eval, normal, flags, pd, ps);
    *f0 = lambert_eval(m); *q0 = lambert_pdf(normal, L);
    if (flags & ORC_FLAG_LAMBERT_ONLY) { *f1 = V3(0, 0, 0); *q1 = 0.0f; }
    else { *f1 = ggx_eval(m, c->mats + (size_t)mid * 32, normal, L, outgoing_eval); *q1 = ggx_pdf(m, normal, L, outgoing_pdf); }
}
/* light triangle by CDF: Sampler_v6.hlsl:291-307 */
static inline const float* pick_light(const orc_ctx* c, float rv) {
    int left = 0, right = (int)c->nlights - 1, sel = 0;
    while (left <= right) { int mid = left + (right - left) / 2; if (rv < c->lights[(size_t)mid * 20 + 3]) { sel = mid; right = mid - 1; } else left = mid + 1; }
    return c->lights + (size_t)sel * 20;
}
typedef struct { v3 sp, Ln, nl; float dist2, dist, pdf_l; const float* lt; } lsample_t;
/* shared front part of SampleLightNEE / SampleLightNEE_GI: :291-331 / :523-573 */
static inline lsample_t light_point(const orc_ctx* c, v3 origin, uint32_t seed[2]) {
    lsample_t r;
    r.lt = pick_light(c, rnd(seed));
    uint32_t li; memcpy(&li, &r.lt[7], 4);
    const float* M = c->insts[li].o2w;
    v3 xv = xform_point(M, V3(r.lt[0], r.lt[1], r.lt[2])), yv = xform_point(M, V3(r.lt[4], r.lt[5], r.lt[6])), zv = xform_point(M, V3(r.lt[8], r.lt[9], r.lt[10]));
    float xi1 = rnd(seed), xi2 = rnd(seed);
    if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
    float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
    r.sp = V3(u * xv.x + v * yv.x + w * zv.x, u * xv.y + v * yv.y + w * zv.y, u * xv.z + v * yv.z + w * zv.z);
    v3 Lv = sub3(r.sp, origin);
    r.dist2 = dot3(Lv, Lv); r.dist = sqrtf(maxf(r.dist2, EPSILON_)); r.Ln = normalize3(Lv);
    v3 cl = cross3(sub3(yv, xv), sub3(zv, xv));
    r.nl = normalize3(cl);
    if (dot3(r.nl, neg3(r.Ln)) < 0.0f) r.nl = neg3(r.nl);
    r.pdf_l = r.lt[11] / maxf(fabsf(length3(cl) * 0.5f), EPSILON_);
    return r;
}
/* UpdateReservoir / UpdateReservoir_GI: Reservoir_v6.hlsl:36-80 */
static inline int res_update(res_t* r, float wi, v3 x, v3 n, v3 L, uint32_t seed[2]) {
    r->w_sum += wi;
    if (rnd(seed) < wi / r->w_sum) { r->x2 = x; r->n2 = n; r->L2 = half3(L); return 1; }
    return 0;
}
/* ReconnectDI: Sampler_v6.hlsl:106-131 */
static v3 reconnect_di(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 x1, v3 n1, v3 x2, v3 n2, v3 L, v3 outgoing) {
    v3 dir = sub3(x2, x1);
    float dist = length3(dir);
    float cos1 = maxf(0.0f, dot3(n1, normalize3(dir)));
    if (dot3(n2, normalize3(neg3(dir))) < 0.0f) n2 = neg3(n2);
    float cos2 = maxf(0.0f, dot3(n2, normalize3(neg3(dir))));
    v3 f0, f1; float q0, q1, pd, ps;
    lobes(c, mid, flags, n1, normalize3(dir), normalize3(outgoing), normalize3(outgoing), &f0, &f1, &q0, &q1, &pd, &ps);
    v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
    float d2 = dist * dist;
    return V3(F.x * L.x * cos1 * cos2 / d2, F.y * L.y * cos1 * cos2 / d2, F.z * L.z * cos1 * cos2 / d2);
}
/* VisibilityCheck: Sampler_v6.hlsl:86-104 */
static float visibility(const orc_ctx* c, v3 x1, v3 n1, v3 dir, float dist, uint64_t cnt[3]) {
    v3 o = add3(x1, scale3(normalize3(n1), S_BIAS));
    cnt[2]++;
    return any_bvh(c, o, dir, 0.0f, maxf(dist - 10.0f * S_BIAS, 2.0f * S_BIAS)) ? 0.0f : 1.0f;
}
/* MaterialOptimized of a hit; returns 0 for a miss */
static inline int hit_material(const orc_ctx* c, v3 o, v3 d, float tmin, surf_t* sf, uint64_t cnt[3], int primary) {
    hit_t h = closest_bvh(c, o, d, tmin, 10000.0f);
    cnt[primary ? 0 : 1]++;
    if (h.prim == MISS_PRIM) return 0;
    *sf = surface(c, o, d, h);
    return sf->mat < c->nmat;
}
static inline float full_ke_len(const orc_ctx* c, uint32_t mid) { const float* m = c->mats + (size_t)mid * 32; return length3(V3(m[8], m[9], m[10])); }

/* SampleRIS: Sampler_v6.hlsl:653-736 with SampleLightNEE (:273-396) and SampleLightBSDF (:199-271) inlined */
static void sample_ris(const orc_ctx* c, uint32_t M1, uint32_t M2, uint32_t flags, v3 outgoing, res_t* rs, const surf_t* pay, uint32_t seed[2], uint64_t cnt[3]) {
    const matopt_t* m = &c->mopt[pay->mat];
    uint32_t strategy = select_strategy(m, outgoing, pay->normal, flags, seed);
    v3 origin = pay->pos, normal = pay->normal;
    for (uint32_t i = 0; i < M1 && c->nlights; i++) {
        lsample_t ls = light_point(c, origin, seed);
        float cos_x = dot3(normal, ls.Ln), cos_y = dot3(ls.nl, neg3(ls.Ln));
        float G = maxf(cos_y * cos_x / ls.dist2, EPSILON_);                                  /* :347 */
        v3 em = V3(ls.lt[12], ls.lt[13], ls.lt[14]);
        v3 f0, f1; float q0, q1, pd, ps;
        lobes(c, pay->mat, flags, normal, ls.Ln, normalize3(outgoing), normalize3(outgoing), &f0, &f1, &q0, &q1, &pd, &ps);
        v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
        float P = safe_mul1(pd, q0 * cos_y / ls.dist2) + safe_mul1(ps, q1 * cos_y / ls.dist2);   /* :357-366 area measure */
        float p_hat = length3(V3(em.x * F.x * G * 1.0f, em.y * F.y * G * 1.0f, em.z * F.z * G * 1.0f));   /* :383 */
        float pdf_light = maxf(EPSILON_, ls.pdf_l);
        float mi = pdf_light / ((float)M1 * pdf_light + (float)M2 * P);
        float wi = mi * p_hat / pdf_light;
        if (p_hat > 0.0f) res_update(rs, wi, ls.sp, ls.nl, em, seed);
    }
    for (uint32_t j = 0; j < M2; j++) {
        float pdf_light = 0.0f, pdf_bsdf = 0.0f, p_hat = 0.0f;
        v3 em = V3(0, 0, 0), x2 = V3(0, 0, 0), n2 = V3(0, 0, 0);
        v3 smp = sample_bsdf(c, pay->mat, strategy, outgoing, normal, seed);
        surf_t h2;
        if (hit_material(c, origin, smp, S_BIAS, &h2, cnt, 0)) {
            const float* mk = c->mats + (size_t)h2.mat * 32;
            float Ke = mk[8] + mk[9] + mk[10];
            em = V3(mk[8], mk[9], mk[10]); x2 = h2.pos; n2 = h2.normal;
            if (Ke > EPSILON_ && c->nlights) {
                float dist = length3(sub3(h2.pos, origin)), dist2 = dist * dist;
                float cos_t = dot3(h2.normal, neg3(smp));
                pdf_light = (Ke / 3.0f) / c->total_weight;
                v3 f0, f1; float q0, q1, pd, ps;
                lobes(c, pay->mat, flags, normal, smp, normalize3(outgoing), outgoing, &f0, &f1, &q0, &q1, &pd, &ps);
                v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
                pdf_bsdf = safe_mul1(pd, q0 * cos_t / dist2) + safe_mul1(ps, q1 * cos_t / dist2);
                float ndot = dot3(normal, smp);
                p_hat = length3(V3(F.x * em.x * ndot * cos_t / dist2, F.y * em.y * ndot * cos_t / dist2, F.z * em.z * ndot * cos_t / dist2));
            }
        }
        float mi = pdf_bsdf / ((float)M1 * pdf_light + (float)M2 * pdf_bsdf);
        float wi = mi * p_hat / pdf_bsdf;
        if (p_hat > 0.0f) res_update(rs, wi, x2, n2, em, seed);
    }
    rs->M = 1;
}

/* SamplePathSimple: Path_Sampler_v6.hlsl:3-286 with SampleLightNEE_GI (Sampler_v6.hlsl:508-647) and
   SampleLightBSDF_GI (:399-505) inlined, visibility flags as the reference passes them (false) */
static v3 sample_path_simple(const orc_ctx* c, const orc_params* p, res_t* rs, v3 init_point, v3 init_normal, v3 init_outgoing, uint32_t init_mat,
                             uint32_t seed[2], uint64_t cnt[3]) {
    const uint32_t flags = p->flags, nee = c->nlights ? p->nee_samples : 0;
    v3 acc_f = V3(1, 1, 1), acc_f_rec = V3(1, 1, 1), acc_L = V3(0, 0, 0);
    float acc_pdf = 1.0f;
    v3 x1s = V3(0, 0, 0), x2s = V3(0, 0, 0);
    v3 origin = init_point, normal = init_normal, outgoing = normalize3(init_outgoing);
    uint32_t mat = init_mat;
    {   /* 1) first BSDF bounce, :37-99 */
        uint32_t st = select_strategy(&c->mopt[mat], outgoing, normal, flags, seed);
        v3 smp = sample_bsdf(c, mat, st, outgoing, normal, seed);
        surf_t h;
        if (!hit_material(c, origin, smp, S_BIAS, &h, cnt, 0)) return V3(0, 0, 0);
        if (full_ke_len(c, h.mat) > 0.0f) return V3(0, 0, 0);                                  /* :55-60 */
        v3 incoming = normalize3(neg3(smp));
        v3 f0, f1; float q0, q1, pd, ps;
        lobes(c, mat, flags, normal, neg3(incoming), outgoing, outgoing, &f0, &f1, &q0, &q1, &pd, &ps);
        v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
        float P = safe_mul1(pd, q0) + safe_mul1(ps, q1);
        float NdotL = dot3(normal, smp);
        acc_pdf *= P;
        acc_f = V3(acc_f.x * (F.x * NdotL), acc_f.y * (F.y * NdotL), acc_f.z * (F.z * NdotL));
        outgoing = incoming; mat = h.mat; normal = h.normal; origin = h.pos;
    }
    const v3 xn = origin, nn = normalize3(normal);                                              /* :104-106 */
    for (uint32_t i = 0; i < p->max_bounces; i++) {
        (void)select_strategy(&c->mopt[mat], outgoing, normal, flags, seed);                     /* :118 (its result only feeds dead code) */
        for (uint32_t j = 0; j < nee; j++) {                                                    /* 3a) :123-195 */
            lsample_t ls = light_point(c, origin, seed);
            float cos_x = fabsf(dot3(normal, ls.Ln)); if (cos_x < EPSILON_) cos_x = 0.0f;       /* :579-581 */
            float cos_y = fabsf(dot3(ls.nl, neg3(ls.Ln))); if (cos_y < EPSILON_) cos_y = 0.0f;
            v3 em = V3(ls.lt[12], ls.lt[13], ls.lt[14]);
            v3 f0, f1; float q0, q1, pd, ps;
            lobes(c, mat, flags, normal, ls.Ln, normalize3(outgoing), normalize3(outgoing), &f0, &f1, &q0, &q1, &pd, &ps);
            v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
            float pdf_bsdf = safe_mul1(pd, q0) + safe_mul1(ps, q1);
            float pdf_light = 1.0f;                                                             /* caller's initial value, :125 */
            if (cos_y > 0.0f) pdf_light = maxf(EPSILON_, ls.pdf_l) * ls.dist2 / cos_y;          /* :629-630 */
            float a_pdf = acc_pdf * pdf_light;
            v3 thr = V3(F.x * cos_x * 1.0f, F.y * cos_x * 1.0f, F.z * cos_x * 1.0f);            /* brdf_light * G * V, V = 1 */
            v3 a_l = mul3(acc_f, thr);
            v3 contribution = a_pdf > 0.0f ? V3(em.x * a_l.x / a_pdf, em.y * a_l.y / a_pdf, em.z * a_l.z / a_pdf) : V3(0, 0, 0);
            float mi = pdf_light / ((float)nee * pdf_light + pdf_bsdf);                         /* :164 */
            v3 E_rec = V3(acc_f_rec.x * mi * em.x * thr.x, acc_f_rec.y * mi * em.y * thr.y, acc_f_rec.z * mi * em.z * thr.z);
            v3 E_path = scale3(contribution, mi);
            float wi = length3(E_path);
            acc_L = add3(acc_L, E_path);
            if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
            if (res_update(rs, wi, xn, normalize3(nn), E_rec, seed)) { x1s = add3(origin, scale3(normalize3(normal), S_BIAS)); x2s = ls.sp; }
        }
        uint32_t st = select_strategy(&c->mopt[mat], outgoing, normal, flags, seed);            /* :205 */
        v3 smp = sample_bsdf(c, mat, st, outgoing, normal, seed);                               /* SampleLightBSDF_GI :423-434 */
        surf_t h;
        if (!hit_material(c, origin, smp, S_BIAS, &h, cnt, 0)) break;
        v3 f0, f1; float q0, q1, pd, ps;
        lobes(c, mat, flags, normal, smp, normalize3(outgoing), outgoing, &f0, &f1, &q0, &q1, &pd, &ps);
        v3 F = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
        float pdf_bsdf = safe_mul1(pd, q0) + safe_mul1(ps, q1);
        float NdotL = dot3(normal, smp);
        const matopt_t* mk = &c->mopt[h.mat];
        v3 thr = V3(F.x * NdotL, F.y * NdotL, F.z * NdotL);
        acc_pdf *= pdf_bsdf;
        acc_f = mul3(acc_f, thr);
        acc_f_rec = mul3(acc_f_rec, thr);                                                       /* :233 */
        if (mk->Ke_len > 0.0f) {                                                                /* light hit: :457-479 */
            float dist = length3(sub3(h.pos, origin)), dist2 = dist * dist;
            float cos_t = dot3(h.normal, neg3(smp));
            float pdf_light = c->nlights ? (((mk->Ke.x + mk->Ke.y + mk->Ke.z) / 3.0f) / c->total_weight) * dist2 / cos_t : 0.0f;
            v3 contribution = V3(mk->Ke.x * acc_f.x / acc_pdf, mk->Ke.y * acc_f.y / acc_pdf, mk->Ke.z * acc_f.z / acc_pdf);
            if (length3(contribution) > 0.0f) {                                                 /* :236-262 */
                float mi = pdf_bsdf / ((float)nee * pdf_light + pdf_bsdf);
                v3 E_rec = V3(acc_f_rec.x * mi * mk->Ke.x, acc_f_rec.y * mi * mk->Ke.y, acc_f_rec.z * mi * mk->Ke.z);
                v3 E_path = scale3(contribution, mi);
                float wi = length3(E_path);
                acc_L = add3(acc_L, E_path);
                if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
                res_update(rs, wi, xn, normalize3(nn), E_rec, seed);
                break;
            }
        }
        origin = h.pos; mat = h.mat; outgoing = neg3(smp); normal = h.normal;                   /* :263-269 */
    }
    if (nee > 0 && length3(sub3(x2s, x1s)) > EPSILON_) {                                        /* :271-283 */
        v3 dv = sub3(x2s, x1s);
        cnt[2]++;
        if (any_bvh(c, x1s, normalize3(dv), 0.5f * S_BIAS, maxf(S_BIAS, length3(dv) - S_BIAS * 5.0f))) rs->w_sum *= 0.0f;
        else rs->w_sum *= 1.0f;
    }
    return acc_L;
}

/* MapPixelID: Common_v6.hlsl:173-198 */
static inline uint32_t map_pixel_id(uint32_t w, uint32_t x, uint32_t y) {
    const uint32_t ts = 4, tcx = (w + ts - 1) / ts;
    return ((y / ts) * tcx + (x / ts)) * (ts * ts) + (y % ts) * ts + (x % ts);
}
uint32_t orc_map_pixel_id(uint32_t w, uint32_t x, uint32_t y) { return map_pixel_id(w, x, y); }
size_t orc_pass1_slots(uint32_t w, uint32_t h) { return (size_t)((w + 3) / 4) * ((h + 3) / 4) * 16; }

static void store_res(uint8_t* dst, const res_t* r) {             /* 40 bytes: Reservoir_v6.hlsl:16-29 */
    float f[8] = {r->x2.x, r->x2.y, r->x2.z, r->w_sum, r->n2.x, r->n2.y, r->n2.z, r->W};
    uint16_t h[4] = {half_bits(r->L2.x), half_bits(r->L2.y), half_bits(r->L2.z), (uint16_t)r->M};
    memcpy(dst, f, 32); memcpy(dst + 32, h, 8);
}

/* one pass-1 sample for every owned pixel; the buffers receive the state of the LAST sample of the call */
int orc_render_v6_pass1(orc_ctx* c, const orc_params* p, float* accum, void* res_di40, void* res_gi40, void* sample60, uint64_t ray_counts[3]) {
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    int nth = c->nthreads;
#ifdef _OPENMP
    if (nth <= 0) nth = omp_get_max_threads();
#else
    nth = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth) reduction(+ : c0, c1, c2)
    for (int64_t y = 0; y < (int64_t)p->height; y++) {
        uint64_t cnt[3] = {0, 0, 0};
        for (uint32_t x = 0; x < p->width; x++) {
            if (!owns_pixel(p, x, (uint32_t)y)) continue;
            for (uint32_t si = 0; si < p->spp; si++) {
                uint32_t seed[2]; orc_seed_init(x, (uint32_t)y, p->sample_base + si, p->frame_seed, seed);
                v3 origin, dir; primary_ray(c, p->width, p->height, x, (uint32_t)y, 0.0f, 0.0f, &origin, &dir);   /* jitter = 0, pass1:80-82 */
                res_t rdi; memset(&rdi, 0, sizeof(rdi)); res_t rgi; memset(&rgi, 0, sizeof(rgi));
                v3 x1 = V3(0, 0, 0), n1 = V3(0, 0, 0), ov = V3(0, 0, 0), debug = V3(0, 0, 0), L1 = V3(0, 0, 0);
                uint32_t mID = 0xFFFFFFFEu, objID = 0;
                surf_t pay;
                v3 out = V3(0, 0, 0);
                if (hit_material(c, origin, dir, 0.0001f, &pay, cnt, 1)) {
                    mID = pay.mat; objID = pay.inst;
                    L1 = c->mopt[mID].Ke;
                    if (!(full_ke_len(c, mID) > 0.0f)) {                                         /* performSampling, pass1:102-106 */
                        v3 outgoing = neg3(dir);
                        sample_ris(c, c->nlights ? p->nee_samples : 0, 1, p->flags, outgoing, &rdi, &pay, seed, cnt);   /* nee_samples_DI, bsdf_samples_DI */
                        x1 = pay.pos; n1 = normalize3(pay.normal); ov = outgoing;
                        float f_g = length3(reconnect_di(c, mID, p->flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov));
                        v3 dv = sub3(rdi.x2, x1);
                        float vis = visibility(c, x1, n1, normalize3(dv), length3(dv), cnt);     /* GetP_Hat(..., true), :163-171 */
                        float p_hat = f_g * vis;
                        rdi.W = p_hat > EPSILON_ ? rdi.w_sum / p_hat : 0.0f;                     /* GetW :183-188 */
                        debug = sample_path_simple(c, p, &rgi, pay.pos, pay.normal, outgoing, mID, seed, cnt);
                        v3 rc = reconnect_di(c, mID, p->flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov);
                        debug = add3(debug, scale3(rc, rdi.W));                                  /* pass1:174 */
                        {   /* ReconnectGI + GetW_GI: Sampler_v6.hlsl:134-160, 173-181; pass1:176-180 */
                            v3 dg = sub3(rgi.x2, x1);
                            float cos1 = fabsf(dot3(n1, normalize3(dg)));
                            v3 f0, f1; float q0, q1, pd, ps;
                            lobes(c, mID, p->flags, n1, normalize3(dg), normalize3(ov), normalize3(ov), &f0, &f1, &q0, &q1, &pd, &ps);
                            v3 Fx = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
                            v3 fr = V3(Fx.x * cos1 * rgi.L2.x, Fx.y * cos1 * rgi.L2.y, Fx.z * cos1 * rgi.L2.z);
                            if (!finite3(fr)) fr = V3(0, 0, 0);
                            float fc = length3(fr);
                            rgi.W = fc > EPSILON_ ? rgi.w_sum / fc : 0.0f;
                            rgi.M = 1;
                        }
                        out = debug;
                    } else out = L1;                                                             /* pass3:457-462 shows L1 */
                }
                size_t slot = map_pixel_id(p->width, x, (uint32_t)y);
                if (res_di40) store_res((uint8_t*)res_di40 + slot * 40, &rdi);
                if (res_gi40) store_res((uint8_t*)res_gi40 + slot * 40, &rgi);
                if (sample60) {                                                                  /* SampleData, Reservoir_v6.hlsl:2-11 */
                    uint8_t* d = (uint8_t*)sample60 + slot * 60;
                    uint16_t m16 = (uint16_t)mID, h[3] = {half_bits(L1.x), half_bits(L1.y), half_bits(L1.z)};
                    memcpy(d, &x1, 12); memcpy(d + 12, &m16, 2); memcpy(d + 14, h, 6); memcpy(d + 20, &n1, 12); memcpy(d + 32, &ov, 12);
                    memcpy(d + 44, &objID, 4); memcpy(d + 48, &debug, 12);
                }
                float* a = accum + ((size_t)y * p->width + x) * 4;
                if (finite3(out)) { a[0] += out.x; a[1] += out.y; a[2] += out.z; a[3] += 1.0f; }
            }
        }
        c0 += cnt[0]; c1 += cnt[1]; c2 += cnt[2];
    }
    if (ray_counts) { ray_counts[0] = c0; ray_counts[1] = c1; ray_counts[2] = c2; }
    return 0;
}

/* =====================================================================================================
 * ReSTIR temporal reuse (pass 2, RayGen_v6_pass2.hlsl:46-204) and spatial reuse + final shade (pass 3,
 * RayGen_v6_pass3.hlsl:46-441) with the pairwise-MIS helpers of MIS_v6.hlsl / MIS_GI_v6.hlsl, restated
 * literally on the reference's packed buffers.  Constants: Common_v6.hlsl:14-26.
 * DEVIATIONS: pixels whose primary ray missed (mID = 0xFFFE) are skipped by both passes (v6 runs them on
 * zero-filled out-of-bounds reads); a reprojected pixel outside the image reads as a zero record (D3D
 * out-of-bounds read); sin/cos as everywhere; uint(time) := frame_seed.  The view-change reset of
 * gPermanentData (pass3:407-423) is the caller's job (rtx_clear_accum / Renderer facade).
 * ===================================================================================================== */
#define SPATIAL_CANDIDATES 3
#define SPATIAL_MAX_TRIES 9
#define SPATIAL_RADIUS 20
#define SPATIAL_M_CAP 128
#define TEMPORAL_M_CAP 16
#define W_SUM_THRESHOLD 5.0f
#define J_THRESHOLD 5.0f

static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) { float f = (float)m * (1.0f / 16777216.0f); return u2f(f2u(f) | sign); }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
typedef struct { v3 x1; uint32_t mID; v3 L1; v3 n1; v3 o; uint32_t objID; v3 debug; } sdata_t;
static res_t load_res(const uint8_t* p) {
    res_t r; float f[8]; uint16_t h[4]; memcpy(f, p, 32); memcpy(h, p + 32, 8);
    r.x2 = V3(f[0], f[1], f[2]); r.w_sum = f[3]; r.n2 = V3(f[4], f[5], f[6]); r.W = f[7];
    r.L2 = V3(half_to_float(h[0]), half_to_float(h[1]), half_to_float(h[2])); r.M = h[3];
    return r;
}
static sdata_t load_sd(const uint8_t* d) {
    sdata_t s; uint16_t m16, h[3];
    memcpy(&s.x1, d, 12); memcpy(&m16, d + 12, 2); memcpy(h, d + 14, 6); memcpy(&s.n1, d + 20, 12); memcpy(&s.o, d + 32, 12);
    memcpy(&s.objID, d + 44, 4); memcpy(&s.debug, d + 48, 12);
    s.mID = m16; s.L1 = V3(half_to_float(h[0]), half_to_float(h[1]), half_to_float(h[2]));
    return s;
}
static const uint8_t g_zero60[60] = {0};
static inline float minf_u(float cap, uint32_t m) { return (float)(m < (uint32_t)cap ? m : (uint32_t)cap); }

/* GetP_Hat / GetP_Hat_GI: Sampler_v6.hlsl:163-181 with ReconnectGI :134-160 */
static float get_p_hat(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 x1, v3 n1, v3 x2, v3 n2, v3 L2, v3 o, int vis, uint64_t cnt[3]) {
    float f_g = length3(reconnect_di(c, mid, flags, x1, n1, x2, n2, L2, o));
    float v = 1.0f;
    if (vis) { v3 dv = sub3(x2, x1); v = visibility(c, x1, n1, normalize3(dv), length3(dv), cnt); }
    return f_g * v;
}
static v3 get_p_hat_gi(const orc_ctx* c, uint32_t mid, uint32_t flags, v3 x1, v3 n1, v3 x2, v3 L, v3 o, int vis, uint64_t cnt[3]) {
    v3 dir = sub3(x2, x1);
    float cos1 = fabsf(dot3(n1, normalize3(dir)));
    v3 f0, f1; float q0, q1, pd, ps;
    lobes(c, mid, flags, n1, normalize3(dir), normalize3(o), normalize3(o), &f0, &f1, &q0, &q1, &pd, &ps);
    v3 Fx = add3(safe_mul3(pd, f0), safe_mul3(ps, f1));
    v3 fr = V3(Fx.x * cos1 * L.x, Fx.y * cos1 * L.y, Fx.z * cos1 * L.z);
    if (!finite3(fr)) fr = V3(0, 0, 0);
    float v = 1.0f;
    if (vis) v = visibility(c, x1, n1, normalize3(dir), length3(dir), cnt);
    return scale3(fr, v);
}
static inline float get_w(float w_sum, float p_hat) { return p_hat > EPSILON_ ? w_sum / p_hat : 0.0f; }
/* Jacobian_Reconnection: Sampler_v6.hlsl:48-68 */
static float jacobian(const sdata_t* r, const sdata_t* q, v3 x2q, v3 n2q) {
    v3 vq = sub3(x2q, q->x1), vr = sub3(x2q, r->x1);
    float cq = fabsf(dot3(normalize3(neg3(vq)), normalize3(n2q))), cr = fabsf(dot3(normalize3(neg3(vr)), normalize3(n2q)));
    return (cq / cr) * (dot3(vr, vr) / dot3(vq, vq));
}
static inline int valid_res(const res_t* r) { return length3(r->n2) > 0.0f && length3(r->L2) > 0.0f && r->w_sum > 0.0f && r->M > 0; }   /* Sampler_v6.hlsl:7-14 */
static inline int valid_res_gi(const res_t* r) { return r->w_sum > 0.0f && r->M > 0; }                                                  /* :17-22 */
static inline int reject_distance(v3 x1, v3 x2, v3 cam, float thr) {                                                                     /* Common_v6.hlsl:342-350 */
    float d1 = length3(sub3(x1, cam)), d2 = length3(sub3(x2, cam));
    return fabsf(d1 - d2) / maxf(d1, d2) > thr;
}
static inline int reject_jacobian(float J, float thr) { return J > thr || J < 1.0f / thr || is_nan(J) || is_inf(J); }                 /* :316-320 */
static inline v3 mul44(const float* m, v3 p, float w, float* ow) {
    *ow = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * w;
    return V3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * w, m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * w, m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * w);
}
/* GetBestReprojectedPixel_d: Sampler_v6.hlsl:738-785 */
static void reproject(const orc_ctx* c, v3 world, uint32_t objID, float W, float H, int* px, int* py) {
    float w0, w1, w2, w3;
    const inst_t* in = &c->insts[objID < c->ninst ? objID : 0];
    v3 lp = mul44(in->o2w_inv, world, 1.0f, &w0);
    v3 pw = mul44(in->prev_o2w, lp, w0, &w1);
    v3 vp = mul44(c->prev_view, pw, w1, &w2);
    v3 cp = mul44(c->prev_proj, vp, w2, &w3);
    if (w3 <= 0.0f) { *px = -1; *py = -1; return; }
    float ux = (cp.x / w3) * 0.5f + 0.5f, uy = (cp.y / w3) * 0.5f + 0.5f;
    uy = 1.0f - uy;
    *px = (int)rintf(ux * W); *py = (int)rintf(uy * H);
}
/* GetRandomPixelCircleWeighted: Common_v6.hlsl:202-244 (spatial_exponent = 1) */
static void random_pixel(uint32_t radius, uint32_t w, uint32_t h, uint32_t x, uint32_t y, uint32_t seed[2], int* ox, int* oy) {
    int nx, ny;
    do {
        float u = rnd(seed);
        float r = (float)radius * u;
        float ang = rnd(seed) * 6.2831853f;
        float sn, cs; orc_sincos(ang, &sn, &cs);
        nx = (int)x + (int)(cs * r); ny = (int)y + (int)(sn * r);
        while (nx < 0 || nx >= (int)w) { if (nx < 0) nx = -nx; else nx = 2 * (int)w - nx - 2; }
        while (ny < 0 || ny >= (int)h) { if (ny < 0) ny = -ny; else ny = 2 * (int)h - ny - 2; }
    } while (nx == (int)x && ny == (int)y);
    *ox = nx; *oy = ny;
}

typedef struct { uint8_t *cur_di, *cur_gi, *cur_sd, *last_di, *last_gi, *last_sd; } restir_bufs;

static void restir_pass2_pixel(const orc_ctx* c, const orc_params* p, const restir_bufs* B, uint32_t x, uint32_t y, uint64_t cnt[3]) {
    const uint32_t flags = p->flags;
    size_t slot = map_pixel_id(p->width, x, y);
    res_t rc = load_res(B->cur_di + slot * 40), gc = load_res(B->cur_gi + slot * 40);
    sdata_t sc = load_sd(B->cur_sd + slot * 60);
    if (!(sc.L1.x == 0.0f && sc.L1.y == 0.0f && sc.L1.z == 0.0f) || sc.mID == 0xFFFEu || sc.mID >= c->nmat) return;
    v3 cam = V3(c->viewI[12], c->viewI[13], c->viewI[14]);
    uint32_t seed[2]; orc_seed_init(x, y, 2, p->frame_seed, seed);                              /* pass2:78-79 */
    int px, py; reproject(c, sc.x1, sc.objID, (float)p->width, (float)p->height, &px, &py);
    int inside = px >= 0 && py >= 0 && px < (int)p->width && py < (int)p->height;
    size_t ts = inside ? map_pixel_id(p->width, (uint32_t)px, (uint32_t)py) : 0;
    static const uint8_t zero40[40] = {0};
    res_t rl = load_res(inside ? B->last_di + ts * 40 : zero40), gl = load_res(inside ? B->last_gi + ts * 40 : zero40);
    sdata_t sl = load_sd(inside ? B->last_sd + ts * 60 : g_zero60);
    int base_ok = (px != -1 && py != -1) && length3(sl.L1) == 0.0f && !reject_distance(sc.x1, sl.x1, cam, 0.1f) && sl.mID == sc.mID;
    int acc_di = base_ok && valid_res(&rl) && (rl.x2.x != 0.0f && rl.x2.y != 0.0f && rl.x2.z != 0.0f);
    int acc_gi = base_ok && !(gl.w_sum > W_SUM_THRESHOLD) && valid_res_gi(&gl);
    uint32_t mid = sc.mID;
    if (acc_di) {                                                                                 /* pass2:112-152 */
        float M_sum = minf_u(TEMPORAL_M_CAP, rc.M) + minf_u(TEMPORAL_M_CAP, rl.M);
        float mc = minf_u(TEMPORAL_M_CAP, rc.M), ml = minf_u(TEMPORAL_M_CAP, rl.M);
        float mi_c = mc / M_sum;                                                                  /* MIS_v6.hlsl:62-70 */
        { float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
        float mi_t;                                                                               /* :72-79 */
        { float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
        if (length3(rl.n2) == 0.0f) { mi_c = 1.0f; mi_t = 0.0f; }
        float w_c = mi_c * get_p_hat(c, mid, flags, sc.x1, sc.n1, rc.x2, rc.n2, rc.L2, sc.o, 0, cnt) * rc.W;
        float w_t = mi_t * get_p_hat(c, mid, flags, sc.x1, sc.n1, rl.x2, rl.n2, rl.L2, sc.o, 1, cnt) * rl.W;
        rc.M = (uint32_t)mc; rc.w_sum = w_c;
        rc.w_sum += w_t; rc.M = (rc.M + (uint32_t)ml) & 0xFFFFu;                                  /* UpdateReservoir */
        if (rnd(seed) < w_t / rc.w_sum) { rc.x2 = rl.x2; rc.n2 = rl.n2; rc.L2 = rl.L2; }
        float p_hat = get_p_hat(c, mid, flags, sc.x1, sc.n1, rc.x2, rc.n2, rc.L2, sc.o, 0, cnt);
        rc.W = get_w(rc.w_sum, p_hat);
    }
    if (acc_gi) {                                                                                 /* pass2:155-198 */
        float mc = minf_u(TEMPORAL_M_CAP, gc.M), ml = minf_u(TEMPORAL_M_CAP, gl.M), M_sum = mc + ml;
        float mi_c = mc / M_sum;
        { float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
        float mi_t;
        { float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
        float w_c = mi_c * length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, gc.x2, gc.L2, sc.o, 0, cnt)) * gc.W;
        float w_t = mi_t * length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, gl.x2, gl.L2, sc.o, 1, cnt)) * gl.W;
        gc.M = (uint32_t)mc; gc.w_sum = w_c;
        gc.w_sum += w_t; gc.M = (gc.M + (uint32_t)ml) & 0xFFFFu;
        if (rnd(seed) < w_t / gc.w_sum) { gc.x2 = gl.x2; gc.n2 = gl.n2; gc.L2 = gl.L2; }
        gc.W = get_w(gc.w_sum, length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, gc.x2, gc.L2, sc.o, 0, cnt)));
    }
    store_res(B->cur_di + slot * 40, &rc); store_res(B->cur_gi + slot * 40, &gc);
}

/* pass 3 reads only `cur_*` of other pixels (written by pass 2) and writes `last_*` of its own pixel */
static int restir_pass3_pixel(const orc_ctx* c, const orc_params* p, const restir_bufs* B, uint32_t x, uint32_t y, v3* out, uint64_t cnt[3]) {
    const uint32_t flags = p->flags, W = p->width, H = p->height;
    size_t slot = map_pixel_id(W, x, y);
    sdata_t sc = load_sd(B->cur_sd + slot * 60);
    if (!(sc.L1.x == 0.0f && sc.L1.y == 0.0f && sc.L1.z == 0.0f)) { *out = sc.L1; return 1; }   /* pass3:457-462 */
    if (sc.mID == 0xFFFEu || sc.mID >= c->nmat) { *out = V3(0, 0, 0); return 1; }
    v3 cam = V3(c->viewI[12], c->viewI[13], c->viewI[14]);
    uint32_t seed[2]; orc_seed_init(x, y, 3, p->frame_seed, seed);
    const uint32_t mid = sc.mID; const matopt_t* m = &c->mopt[mid];
    res_t rcur = load_res(B->cur_di + slot * 40), gcur = load_res(B->cur_gi + slot * 40);
    size_t cand_di[SPATIAL_CANDIDATES], cand_gi[SPATIAL_CANDIDATES]; int n_di = 0, n_gi = 0;
    float M_sum_DI = minf_u(SPATIAL_M_CAP, rcur.M), M_sum_GI = minf_u(SPATIAL_M_CAP, gcur.M);
    for (int a = 0; a < SPATIAL_MAX_TRIES && n_di < SPATIAL_CANDIDATES; a++) {                    /* pass3:106-135 */
        int nx, ny; random_pixel(SPATIAL_RADIUS, W, H, x, y, seed, &nx, &ny);
        size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        sdata_t sn = load_sd(B->cur_sd + pr * 60); res_t rn = load_res(B->cur_di + pr * 40);
        int ok = !(dot3(sc.n1, sn.n1) < 0.9f) && !reject_distance(sc.x1, sn.x1, cam, 0.1f) && valid_res(&rn) && length3(sn.L1) == 0.0f && sn.mID == sc.mID;
        if (ok) { cand_di[n_di++] = pr; M_sum_DI += minf_u(SPATIAL_M_CAP, rn.M); }
    }
    for (int a = 0; a < SPATIAL_MAX_TRIES && n_gi < SPATIAL_CANDIDATES; a++) {                    /* pass3:146-186 */
        int nx, ny; random_pixel(SPATIAL_RADIUS, W, H, x, y, seed, &nx, &ny);
        size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        sdata_t sn = load_sd(B->cur_sd + pr * 60); res_t gn = load_res(B->cur_gi + pr * 40);
        int ok = m->Pr > 0.3f && !reject_distance(sc.x1, sn.x1, cam, 0.1f) && !(dot3(normalize3(sub3(gn.x2, sc.x1)), sc.n1) < 0.0f) &&
                 !(gn.w_sum > W_SUM_THRESHOLD) && valid_res_gi(&gn) && !reject_jacobian(jacobian(&sn, &sc, gn.x2, gn.n2), J_THRESHOLD) &&
                 length3(sn.L1) == 0.0f && sn.mID == sc.mID;
        if (ok) { cand_gi[n_gi++] = pr; M_sum_GI += minf_u(SPATIAL_M_CAP, gn.M); }
    }
    const res_t can = rcur, can_gi = gcur;
    /* GenPairwiseMIS_canonical: MIS_v6.hlsl:2-37 */
    float cMmin = minf_u(SPATIAL_M_CAP, can.M), cMmax = M_sum_DI - cMmin;
    float p_c = get_p_hat(c, mid, flags, sc.x1, sc.n1, can.x2, can.n2, can.L2, sc.o, 0, cnt);
    float c_m_num = cMmin * p_c, mi_c = cMmin / M_sum_DI;
    for (int j = 0; j < n_di; j++) {
        sdata_t sn = load_sd(B->cur_sd + cand_di[j] * 60); res_t rn = load_res(B->cur_di + cand_di[j] * 40);
        float nM = minf_u(SPATIAL_M_CAP, rn.M);
        float p_from = get_p_hat(c, mid, flags, sn.x1, sn.n1, can.x2, can.n2, can.L2, sn.o, 1, cnt);
        float m_den = c_m_num + (cMmax * p_from);
        if (m_den > 0.0f) mi_c += (nM / M_sum_DI) * (c_m_num / m_den);
    }
    float w_c = mi_c * get_p_hat(c, mid, flags, sc.x1, sc.n1, can.x2, can.n2, can.L2, sc.o, 0, cnt) * can.W;
    /* GenPairwiseMIS_canonical_GI: MIS_GI_v6.hlsl:2-41 */
    float gMmin = minf_u(SPATIAL_M_CAP, can_gi.M), gMmax = M_sum_GI - gMmin;
    float pg_c = length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, can_gi.x2, can_gi.L2, sc.o, 0, cnt));
    float g_m_num = gMmin * pg_c, mi_c_gi = gMmin / M_sum_GI;
    for (int j = 0; j < n_gi; j++) {
        sdata_t sn = load_sd(B->cur_sd + cand_gi[j] * 60); res_t gn = load_res(B->cur_gi + cand_gi[j] * 40);
        float nM = minf_u(SPATIAL_M_CAP, gn.M);
        float j_gi = jacobian(&sc, &sn, can_gi.x2, can_gi.n2);
        float p_from = length3(get_p_hat_gi(c, mid, flags, sn.x1, sn.n1, can_gi.x2, can_gi.L2, sn.o, 1, cnt)) * j_gi;
        float m_den = g_m_num + (gMmax * p_from);
        if (m_den > 0.0f) mi_c_gi += (nM / M_sum_GI) * (g_m_num / m_den);
    }
    mi_c_gi = minf(maxf(mi_c_gi, 0.0f), 1.0f);
    float w_c_gi = mi_c_gi * length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, can_gi.x2, can_gi.L2, sc.o, 0, cnt)) * can_gi.W;
    rcur.M = (uint32_t)cMmin; rcur.w_sum = w_c;
    gcur.M = (uint32_t)gMmin; gcur.w_sum = w_c_gi;
    for (int v = 0; v < n_di; v++) {                                                              /* pass3:247-283 + MIS_v6.hlsl:40-59 */
        sdata_t sn = load_sd(B->cur_sd + cand_di[v] * 60); res_t rn = load_res(B->cur_di + cand_di[v] * 40);
        float pc2 = get_p_hat(c, mid, flags, sc.x1, sc.n1, can.x2, can.n2, can.L2, sc.o, 0, cnt);
        float p_from = get_p_hat(c, mid, flags, sn.x1, sn.n1, can.x2, can.n2, can.L2, sn.o, 0, cnt);
        float m_num = (M_sum_DI - cMmin) * p_from, m_den = m_num + (cMmin * pc2);
        float mi_s = m_den > 0.0f ? (minf_u(SPATIAL_M_CAP, rn.M) / M_sum_DI) * (m_num / m_den) : 0.0f;
        float w_s = mi_s * get_p_hat(c, mid, flags, sc.x1, sc.n1, rn.x2, rn.n2, rn.L2, sc.o, 0, cnt) * rn.W;
        rcur.w_sum += w_s; rcur.M = (rcur.M + (uint32_t)minf_u(SPATIAL_M_CAP, rn.M)) & 0xFFFFu;
        if (rnd(seed) < w_s / rcur.w_sum) { rcur.x2 = rn.x2; rcur.n2 = rn.n2; rcur.L2 = rn.L2; }
    }
    for (int v = 0; v < n_gi; v++) {                                                              /* pass3:286-334 + MIS_GI_v6.hlsl:44-75 */
        sdata_t sn = load_sd(B->cur_sd + cand_gi[v] * 60); res_t gn = load_res(B->cur_gi + cand_gi[v] * 40);
        float pc2 = length3(get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, can_gi.x2, can_gi.L2, sc.o, 0, cnt));
        float jj = jacobian(&sc, &sn, can_gi.x2, can_gi.n2);
        float p_from = length3(get_p_hat_gi(c, mid, flags, sn.x1, sn.n1, can_gi.x2, can_gi.L2, sn.o, 0, cnt)) * jj;
        float m_num = (M_sum_GI - gMmin) * p_from, m_den = m_num + (gMmin * pc2);
        float mi_s = m_den > 0.0f ? minf(maxf((minf_u(SPATIAL_M_CAP, gn.M) / M_sum_GI) * (m_num / m_den), 0.0f), 1.0f) : 0.0f;
        float j_gi = jacobian(&sn, &sc, gn.x2, gn.n2);
        v3 f_gi = get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, gn.x2, gn.L2, sc.o, 1, cnt);
        float w_s = mi_s * length3(f_gi) * gn.W * j_gi;
        if (j_gi != 0.0f) {
            gcur.w_sum += w_s; gcur.M = (gcur.M + (uint32_t)minf_u(SPATIAL_M_CAP, gn.M)) & 0xFFFFu;
            if (rnd(seed) < w_s / gcur.w_sum) { gcur.x2 = gn.x2; gcur.n2 = gn.n2; gcur.L2 = gn.L2; }
        }
    }
    float p_hat = get_p_hat(c, mid, flags, sc.x1, sc.n1, rcur.x2, rcur.n2, rcur.L2, sc.o, 1, cnt);                 /* pass3:336-347 */
    rcur.W = get_w(rcur.w_sum, p_hat);
    v3 acc = scale3(reconnect_di(c, mid, flags, sc.x1, sc.n1, rcur.x2, rcur.n2, rcur.L2, sc.o), rcur.W);
    v3 f_fin = get_p_hat_gi(c, mid, flags, sc.x1, sc.n1, gcur.x2, gcur.L2, sc.o, 0, cnt);                          /* :361-372 */
    gcur.W = get_w(gcur.w_sum, length3(f_fin));
    acc = add3(acc, scale3(f_fin, gcur.W));
    store_res(B->last_di + slot * 40, &rcur); store_res(B->last_gi + slot * 40, &gcur);                             /* :434-436 */
    memcpy(B->last_sd + slot * 60, B->cur_sd + slot * 60, 60);
    *out = acc;
    return 1;
}

/* one ReSTIR frame = pass 1 (one sample, id 1) + pass 2 + pass 3; `last_*` carry the state to the next frame */
int orc_restir_frame(orc_ctx* c, const orc_params* p, float* accum, void* cur_di, void* cur_gi, void* cur_sd, void* last_di, void* last_gi, void* last_sd, uint64_t ray_counts[3]) {
    restir_bufs B = {(uint8_t*)cur_di, (uint8_t*)cur_gi, (uint8_t*)cur_sd, (uint8_t*)last_di, (uint8_t*)last_gi, (uint8_t*)last_sd};
    orc_params p1 = *p; p1.spp = 1; p1.sample_base = 1;
    float* scratch = (float*)calloc((size_t)p->width * p->height * 4, sizeof(float));
    uint64_t cnt1[3];
    orc_render_v6_pass1(c, &p1, scratch, cur_di, cur_gi, cur_sd, cnt1);
    free(scratch);
    uint64_t c0 = cnt1[0], c1 = cnt1[1], c2 = cnt1[2];
    int nth = c->nthreads;
#ifdef _OPENMP
    if (nth <= 0) nth = omp_get_max_threads();
#else
    nth = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth) reduction(+ : c2)
    for (int64_t y = 0; y < (int64_t)p->height; y++) {
        uint64_t cnt[3] = {0, 0, 0};
        for (uint32_t x = 0; x < p->width; x++) if (owns_pixel(p, x, (uint32_t)y)) restir_pass2_pixel(c, p, &B, x, (uint32_t)y, cnt);
        c2 += cnt[2];
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth) reduction(+ : c2)
    for (int64_t y = 0; y < (int64_t)p->height; y++) {
        uint64_t cnt[3] = {0, 0, 0};
        for (uint32_t x = 0; x < p->width; x++) {
            if (!owns_pixel(p, x, (uint32_t)y)) continue;
            v3 out;
            if (restir_pass3_pixel(c, p, &B, x, (uint32_t)y, &out, cnt)) {
                float* a = accum + ((size_t)y * p->width + x) * 4;
                if (finite3(out)) { a[0] += out.x; a[1] += out.y; a[2] += out.z; a[3] += 1.0f; }    /* pass3:388-405 */
            }
        }
        c2 += cnt[2];
    }
    if (ray_counts) { ray_counts[0] = c0; ray_counts[1] = c1; ray_counts[2] = c2; }
    return 0;
}

/* RayGen_v6_pass3.hlsl:405,428-441 + Common_v6.hlsl:353-376; RGBA8 UNORM store */
void orc_srgb8(const float* accum, uint32_t npix, uint8_t* out) {
    for (uint32_t i = 0; i < npix; i++) {
        const float* a = accum + (size_t)i * 4;
        float cnt = maxf(a[3], 1.0f);
        float c[3] = {a[0] / cnt, a[1] / cnt, a[2] / cnt};
        if (is_nan(c[0]) || is_nan(c[1]) || is_nan(c[2])) { c[0] = 1; c[1] = 0; c[2] = 1; }
        if (is_inf(c[0]) || is_inf(c[1]) || is_inf(c[2])) { c[0] = 0; c[1] = 1; c[2] = 1; }
        for (int k = 0; k < 3; k++) {
            float v = c[k] <= 0.0031308f ? 12.92f * c[k] : 1.055f * orc_pow(c[k], 1.0f / 2.4f) - 0.055f;
            v = saturatef(v);
            out[(size_t)i * 4 + k] = (uint8_t)(int)(v * 255.0f + 0.5f);
        }
        out[(size_t)i * 4 + 3] = 255;
    }
}
