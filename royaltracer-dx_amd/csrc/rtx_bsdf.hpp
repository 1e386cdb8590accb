// rtx_bsdf.hpp — BSDF / light leaf math of the v6 shader set, as __host__ __device__ functions.
// Each function cites the HLSL it restates (paths relative to /root/reference/Pathtracer/include).
#pragma once
#include "rtx_math.hpp"

namespace rtx {

// Working copy of a material on the GPU: the fp16-rounded MaterialOptimized (Common_v6.hlsl:62-74,
// CreateMaterialOptimized Sampler_v6.hlsl:71-83) widened back to float, plus the full-precision
// multiscatter LUT that ESS_LUT reads from materials[mID] (GGX_v6.hlsl:17-18), plus Kd / PI, plus dissolve and Ni.  160 bytes.
struct MatGPU {
    float Kd[3]; float Pr;
    float Ks[3]; float Pm;
    float Ke[3]; float Ke_len;      // length(Ke) of the rounded copy
    float KeFull[3]; float KeFullLen;   // full-precision Ke and its length: `length(materials[mID].Ke) > 0` tests (pass1:104, Path_Sampler_v6.hlsl:55)
    float LUT[16];
    float KdPi[3]; float pad;       // Kd / PI, divided once on the host (the same IEEE division the shader would do per evaluation)
    float alpha, Ni, pad1, pad2;    // EXTENSION (strategy 3, RTX_FLAG_TRANSMISSION): dissolve = Kd.w (fp16-rounded, MaterialOptimized.Kd.w) and the full-precision Material.Ni
};
static_assert(sizeof(MatGPU) == 160, "MatGPU must be 160 bytes");

// GGX_v6.hlsl:26-29; pow(abs(1-c),5) written as repeated multiplication
RTX_HD f3 schlick(f3 F0, float cosT) {
    float x = fabsf(1.0f - cosT);
    float x2 = x * x; float x5 = x2 * x2 * x;
    return mk3(saturate(F0.x + (1.0f - F0.x) * x5), saturate(F0.y + (1.0f - F0.y) * x5), saturate(F0.z + (1.0f - F0.z) * x5));
}
// GGX_v6.hlsl:31-40
RTX_HD float d_ggx(float NdotH, float rough) {
    float alpha = rough * rough, alpha2 = alpha * alpha, nh2 = NdotH * NdotH;
    float den = nh2 * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (kPI * den * den);
}
// GGX_v6.hlsl:43-52
RTX_HD float g2_smith(float NdotV, float NdotL, float alpha) {
    float a2 = alpha * alpha;
    float dA = NdotV * sqrtf(a2 + (1.0f - a2) * NdotL * NdotL);
    float dB = NdotL * sqrtf(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * NdotL * NdotV / (dA + dB);
}
// GGX_v6.hlsl:55-61
RTX_HD float g1_smith(float NdotV, float alpha) {
    float a2 = alpha * alpha;
    float dC = sqrtf(a2 + (1.0f - a2) * NdotV * NdotV) + NdotV;
    return 2.0f * NdotV / dC;
}
// GGX_v6.hlsl:1-23
RTX_HD float ess_lut(const MatGPU& m, float NdotV) {
    NdotV = saturate(NdotV);
    float f = NdotV * 15.0f;
    int i0 = (int)floorf(f);
    int i1 = i0 + 1 < 15 ? i0 + 1 : 15;
    float w = f - (float)i0;
    float v0 = m.LUT[i0], v1 = m.LUT[i1];
    return v0 + w * (v1 - v0);
}
// Lambertian_v6.hlsl:54-58
RTX_HD f3 lambert_eval(const MatGPU& m) { return mk3(m.KdPi[0], m.KdPi[1], m.KdPi[2]); }   // Kd / PI (Lambertian_v6.hlsl:44-50), precomputed in MatGPU
// Lambertian_v6.hlsl:61-64 (L = -incoming)
RTX_HD float lambert_pdf(f3 n, f3 L) { return maxf_(dot(n, L), kEps) * kInvPI; }
// GGX_v6.hlsl:174-206 (dot products are not clamped in v6)
RTX_HD f3 ggx_eval(const MatGPU& m, f3 normal, f3 Lin, f3 Vin) {
    f3 N = normalize(normal), V = normalize(Vin), L = normalize(Lin);
    f3 H = normalize(V + L);
    float NdotV = dot(N, V), NdotL = dot(N, L), NdotH = dot(N, H), VdotH = dot(V, H);
    f3 Ks = mk3(m.Ks[0], m.Ks[1], m.Ks[2]);
    f3 F = schlick(Ks, VdotH);
    float D = d_ggx(NdotH, m.Pr);
    float G = g2_smith(NdotV, NdotL, m.Pr * m.Pr);
    float den = 4.0f * NdotV * NdotL;
    if (den < kEps) return mk3(0.0f, 0.0f, 0.0f);
    f3 spec = mk3(F.x * D * G / den, F.y * D * G / den, F.z * D * G / den);
    float Ess = ess_lut(m, NdotV);
    float kms = (1.0f - Ess) / Ess;
    f3 r = mk3(spec.x * (1.0f + Ks.x * kms), spec.y * (1.0f + Ks.y * kms), spec.z * (1.0f + Ks.z * kms));
    return finite3(r) ? r : mk3(0.0f, 0.0f, 0.0f);
}
// GGX_v6.hlsl:209-224
RTX_HD float ggx_pdf(const MatGPU& m, f3 normal, f3 Lin, f3 Vin) {
    f3 N = normalize(normal), V = normalize(Vin), L = normalize(Lin);
    f3 H = normalize(V + L);
    float NdotH = dot(N, H), NdotV = dot(N, V);
    float alpha = m.Pr * m.Pr;
    return g1_smith(NdotV, alpha) * d_ggx(NdotH, m.Pr) / (NdotV * 4.0f);
}
// ---- EXTENSION: strategy 3, rough dielectric transmission (RTX_FLAG_TRANSMISSION = 4) ----------------------------------------------------
// The reference names the strategy and leaves it a stub: "3 - Refraction", `//p_d *= alpha;` ("Adjust for translucency"), `// Refraction,
// currently replaced by diffuse (later 3)`, `//SampleBTDF_GGX / EvaluateBTDF_GGX / BTDF_PDF_GGX` (BRDF_v6.hlsl:5,28-29,44-47,85-87,102-104,
// 120-122); Material.Ni exists and is never filled (ObjLoader.h:428-435).  Built here the way those comments point: the diffuse share of a
// material with dissolve alpha = Kd.w < 1 splits into p_d alpha (Lambert) and p_d (1 - alpha) (transmission through a GGX interface of index
// Ni; Walter et al. 2007, visible-normal sampling, Schlick Fresnel with F0 = Ks like the reflection lobe).  Parity for it is UNPINNED BY
// DEFINITION (nothing to restate); it is pinned by its own properties in tests/test_dielectric.py.  Without the flag nothing below runs
// and every result is what it was; with the flag an opaque material (alpha = 1, or Ni within 1 % of 1: the formulation is singular at
// Ni = 1, and 1 is what the reference's loader leaves there) behaves exactly as without.
//   n     : the shading normal ON wo's SIDE (callers flip it for hits from behind: orient_transmission), so dot(n, wo) >= 0
//   eta_p : n_t / n_i of the crossing; 0 = opaque
// THIN-PANE MODEL: a transmitting surface is ONE interface standing for a whole window, and the path carries no "inside the medium" state: every
// crossing, from either side, is air -> Ni (eta_p = Ni; the ray bends towards the normal, is never totally reflected, and radiance is not scaled
// by 1 / eta_p^2).  A closed glass solid would need the medium tracked per path; the 128-byte Material / 36-byte HitInfo have no room for it.
RTX_HD float transmission_eta(const MatGPU& m, uint32_t flags, f3 outgoing, f3& normal) {
    if (!(flags & 4u) || (flags & 1u) || !(m.alpha < 1.0f) || fabsf(m.Ni - 1.0f) < 0.01f) return 0.0f;
    if (dot(normal, outgoing) < 0.0f) normal = -normal;
    return m.Ni;
}
RTX_HD f3 btdf_eval(const MatGPU& m, f3 normal, f3 Lin, f3 Vin, float eta_p, float& pdf) {
    pdf = 0.0f;
    const f3 zero = mk3(0.0f, 0.0f, 0.0f);
    f3 N = normalize(normal), V = normalize(Vin), L = normalize(Lin);
    float NdotV = dot(N, V), NdotL = dot(N, L);
    if (!(NdotV > 0.0f) || !(NdotL < 0.0f)) return zero;
    f3 H = normalize(madd3(L, eta_p, V));                       // -(eta_i wo + eta_t wi) up to sign and scale
    if (dot(N, H) < 0.0f) H = -H;
    float VdotH = dot(V, H), LdotH = dot(L, H);
    if (!(VdotH > 0.0f) || !(LdotH < 0.0f)) return zero;        // not a refraction through this microfacet
    float sq = VdotH + eta_p * LdotH;
    float den = sq * sq;
    if (den < kEps) return zero;
    float alpha = m.Pr * m.Pr;
    float D = d_ggx(dot(N, H), m.Pr);
    float G = g2_smith(NdotV, -NdotL, alpha);
    float e2 = eta_p * eta_p;
    float c = D * G * e2 * (-LdotH) * VdotH / (NdotV * (-NdotL) * den);
    f3 Fr = schlick(mk3(m.Ks[0], m.Ks[1], m.Ks[2]), VdotH);
    float q = g1_smith(NdotV, alpha) * VdotH * D / NdotV * (e2 * (-LdotH) / den);     // D_V(h) |dh / dwi|
    f3 f = mk3((1.0f - Fr.x) * c, (1.0f - Fr.y) * c, (1.0f - Fr.z) * c);
    if (!finite3(f) || is_nan(q) || is_inf(q)) return zero;
    pdf = q;
    return f;
}
// CalculateStrategyProbabilities, BRDF_v6.hlsl:50-70 -> (p_d, p_s); pt = the transmitted share of the diffuse part (extension; 0 when eta_p = 0)
RTX_HD void strategy_probs(const MatGPU& m, f3 outgoing, f3 normal, uint32_t flags, float& pd, float& ps, float eta_p, float& pt) {
    pt = 0.0f;
    if (flags & 1u) { pd = 1.0f; ps = 0.0f; return; }
    f3 fr = schlick(mk3(m.Ks[0], m.Ks[1], m.Ks[2]), dot(normal, outgoing));
    float p_s = minf_(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m.Pm);
    ps = p_s; pd = 1.0f - p_s;
    if (eta_p != 0.0f) { pt = pd * (1.0f - m.alpha); pd = pd * m.alpha; }                // BRDF_v6.hlsl:28-29 `p_d *= alpha`
}
RTX_HD void strategy_probs(const MatGPU& m, f3 outgoing, f3 normal, uint32_t flags, float& pd, float& ps) { float pt; strategy_probs(m, outgoing, normal, flags, pd, ps, 0.0f, pt); }
// SelectSamplingStrategy, BRDF_v6.hlsl:7-48.  RTX_FLAG_LAMBERT_ONLY draws no random number.
RTX_HD uint32_t select_strategy(const MatGPU& m, f3 outgoing, f3 normal, uint32_t flags, uint32_t& s0, uint32_t& s1, float eta_p = 0.0f) {
    if (flags & 1u) return 0u;
    float r = tea_next(s0, s1);
    f3 fr = schlick(mk3(m.Ks[0], m.Ks[1], m.Ks[2]), dot(normal, outgoing));
    float p_s = minf_(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m.Pm);
    if (r <= p_s) return m.Pr < 0.04f ? 0u : 1u;
    if (eta_p != 0.0f) {                                                                  // :41-47 with `p_d *= alpha` un-commented
        float p_d = (1.0f - p_s) * m.alpha;
        return r <= p_s + p_d ? 0u : 3u;
    }
    return 0u;
}
// F = p_d f_lambert + p_s f_ggx, P = p_d pdf_lambert + p_s pdf_ggx
// (Sampler_v6.hlsl:443-457, Path_Sampler_v6.hlsl:66-80); with eta_p != 0 a direction on the far side of the interface gets p_t f_t, p_t pdf_t alone
RTX_HD void bsdf_mixture(const MatGPU& m, uint32_t flags, f3 normal, f3 L, f3 outgoing, f3& F, float& P, float& pd, float& ps, float eta_p = 0.0f) {
    float pt;
    strategy_probs(m, outgoing, normal, flags, pd, ps, eta_p, pt);
    f3 f0 = lambert_eval(m); float q0 = lambert_pdf(normal, L);
    if (flags & 1u) { F = safe_mul(pd, f0); P = safe_mul(pd, q0); return; }
    if (eta_p != 0.0f && dot(normal, L) < 0.0f) {
        float q3; f3 f3_ = btdf_eval(m, normal, L, outgoing, eta_p, q3);
        F = safe_mul(pt, f3_); P = safe_mul(pt, q3);
        return;
    }
    f3 f1 = ggx_eval(m, normal, L, outgoing);
    float q1 = ggx_pdf(m, normal, L, outgoing);
    F = safe_mul(pd, f0) + safe_mul(ps, f1);
    P = safe_mul(pd, q0) + safe_mul(ps, q1);
}
// ---- the part of the mixture that depends on (material, normal, outgoing) only, computed ONCE per shading point -----------------------------------------------
// k_shade evaluates bsdf_mixture for every NEE sample and again for the sampled continuation, and draws the strategy in between — three times the strategy
// probabilities (Schlick at the view angle), and per mixture the two normalisations of N and V, the Ess look-up with its division, and Smith's G1 of the view
// direction with its square root and division (~110 VALU instructions per call).  MixView holds them; bsdf_mixture_v / select_strategy_v are bsdf_mixture /
// select_strategy with those terms handed in: the SAME operations on the same operands in the same order, so every result is the same bits
// (tests: test_bsdf_eval_and_sample_bit_exact runs the *_v forms; full-size frames unchanged).
struct MixView { float pd, ps, pt; f3 N, V; float NdotV, alpha, g1v, kms; };
RTX_HD MixView mix_view(const MatGPU& m, uint32_t flags, f3 normal, f3 outgoing, float eta_p = 0.0f) {
    MixView mv;
    strategy_probs(m, outgoing, normal, flags, mv.pd, mv.ps, eta_p, mv.pt);
    mv.N = mk3(0.0f, 0.0f, 1.0f); mv.V = mv.N; mv.NdotV = 0.0f; mv.alpha = 0.0f; mv.g1v = 0.0f; mv.kms = 0.0f;
    if (flags & 1u) return mv;
    mv.N = normalize(normal); mv.V = normalize(outgoing);
    mv.NdotV = dot(mv.N, mv.V);
    mv.alpha = m.Pr * m.Pr;
    mv.g1v = g1_smith(mv.NdotV, mv.alpha);
    const float Ess = ess_lut(m, mv.NdotV);
    mv.kms = (1.0f - Ess) / Ess;
    return mv;
}
// ggx_eval / ggx_pdf with the view terms of mv (GGX_v6.hlsl:174-224)
RTX_HD void ggx_eval_pdf_v(const MatGPU& m, const MixView& mv, f3 Lin, f3& f1, float& q1) {
    const f3 L = normalize(Lin);
    const f3 H = normalize(mv.V + L);
    const float NdotL = dot(mv.N, L), NdotH = dot(mv.N, H), VdotH = dot(mv.V, H);
    const f3 Ks = mk3(m.Ks[0], m.Ks[1], m.Ks[2]);
    const f3 F = schlick(Ks, VdotH);
    const float D = d_ggx(NdotH, m.Pr);
    const float G = g2_smith(mv.NdotV, NdotL, m.Pr * m.Pr);
    const float den = 4.0f * mv.NdotV * NdotL;
    q1 = mv.g1v * D / (mv.NdotV * 4.0f);
    if (den < kEps) { f1 = mk3(0.0f, 0.0f, 0.0f); return; }
    const f3 spec = mk3(F.x * D * G / den, F.y * D * G / den, F.z * D * G / den);
    const f3 r = mk3(spec.x * (1.0f + Ks.x * mv.kms), spec.y * (1.0f + Ks.y * mv.kms), spec.z * (1.0f + Ks.z * mv.kms));
    f1 = finite3(r) ? r : mk3(0.0f, 0.0f, 0.0f);
}
RTX_HD void bsdf_mixture_v(const MatGPU& m, uint32_t flags, const MixView& mv, f3 normal, f3 L, f3 outgoing, f3& F, float& P, float eta_p = 0.0f) {
    f3 f0 = lambert_eval(m); float q0 = lambert_pdf(normal, L);
    if (flags & 1u) { F = safe_mul(mv.pd, f0); P = safe_mul(mv.pd, q0); return; }
    if (eta_p != 0.0f && dot(normal, L) < 0.0f) {
        float q3; f3 f3_ = btdf_eval(m, normal, L, outgoing, eta_p, q3);
        F = safe_mul(mv.pt, f3_); P = safe_mul(mv.pt, q3);
        return;
    }
    f3 f1; float q1; ggx_eval_pdf_v(m, mv, L, f1, q1);
    F = safe_mul(mv.pd, f0) + safe_mul(mv.ps, f1);
    P = safe_mul(mv.pd, q0) + safe_mul(mv.ps, q1);
}
RTX_HD uint32_t select_strategy_v(const MatGPU& m, const MixView& mv, uint32_t flags, uint32_t& s0, uint32_t& s1, float eta_p = 0.0f) {
    if (flags & 1u) return 0u;
    const float r = tea_next(s0, s1);
    const float p_s = mv.ps;
    if (r <= p_s) return m.Pr < 0.04f ? 0u : 1u;
    if (eta_p != 0.0f) return r <= p_s + mv.pd ? 0u : 3u;             // mv.pd = (1 - p_s) alpha here (strategy_probs with eta_p != 0)
    return 0u;
}
// RandomUnitVectorInHemisphere, Lambertian_v6.hlsl:2-38
// (round 5) the disk sample (two draws, sqrt, sine / cosine) is separated from the geometry that uses it: sample_bsdf takes it ONCE for whatever strategy a lane drew, so a wave
// whose lanes drew different strategies does not run the generator and the trigonometry once per strategy branch (same operations per lane, same bits)
RTX_HD f3 sample_lambert_disk(f3 normal, float r, float sn, float cs) {
    float x = r * cs, y = r * sn;
    float z = sqrtf(maxf_(0.0f, 1.0f - x * x - y * y));
    f3 h = normal;
    f3 up = fabsf(normal.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    f3 right = normalize(cross(up, h));
    f3 fwd = cross(h, right);
    f3 s = lincomb3(right, x, fwd, y, h, z);
    s = normalize(s);
    if (dot(s, normal) < 0.0f) s = -s;
    return s;
}
RTX_HD f3 sample_lambert(f3 normal, uint32_t& s0, uint32_t& s1) {
    float u1 = tea_next(s0, s1), u2 = tea_next(s0, s1);
    float r = sqrtf(u1);
    float theta = kTwoPi * u2;
    float sn, cs; sincos_(theta, sn, cs);
    return sample_lambert_disk(normal, r, sn, cs);
}
// CoordinateSystem, GGX_v6.hlsl:65-76
RTX_HD void coord_system(f3 N, f3& T, f3& B) {
    if (fabsf(N.z) < 0.999f) T = normalize(cross(mk3(0.0f, 0.0f, 1.0f), N));
    else T = normalize(cross(mk3(1.0f, 0.0f, 0.0f), N));
    B = cross(N, T);
}
// SampleBRDF_GGX (Heitz 2018 VNDF), GGX_v6.hlsl:93-169: the visible half vector H (:104-157) ...
RTX_HD f3 sample_ggx_h_disk(const MatGPU& m, f3 outgoing, f3 normal, float r, float sn, float cs, f3& V) {
    float alpha = m.Pr * m.Pr;
    f3 N = normalize(normal), T1, T2;
    V = normalize(outgoing);
    coord_system(N, T1, T2);
    float vx = dot(T1, V), vy = dot(T2, V), vz = dot(N, V);
    f3 Ve = normalize(mk3(alpha * vx, alpha * vy, vz));
    float lensq = Ve.x * Ve.x + Ve.y * Ve.y;
    f3 T1h;
    if (lensq > 0.0f) { float rs = 1.0f / sqrtf(lensq); T1h = mk3(-Ve.y * rs, Ve.x * rs, 0.0f * rs); }
    else T1h = mk3(1.0f, 0.0f, 0.0f);
    f3 T2h = cross(Ve, T1h);
    float t1 = r * cs, t2 = r * sn;
    float s = 0.5f * (1.0f + Ve.z);
    t2 = (1.0f - s) * sqrtf(saturate(1.0f - t1 * t1)) + s * t2;
    float w = sqrtf(saturate(1.0f - t1 * t1 - t2 * t2));
    f3 Nh = mk3(t1 * T1h.x + t2 * T2h.x + w * Ve.x, t1 * T1h.y + t2 * T2h.y + w * Ve.y, t1 * T1h.z + t2 * T2h.z + w * Ve.z);
    f3 Ne = normalize(mk3(alpha * Nh.x, alpha * Nh.y, maxf_(0.0f, Nh.z)));
    return mk3(Ne.x * T1.x + Ne.y * T2.x + Ne.z * N.x, Ne.x * T1.y + Ne.y * T2.y + Ne.z * N.y, Ne.x * T1.z + Ne.y * T2.z + Ne.z * N.z);
}
RTX_HD f3 sample_ggx_h(const MatGPU& m, f3 outgoing, f3 normal, uint32_t& s0, uint32_t& s1, f3& V) {
    float U1 = tea_next(s0, s1), U2 = tea_next(s0, s1);
    float r = sqrtf(U1);
    float phi = 2.0f * kPI * U2;
    float sn, cs; sincos_(phi, sn, cs);
    return sample_ggx_h_disk(m, outgoing, normal, r, sn, cs, V);
}
// ... and the direction reflected about it (:159-165)
RTX_HD f3 ggx_reflect(f3 H, f3 V, f3 normal) {
    f3 I = -V;
    float k = 2.0f * dot(H, I);
    f3 smp = mk3(I.x - k * H.x, I.y - k * H.y, I.z - k * H.z);
    if (dot(smp, normal) < 0.0f) smp = -smp;   // :164-165: flipped, not rejected
    return smp;
}
RTX_HD f3 sample_ggx(const MatGPU& m, f3 outgoing, f3 normal, uint32_t& s0, uint32_t& s1) {
    f3 V; f3 H = sample_ggx_h(m, outgoing, normal, s0, s1, V);
    f3 I = -V;
    float k = 2.0f * dot(H, I);
    f3 smp = mk3(I.x - k * H.x, I.y - k * H.y, I.z - k * H.z);
    if (dot(smp, normal) < 0.0f) smp = -smp;   // :164-165: flipped, not rejected
    return smp;
}
// EXTENSION, strategy 3: refract about the same visible half vector (`//SampleBTDF_GGX`, BRDF_v6.hlsl:85-87); total internal reflection ends the path (zero vector)
RTX_HD f3 btdf_refract(f3 H, f3 V, float eta_p) {
    float eta = 1.0f / eta_p;
    float c = dot(V, H);
    float s2 = eta * eta * (1.0f - c * c);
    if (!(s2 < 1.0f)) return mk3(0.0f, 0.0f, 0.0f);
    float k = eta * c - sqrtf(1.0f - s2);
    return normalize(mk3(k * H.x - eta * V.x, k * H.y - eta * V.y, k * H.z - eta * V.z));
}
RTX_HD f3 sample_btdf(const MatGPU& m, f3 outgoing, f3 normal, float eta_p, uint32_t& s0, uint32_t& s1) {
    f3 V; f3 H = sample_ggx_h(m, outgoing, normal, s0, s1, V);
    float eta = 1.0f / eta_p;
    float c = dot(V, H);
    float s2 = eta * eta * (1.0f - c * c);
    if (!(s2 < 1.0f)) return mk3(0.0f, 0.0f, 0.0f);
    float k = eta * c - sqrtf(1.0f - s2);
    return normalize(mk3(k * H.x - eta * V.x, k * H.y - eta * V.y, k * H.z - eta * V.z));
}
// SampleBRDF, BRDF_v6.hlsl:74-88
RTX_HD f3 sample_bsdf(const MatGPU& m, uint32_t strategy, f3 outgoing, f3 normal, uint32_t& s0, uint32_t& s1, float eta_p = 0.0f) {
#ifdef RTX_SAMPLE_PER_BRANCH    // (A/B build: every strategy branch with its own draws, as until round 4)
    if (strategy == 3u) return sample_btdf(m, outgoing, normal, eta_p, s0, s1);
    return strategy == 1u ? sample_ggx(m, outgoing, normal, s0, s1) : sample_lambert(normal, s0, s1);
#else
    // every strategy starts from the same disk sample: two draws, r = sqrt(U1), sine and cosine of 2 pi U2 — with the reference's two different values of "2 pi"
    // (Lambertian_v6.hlsl:10 against GGX_v6.hlsl's 2 * PI with PI = 3.1415), selected per lane before the one evaluation
    const float U1 = tea_next(s0, s1), U2 = tea_next(s0, s1);
    const float r = sqrtf(U1);
    const bool lambert = strategy != 1u && strategy != 3u;
    const float ang = lambert ? kTwoPi * U2 : 2.0f * kPI * U2;
    float sn, cs; sincos_(ang, sn, cs);
    if (lambert) return sample_lambert_disk(normal, r, sn, cs);
    f3 V; const f3 H = sample_ggx_h_disk(m, outgoing, normal, r, sn, cs, V);       // strategies 1 and 3 share the visible half vector
    return strategy == 3u ? btdf_refract(H, V, eta_p) : ggx_reflect(H, V, normal);
#endif
}

}  // namespace rtx
