// rtx_bsdf.hpp — BSDF / light leaf math of the v6 shader set, as __host__ __device__ functions.
// Each function cites the HLSL it restates (paths relative to /root/reference/Pathtracer/include).
#pragma once
#include "rtx_math.hpp"

namespace rtx {

// Working copy of a material on the GPU: the fp16-rounded MaterialOptimized (Common_v6.hlsl:62-74,
// CreateMaterialOptimized Sampler_v6.hlsl:71-83) widened back to float, plus the full-precision
// multiscatter LUT that ESS_LUT reads from materials[mID] (GGX_v6.hlsl:17-18), plus Kd / PI.  144 bytes.
struct MatGPU {
    float Kd[3]; float Pr;
    float Ks[3]; float Pm;
    float Ke[3]; float Ke_len;      // length(Ke) of the rounded copy
    float KeFull[3]; float KeFullLen;   // full-precision Ke and its length: `length(materials[mID].Ke) > 0` tests (pass1:104, Path_Sampler_v6.hlsl:55)
    float LUT[16];
    float KdPi[3]; float pad;       // Kd / PI, divided once on the host (the same IEEE division the shader would do per evaluation)
};

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
// CalculateStrategyProbabilities, BRDF_v6.hlsl:50-70 -> (p_d, p_s)
RTX_HD void strategy_probs(const MatGPU& m, f3 outgoing, f3 normal, uint32_t flags, float& pd, float& ps) {
    if (flags & 1u) { pd = 1.0f; ps = 0.0f; return; }
    f3 fr = schlick(mk3(m.Ks[0], m.Ks[1], m.Ks[2]), dot(normal, outgoing));
    float p_s = minf_(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m.Pm);
    ps = p_s; pd = 1.0f - p_s;
}
// SelectSamplingStrategy, BRDF_v6.hlsl:7-48.  RTX_FLAG_LAMBERT_ONLY draws no random number.
RTX_HD uint32_t select_strategy(const MatGPU& m, f3 outgoing, f3 normal, uint32_t flags, uint32_t& s0, uint32_t& s1) {
    if (flags & 1u) return 0u;
    float r = tea_next(s0, s1);
    f3 fr = schlick(mk3(m.Ks[0], m.Ks[1], m.Ks[2]), dot(normal, outgoing));
    float p_s = minf_(1.0f, (fr.x + fr.y + fr.z) / 3.0f + m.Pm);
    if (r <= p_s) return m.Pr < 0.04f ? 0u : 1u;
    return 0u;
}
// F = p_d f_lambert + p_s f_ggx, P = p_d pdf_lambert + p_s pdf_ggx
// (Sampler_v6.hlsl:443-457, Path_Sampler_v6.hlsl:66-80)
RTX_HD void bsdf_mixture(const MatGPU& m, uint32_t flags, f3 normal, f3 L, f3 outgoing, f3& F, float& P, float& pd, float& ps) {
    strategy_probs(m, outgoing, normal, flags, pd, ps);
    f3 f0 = lambert_eval(m); float q0 = lambert_pdf(normal, L);
    if (flags & 1u) { F = safe_mul(pd, f0); P = safe_mul(pd, q0); return; }
    f3 f1 = ggx_eval(m, normal, L, outgoing);
    float q1 = ggx_pdf(m, normal, L, outgoing);
    F = safe_mul(pd, f0) + safe_mul(ps, f1);
    P = safe_mul(pd, q0) + safe_mul(ps, q1);
}
// RandomUnitVectorInHemisphere, Lambertian_v6.hlsl:2-38
RTX_HD f3 sample_lambert(f3 normal, uint32_t& s0, uint32_t& s1) {
    float u1 = tea_next(s0, s1), u2 = tea_next(s0, s1);
    float r = sqrtf(u1);
    float theta = kTwoPi * u2;
    float sn, cs; sincos_(theta, sn, cs);
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
// CoordinateSystem, GGX_v6.hlsl:65-76
RTX_HD void coord_system(f3 N, f3& T, f3& B) {
    if (fabsf(N.z) < 0.999f) T = normalize(cross(mk3(0.0f, 0.0f, 1.0f), N));
    else T = normalize(cross(mk3(1.0f, 0.0f, 0.0f), N));
    B = cross(N, T);
}
// SampleBRDF_GGX (Heitz 2018 VNDF), GGX_v6.hlsl:93-169
RTX_HD f3 sample_ggx(const MatGPU& m, f3 outgoing, f3 normal, uint32_t& s0, uint32_t& s1) {
    float alpha = m.Pr * m.Pr;
    f3 N = normalize(normal), V = normalize(outgoing), T1, T2;
    coord_system(N, T1, T2);
    float vx = dot(T1, V), vy = dot(T2, V), vz = dot(N, V);
    f3 Ve = normalize(mk3(alpha * vx, alpha * vy, vz));
    float lensq = Ve.x * Ve.x + Ve.y * Ve.y;
    f3 T1h;
    if (lensq > 0.0f) { float rs = 1.0f / sqrtf(lensq); T1h = mk3(-Ve.y * rs, Ve.x * rs, 0.0f * rs); }
    else T1h = mk3(1.0f, 0.0f, 0.0f);
    f3 T2h = cross(Ve, T1h);
    float U1 = tea_next(s0, s1), U2 = tea_next(s0, s1);
    float r = sqrtf(U1);
    float phi = 2.0f * kPI * U2;
    float sn, cs; sincos_(phi, sn, cs);
    float t1 = r * cs, t2 = r * sn;
    float s = 0.5f * (1.0f + Ve.z);
    t2 = (1.0f - s) * sqrtf(saturate(1.0f - t1 * t1)) + s * t2;
    float w = sqrtf(saturate(1.0f - t1 * t1 - t2 * t2));
    f3 Nh = mk3(t1 * T1h.x + t2 * T2h.x + w * Ve.x, t1 * T1h.y + t2 * T2h.y + w * Ve.y, t1 * T1h.z + t2 * T2h.z + w * Ve.z);
    f3 Ne = normalize(mk3(alpha * Nh.x, alpha * Nh.y, maxf_(0.0f, Nh.z)));
    f3 H = mk3(Ne.x * T1.x + Ne.y * T2.x + Ne.z * N.x, Ne.x * T1.y + Ne.y * T2.y + Ne.z * N.y, Ne.x * T1.z + Ne.y * T2.z + Ne.z * N.z);
    f3 I = -V;
    float k = 2.0f * dot(H, I);
    f3 smp = mk3(I.x - k * H.x, I.y - k * H.y, I.z - k * H.z);
    if (dot(smp, normal) < 0.0f) smp = -smp;   // :164-165: flipped, not rejected
    return smp;
}
// SampleBRDF, BRDF_v6.hlsl:74-88
RTX_HD f3 sample_bsdf(const MatGPU& m, uint32_t strategy, f3 outgoing, f3 normal, uint32_t& s0, uint32_t& s1) {
    return strategy == 1u ? sample_ggx(m, outgoing, normal, s0, s1) : sample_lambert(normal, s0, s1);
}

}  // namespace rtx
