// rtx_restir.hpp — the reference's own pipeline, literally: pass-1 estimator (SampleRIS + SamplePathSimple) and the ReSTIR temporal / spatial
// passes as thread-per-pixel kernels
#pragma once
#include "rtx_shade.hpp"

namespace rtx {

// ---------------------------------------------------------------------------------------------
// The v6 PASS-1 estimator, literally: RayGen_v6_pass1.hlsl:48-190 = primary hit, SampleRIS (Sampler_v6.hlsl:653-736),
// its visibility ray, SamplePathSimple (Path_Sampler_v6.hlsl:3-286), written as ONE kernel with a thread per pixel
// like the reference's raygen shader (this is the reference's own formulation; the wavefront kernels above are the
// product's estimator).  Quirks are kept (abs cosines and unshadowed NEE in the GI loop, two strategy draws per
// bounce, reservoir updates consuming random numbers, half-precision L2/E3/L1, `pdf_light = 1` initial value);
// the only deviations: a miss ends the estimator at that point, frame_seed / sample id are explicit.
// Statement order = oracle/rt_oracle.c:orc_render_v6_pass1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_round_dev(float x) {          // float -> binary16 (RNE) -> float
    const uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;
    if (a >= 0x477FF000u) return u2f(sign | 0x7F800000u);
    if (a < 0x33000001u) return u2f(sign);
    if (a < 0x38800000u) { const float r = rintf(u2f(a) * 16777216.0f); return u2f(sign | f2u(r * (1.0f / 16777216.0f))); }
    const uint32_t rem = a & 0x1FFFu; uint32_t base = a & ~0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}
__device__ __forceinline__ uint32_t half_bits_dev(float x) {
    const float r = half_round_dev(x);
    const uint32_t u = f2u(r), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return sign | 0x7C00u | ((a & 0x007FFFFFu) ? 0x200u : 0u);
    if (a == 0) return sign;
    const int e = (int)(a >> 23) - 127;
    if (e < -14) { const float q = u2f(a) * 16777216.0f; return sign | (uint32_t)q; }
    return sign | (uint32_t)((e + 15) << 10) | ((a >> 13) & 0x3FFu);
}
__device__ __forceinline__ f3 half3_dev(f3 a) { return mk3(half_round_dev(a.x), half_round_dev(a.y), half_round_dev(a.z)); }

struct Res { f3 x2; float w_sum; f3 n2; float W; f3 L2; uint32_t M; };
struct P1Ctx { const DevScene* sc; const SmallRecPair* small; const TraceLds* L; uint32_t flags; uint32_t cnt_ext, cnt_sh; };

__device__ __forceinline__ void lobes_dev(const MatGPU& m, uint32_t flags, f3 normal, f3 L, f3 out_eval, f3 out_pdf, f3& f0, f3& f1, float& q0, float& q1, float& pd, float& ps) {
    strategy_probs(m, out_eval, normal, flags, pd, ps);
    f0 = lambert_eval(m); q0 = lambert_pdf(normal, L);
    if (flags & 1u) { f1 = mk3(0.0f, 0.0f, 0.0f); q1 = 0.0f; }
    else { f1 = ggx_eval(m, normal, L, out_eval); q1 = ggx_pdf(m, normal, L, out_pdf); }
}
// lobes_dev with the terms that depend on (material, normal, view direction) only taken from mv = mix_view(m, flags, normal, V) (rtx_bsdf.hpp): for the nee light
// candidates of ONE shading point, which all call lobes_dev(m, flags, normal, L_i, V, V).  Same operations on the same operands (strategy_probs, normalize(N), normalize(V),
// N.V, Smith's G1 of V, the Ess look-up): the same bits; pd / ps are mv.pd / mv.ps.
__device__ __forceinline__ void lobes_dev_v(const MatGPU& m, uint32_t flags, const MixView& mv, f3 normal, f3 L, f3& f0, f3& f1, float& q0, float& q1) {
    f0 = lambert_eval(m); q0 = lambert_pdf(normal, L);
    if (flags & 1u) { f1 = mk3(0.0f, 0.0f, 0.0f); q1 = 0.0f; }
    else ggx_eval_pdf_v(m, mv, L, f1, q1);
}
struct LSample { f3 sp, Ln, nl; float dist2, dist, pdf_l; f3 em; };
__device__ __forceinline__ LSample light_point_dev(const DevScene& sc, f3 origin, uint32_t& s0, uint32_t& s1) {
    LSample r;
    const float rv = tea_next(s0, s1);
    int left = 0, right = (int)sc.nlights - 1, sel = 0;
    while (left <= right) { const int mid = left + (right - left) / 2; if (rv < sc.cdf[mid]) { sel = mid; right = mid - 1; } else left = mid + 1; }
    const LightGPU& lt = sc.lights[sel];
    const f3 xv = mk3(lt.xv[0], lt.xv[1], lt.xv[2]), yv = mk3(lt.yv[0], lt.yv[1], lt.yv[2]), zv = mk3(lt.zv[0], lt.zv[1], lt.zv[2]);
    float xi1 = tea_next(s0, s1), xi2 = tea_next(s0, s1);
    if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
    const float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
    r.sp = mk3(u * xv.x + v * yv.x + w * zv.x, u * xv.y + v * yv.y + w * zv.y, u * xv.z + v * yv.z + w * zv.z);
    const f3 Lv = r.sp - origin;
    r.dist2 = dot(Lv, Lv); r.dist = sqrtf(maxf_(r.dist2, kEps)); r.Ln = normalize(Lv);
    r.nl = mk3(lt.nl[0], lt.nl[1], lt.nl[2]);
    if (dot(r.nl, -r.Ln) < 0.0f) r.nl = -r.nl;
    r.pdf_l = lt.pdf_l;      // already max(EPS, weight / max(area, EPS)); the callers' max(EPS, .) is idempotent
    r.em = mk3(lt.em[0], lt.em[1], lt.em[2]);
    return r;
}
__device__ __forceinline__ bool res_update_dev(Res& r, float wi, f3 x, f3 n, f3 L, uint32_t& s0, uint32_t& s1) {
    r.w_sum += wi;
    if (tea_next(s0, s1) < wi / r.w_sum) { r.x2 = x; r.n2 = n; r.L2 = half3_dev(L); return true; }
    return false;
}
__device__ __forceinline__ f3 reconnect_di_dev(const MatGPU& m, uint32_t flags, f3 x1, f3 n1, f3 x2, f3 n2, f3 L, f3 outgoing) {
    const f3 dir = x2 - x1;
    const float dist = length(dir);
    const float cos1 = maxf_(0.0f, dot(n1, normalize(dir)));
    if (dot(n2, normalize(-dir)) < 0.0f) n2 = -n2;
    const float cos2 = maxf_(0.0f, dot(n2, normalize(-dir)));
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(m, flags, n1, normalize(dir), normalize(outgoing), normalize(outgoing), f0, f1, q0, q1, pd, ps);
    const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
    const float d2 = dist * dist;
    return mk3(F.x * L.x * cos1 * cos2 / d2, F.y * L.y * cos1 * cos2 / d2, F.z * L.z * cos1 * cos2 / d2);
}
// reconnect_di_dev with the view terms of the shading point (x1, n1, outgoing) handed in: mv = mix_view(m, flags, n1, normalize(outgoing)).  For the passes that evaluate
// several samples from ONE shading point (the merges, the final shade): same operations on the same operands, the same bits.
__device__ __forceinline__ f3 reconnect_di_dev_v(const MatGPU& m, uint32_t flags, const MixView& mv, f3 x1, f3 n1, f3 x2, f3 n2, f3 L) {
    const f3 dir = x2 - x1;
    const float dist = length(dir);
    const float cos1 = maxf_(0.0f, dot(n1, normalize(dir)));
    if (dot(n2, normalize(-dir)) < 0.0f) n2 = -n2;
    const float cos2 = maxf_(0.0f, dot(n2, normalize(-dir)));
    f3 f0, f1; float q0, q1;
    lobes_dev_v(m, flags, mv, n1, normalize(dir), f0, f1, q0, q1);
    const f3 F = safe_mul(mv.pd, f0) + safe_mul(mv.ps, f1);
    const float d2 = dist * dist;
    return mk3(F.x * L.x * cos1 * cos2 / d2, F.y * L.y * cos1 * cos2 / d2, F.z * L.z * cos1 * cos2 / d2);
}
__device__ __forceinline__ bool p1_any(P1Ctx& C, f3 o, f3 d, float tmin, float tmax) {
    float t, u, v; uint32_t prim;
    trace_ray<true>(*C.sc, C.small, *C.L, o, d, tmin, tmax, t, u, v, prim);
    C.cnt_sh++;
    return prim != kMissPrim;
}
__device__ __forceinline__ bool p1_hit(P1Ctx& C, f3 o, f3 d, float tmin, Surf& sf) {
    float t, u, v; uint32_t prim;
    trace_ray<false>(*C.sc, C.small, *C.L, o, d, tmin, kTMax, t, u, v, prim);
    if (prim == kMissPrim) return false;
    sf = surface(*C.sc, o, d, t, u, v, prim);
    return sf.mat < C.sc->nmat;
}

// ---- SampleRIS (Sampler_v6.hlsl:653-736), cut at its ONE ray (the BSDF candidate): ris_front = strategy draw + M1 light candidates + the candidate's direction,
// ris_back = the candidate's evaluation once its hit is known.  The thread-per-pixel kernel traces in between; the wavefront stages (rtx_restir_wave.hpp)
// write the ray to a queue and run ris_back in the next stage.  M2 = bsdf_samples_DI = 1 (Common_v6.hlsl:10).
constexpr uint32_t kRisM2 = 1u;
__device__ __forceinline__ f3 ris_front(const DevScene& sc, uint32_t flags, uint32_t M1, f3 outgoing, Res& rs, f3 origin, f3 normal, uint32_t mat, uint32_t& s0, uint32_t& s1) {
    const MatGPU& m = sc.mats[mat];
    const uint32_t strategy = select_strategy(m, outgoing, normal, flags, s0, s1);
    MixView mv;
    if (M1 && sc.nlights) mv = mix_view(m, flags, normal, normalize(outgoing));        // (round 4) the view terms of the M1 candidates' mixtures, once
    for (uint32_t i = 0; i < M1 && sc.nlights; i++) {
        const LSample ls = light_point_dev(sc, origin, s0, s1);
        const float cos_x = dot(normal, ls.Ln), cos_y = dot(ls.nl, -ls.Ln);
        const float G = maxf_(cos_y * cos_x / ls.dist2, kEps);
        f3 f0, f1; float q0, q1; const float pd = mv.pd, ps = mv.ps;
        lobes_dev_v(m, flags, mv, normal, ls.Ln, f0, f1, q0, q1);
        const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
        const float P = safe_mul(pd, q0 * cos_y / ls.dist2) + safe_mul(ps, q1 * cos_y / ls.dist2);
        const float p_hat = length(mk3(ls.em.x * F.x * G * 1.0f, ls.em.y * F.y * G * 1.0f, ls.em.z * F.z * G * 1.0f));
        const float pdf_light = maxf_(kEps, ls.pdf_l);
        const float mi = pdf_light / ((float)M1 * pdf_light + (float)kRisM2 * P);
        const float wi = mi * p_hat / pdf_light;
        if (p_hat > 0.0f) res_update_dev(rs, wi, ls.sp, ls.nl, ls.em, s0, s1);
    }
    return sample_bsdf(m, strategy, outgoing, normal, s0, s1);
}
// hit: the candidate ray found a surface with a valid material (p1_hit); h2 = that surface
__device__ __forceinline__ void ris_back(const DevScene& sc, uint32_t flags, uint32_t M1, f3 outgoing, Res& rs, f3 origin, f3 normal, uint32_t mat, f3 smp, bool hit, const Surf& h2,
                                         uint32_t& s0, uint32_t& s1) {
    const MatGPU& m = sc.mats[mat];
    float pdf_light = 0.0f, pdf_bsdf = 0.0f, p_hat = 0.0f;
    f3 em = mk3(0, 0, 0), x2 = mk3(0, 0, 0), n2 = mk3(0, 0, 0);
    if (hit) {
        const MatGPU& mk = sc.mats[h2.mat];
        const float Ke = mk.KeFull[0] + mk.KeFull[1] + mk.KeFull[2];
        em = mk3(mk.KeFull[0], mk.KeFull[1], mk.KeFull[2]); x2 = h2.pos; n2 = h2.normal;
        if (Ke > kEps && sc.nlights) {
            const float dist = length(h2.pos - origin), dist2 = dist * dist;
            const float cos_t = dot(h2.normal, -smp);
            pdf_light = (Ke / 3.0f) / sc.total_weight;
            f3 f0, f1; float q0, q1, pd, ps;
            lobes_dev(m, flags, normal, smp, normalize(outgoing), outgoing, f0, f1, q0, q1, pd, ps);
            const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
            pdf_bsdf = safe_mul(pd, q0 * cos_t / dist2) + safe_mul(ps, q1 * cos_t / dist2);
            const float ndot = dot(normal, smp);
            p_hat = length(mk3(F.x * em.x * ndot * cos_t / dist2, F.y * em.y * ndot * cos_t / dist2, F.z * em.z * ndot * cos_t / dist2));
        }
    }
    const float mi = pdf_bsdf / ((float)M1 * pdf_light + (float)kRisM2 * pdf_bsdf);
    const float wi = mi * p_hat / pdf_bsdf;
    if (p_hat > 0.0f) res_update_dev(rs, wi, x2, n2, em, s0, s1);
    rs.M = 1;
}

// ---- SamplePathSimple (Path_Sampler_v6.hlsl:3-286), cut at its closest-hit rays.  GiHot = what a path carries from one ray to the next; what a reservoir
// update SELECTS (the reconnection radiance L2 and, for a light sample, the end points x1s / x2s of the final shadow ray) goes to a caller-supplied sink:
// local variables in the thread-per-pixel kernel, per-pixel records in HBM in the wavefront stages (selections are rare; the hot state stays small).
//   gi_first_sample -> ray -> gi_first_hit -> [ gi_front -> ray -> gi_back ] x `bounces` -> final shadow ray (callers)
struct GiHot { f3 origin, normal, outgoing; uint32_t mat; f3 acc_f, acc_f_rec, acc_L; float acc_pdf, w_sum; };
__device__ __forceinline__ void gi_begin(GiHot& H, f3 init_point, f3 init_normal, f3 init_outgoing, uint32_t init_mat) {
    H.acc_f = mk3(1, 1, 1); H.acc_f_rec = mk3(1, 1, 1); H.acc_L = mk3(0, 0, 0); H.acc_pdf = 1.0f; H.w_sum = 0.0f;
    H.origin = init_point; H.normal = init_normal; H.outgoing = normalize(init_outgoing); H.mat = init_mat;
}
__device__ __forceinline__ f3 gi_first_sample(const DevScene& sc, uint32_t flags, const GiHot& H, uint32_t& s0, uint32_t& s1) {      // :37-52
    const uint32_t st = select_strategy(sc.mats[H.mat], H.outgoing, H.normal, flags, s0, s1);
    return sample_bsdf(sc.mats[H.mat], st, H.outgoing, H.normal, s0, s1);
}
// false: the estimator returns zero here (miss, or the first bounce ends on a light: :53-57)
__device__ __forceinline__ bool gi_first_hit(const DevScene& sc, uint32_t flags, GiHot& H, f3 smp, bool hit, const Surf& h) {
    if (!hit) return false;
    if (sc.mats[h.mat].KeFullLen > 0.0f) return false;
    const f3 incoming = normalize(-smp);
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(sc.mats[H.mat], flags, H.normal, -incoming, H.outgoing, H.outgoing, f0, f1, q0, q1, pd, ps);
    const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
    const float P = safe_mul(pd, q0) + safe_mul(ps, q1);
    const float NdotL = dot(H.normal, smp);
    H.acc_pdf *= P;
    H.acc_f = mk3(H.acc_f.x * (F.x * NdotL), H.acc_f.y * (F.y * NdotL), H.acc_f.z * (F.z * NdotL));
    H.outgoing = incoming; H.mat = h.mat; H.normal = h.normal; H.origin = h.pos;
    return true;
}
// the reservoir update of the GI path (UpdateReservoir_GI): the stored point is always (xn, nn) — the first path vertex — so only w_sum and the selection matter
__device__ __forceinline__ bool gi_update(GiHot& H, float wi, uint32_t& s0, uint32_t& s1) {
    H.w_sum += wi;
    return tea_next(s0, s1) < wi / H.w_sum;
}
// loop body up to its ray (:111-216): the unused strategy draw, `nee` unshadowed light samples, the BSDF sample.  sel(L2, true, x1s, x2s) on a selected light sample
template <class Sel>
__device__ __forceinline__ f3 gi_front(const DevScene& sc, uint32_t flags, uint32_t nee, GiHot& H, uint32_t& s0, uint32_t& s1, Sel&& sel) {
    (void)select_strategy(sc.mats[H.mat], H.outgoing, H.normal, flags, s0, s1);
    MixView mv;
    if (nee) mv = mix_view(sc.mats[H.mat], flags, H.normal, normalize(H.outgoing));   // (round 4) the view terms of the nee light samples' mixtures, once
    for (uint32_t j = 0; j < nee; j++) {
        const LSample ls = light_point_dev(sc, H.origin, s0, s1);
        float cos_x = fabsf(dot(H.normal, ls.Ln)); if (cos_x < kEps) cos_x = 0.0f;
        float cos_y = fabsf(dot(ls.nl, -ls.Ln)); if (cos_y < kEps) cos_y = 0.0f;
        f3 f0, f1; float q0, q1; const float pd = mv.pd, ps = mv.ps;
        lobes_dev_v(sc.mats[H.mat], flags, mv, H.normal, ls.Ln, f0, f1, q0, q1);
        const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
        const float pdf_bsdf = safe_mul(pd, q0) + safe_mul(ps, q1);
        float pdf_light = 1.0f;
        if (cos_y > 0.0f) pdf_light = maxf_(kEps, ls.pdf_l) * ls.dist2 / cos_y;
        const float a_pdf = H.acc_pdf * pdf_light;
        const f3 thr = mk3(F.x * cos_x * 1.0f, F.y * cos_x * 1.0f, F.z * cos_x * 1.0f);
        const f3 a_l = mk3(H.acc_f.x * thr.x, H.acc_f.y * thr.y, H.acc_f.z * thr.z);
        const f3 contribution = a_pdf > 0.0f ? mk3(ls.em.x * a_l.x / a_pdf, ls.em.y * a_l.y / a_pdf, ls.em.z * a_l.z / a_pdf) : mk3(0, 0, 0);
        const float mi = pdf_light / ((float)nee * pdf_light + pdf_bsdf);
        const f3 E_rec = mk3(H.acc_f_rec.x * mi * ls.em.x * thr.x, H.acc_f_rec.y * mi * ls.em.y * thr.y, H.acc_f_rec.z * mi * ls.em.z * thr.z);
        const f3 E_path = contribution * mi;
        float wi = length(E_path);
        H.acc_L = H.acc_L + E_path;
        if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
        if (gi_update(H, wi, s0, s1)) sel(half3_dev(E_rec), true, H.origin + normalize(H.normal) * kSBias, ls.sp);
    }
    const uint32_t st = select_strategy(sc.mats[H.mat], H.outgoing, H.normal, flags, s0, s1);
    return sample_bsdf(sc.mats[H.mat], st, H.outgoing, H.normal, s0, s1);
}
// loop body after its ray (:217-269).  false: the path ends here (miss, or a light was reached).  sel(L2, false, -, -) when the light hit is selected
template <class Sel>
__device__ __forceinline__ bool gi_back(const DevScene& sc, uint32_t flags, uint32_t nee, GiHot& H, f3 smp, bool hit, const Surf& h, uint32_t& s0, uint32_t& s1, Sel&& sel) {
    if (!hit) return false;
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(sc.mats[H.mat], flags, H.normal, smp, normalize(H.outgoing), H.outgoing, f0, f1, q0, q1, pd, ps);
    const f3 F = safe_mul(pd, f0) + safe_mul(ps, f1);
    const float pdf_bsdf = safe_mul(pd, q0) + safe_mul(ps, q1);
    const float NdotL = dot(H.normal, smp);
    const MatGPU& mk = sc.mats[h.mat];
    const f3 thr = mk3(F.x * NdotL, F.y * NdotL, F.z * NdotL);
    H.acc_pdf *= pdf_bsdf;
    H.acc_f = mk3(H.acc_f.x * thr.x, H.acc_f.y * thr.y, H.acc_f.z * thr.z);
    H.acc_f_rec = mk3(H.acc_f_rec.x * thr.x, H.acc_f_rec.y * thr.y, H.acc_f_rec.z * thr.z);
    // outgoing = -sample (Path_Sampler_v6.hlsl:263-269).  Assigned HERE, after its last use of this iteration, not with origin / mat / normal at the end: in the
    // one-function form of round 1, placed there, hipcc (ROCm 7.2, gfx950) dropped the update on the path "light hit whose contribution is zero, fall through" and
    // the next iteration sampled with the stale direction (found by the ReSTIR fuzz: ray counts off by one in 3 % of random scenes while every buffer stayed
    // byte-identical, because such paths carry zero weight).
    H.outgoing = -smp;
    if (mk.Ke_len > 0.0f) {
        const float dist = length(h.pos - H.origin), dist2 = dist * dist;
        const float cos_t = dot(h.normal, -smp);
        const float pdf_light = sc.nlights ? (((mk.Ke[0] + mk.Ke[1] + mk.Ke[2]) / 3.0f) / sc.total_weight) * dist2 / cos_t : 0.0f;
        const f3 contribution = mk3(mk.Ke[0] * H.acc_f.x / H.acc_pdf, mk.Ke[1] * H.acc_f.y / H.acc_pdf, mk.Ke[2] * H.acc_f.z / H.acc_pdf);
        if (length(contribution) > 0.0f) {
            const float mi = pdf_bsdf / ((float)nee * pdf_light + pdf_bsdf);
            const f3 E_rec = mk3(H.acc_f_rec.x * mi * mk.Ke[0], H.acc_f_rec.y * mi * mk.Ke[1], H.acc_f_rec.z * mi * mk.Ke[2]);
            const f3 E_path = contribution * mi;
            float wi = length(E_path);
            H.acc_L = H.acc_L + E_path;
            if (is_nan(wi) || is_inf(wi)) wi = 0.0f;
            if (gi_update(H, wi, s0, s1)) sel(half3_dev(E_rec), false, mk3(0, 0, 0), mk3(0, 0, 0));
            return false;
        }
    }
    H.origin = h.pos; H.mat = h.mat; H.normal = h.normal;
    return true;
}
// the final shadow ray of the selected reconnection (:271-283): cast iff nee > 0 and the end points differ
__device__ __forceinline__ bool gi_final_ray(uint32_t nee, f3 x1s, f3 x2s, F4& so, F4& sd) {
    if (!(nee > 0 && length(x2s - x1s) > kEps)) return false;
    const f3 dv = x2s - x1s, dn = normalize(dv);
    so = {x1s.x, x1s.y, x1s.z, 0.5f * kSBias};
    sd = {dn.x, dn.y, dn.z, maxf_(kSBias, length(dv) - kSBias * 5.0f)};
    return true;
}
// pass1:150-186 after the path sampler: the GI reservoir's W at the primary hit; rgi = (x2, n2, L2, w_sum) on entry
__device__ __forceinline__ void gi_finish(const MatGPU& m, uint32_t flags, f3 x1, f3 n1, f3 ov, Res& rgi) {
    const f3 dg = rgi.x2 - x1;
    const float cos1 = fabsf(dot(n1, normalize(dg)));
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(m, flags, n1, normalize(dg), normalize(ov), normalize(ov), f0, f1, q0, q1, pd, ps);
    const f3 Fx = safe_mul(pd, f0) + safe_mul(ps, f1);
    f3 fr = mk3(Fx.x * cos1 * rgi.L2.x, Fx.y * cos1 * rgi.L2.y, Fx.z * cos1 * rgi.L2.z);
    if (!finite3(fr)) fr = mk3(0, 0, 0);
    const float fc = length(fr);
    rgi.W = fc > kEps ? rgi.w_sum / fc : 0.0f;
    rgi.M = 1;
}
// the visibility ray of GetP_Hat / VisibilityCheck (Sampler_v6.hlsl:86-104): from x1 (lifted by s_bias along n1) towards x2
__device__ __forceinline__ void vis_ray(f3 x1, f3 n1, f3 x2, F4& so, F4& sd) {
    const f3 dv = x2 - x1, o = x1 + normalize(n1) * kSBias, dn = normalize(dv);
    so = {o.x, o.y, o.z, 0.0f};
    sd = {dn.x, dn.y, dn.z, maxf_(length(dv) - 10.0f * kSBias, 2.0f * kSBias)};
}

__device__ void sample_ris_dev(P1Ctx& C, uint32_t M1, f3 outgoing, Res& rs, const Surf& pay, uint32_t& s0, uint32_t& s1) {
    const f3 smp = ris_front(*C.sc, C.flags, M1, outgoing, rs, pay.pos, pay.normal, pay.mat, s0, s1);
    Surf h2;
    C.cnt_ext++;
    const bool hit = p1_hit(C, pay.pos, smp, kSBias, h2);
    ris_back(*C.sc, C.flags, M1, outgoing, rs, pay.pos, pay.normal, pay.mat, smp, hit, h2, s0, s1);
}

__device__ f3 sample_path_simple_dev(P1Ctx& C, const DevFrame& f, Res& rs, f3 init_point, f3 init_normal, f3 init_outgoing, uint32_t init_mat, uint32_t& s0, uint32_t& s1) {
    const DevScene& sc = *C.sc; const uint32_t flags = C.flags;
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    GiHot H; gi_begin(H, init_point, init_normal, init_outgoing, init_mat);
    H.w_sum = rs.w_sum;
    f3 x1s = mk3(0, 0, 0), x2s = mk3(0, 0, 0);
    {
        const f3 smp = gi_first_sample(sc, flags, H, s0, s1);
        Surf h;
        C.cnt_ext++;
        const bool hit = p1_hit(C, H.origin, smp, kSBias, h);
        if (!gi_first_hit(sc, flags, H, smp, hit, h)) return mk3(0, 0, 0);
    }
    const f3 xn = H.origin, nn = normalize(H.normal);
    auto sel = [&](f3 L2h, bool light, f3 a, f3 b) { rs.x2 = xn; rs.n2 = normalize(nn); rs.L2 = L2h; if (light) { x1s = a; x2s = b; } };
    for (uint32_t i = 0; i < f.max_bounces; i++) {
        const f3 smp = gi_front(sc, flags, nee, H, s0, s1, sel);
        Surf h;
        C.cnt_ext++;
        const bool hit = p1_hit(C, H.origin, smp, kSBias, h);
        if (!gi_back(sc, flags, nee, H, smp, hit, h, s0, s1, sel)) break;
    }
    rs.w_sum = H.w_sum;
    F4 so, sd;
    if (gi_final_ray(nee, x1s, x2s, so, sd)) {
        if (p1_any(C, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, sd.w)) rs.w_sum *= 0.0f;
        else rs.w_sum *= 1.0f;
    }
    return H.acc_L;
}

__device__ __forceinline__ uint32_t map_pixel_id(uint32_t w, uint32_t x, uint32_t y) {    // Common_v6.hlsl:173-198
    const uint32_t tcx = (w + 3u) >> 2;
    return ((y >> 2) * tcx + (x >> 2)) * 16u + (y & 3u) * 4u + (x & 3u);
}
__device__ __forceinline__ void store_res(uint32_t* dst, const Res& r) {                  // 40 bytes = 10 dwords
    dst[0] = f2u(r.x2.x); dst[1] = f2u(r.x2.y); dst[2] = f2u(r.x2.z); dst[3] = f2u(r.w_sum);
    dst[4] = f2u(r.n2.x); dst[5] = f2u(r.n2.y); dst[6] = f2u(r.n2.z); dst[7] = f2u(r.W);
    dst[8] = half_bits_dev(r.L2.x) | (half_bits_dev(r.L2.y) << 16); dst[9] = half_bits_dev(r.L2.z) | ((r.M & 0xFFFFu) << 16);
}

// Work items of the thread-per-pixel passes: the shard's own tiles (slot_to_pixel), or — ReSTIR on shards, passes 1 and 2 — an explicit pixel list
// (x | y << 16): the shard's tiles DILATED by the 20-pixel radius of the spatial pass, whose neighbour reads (RayGen_v6_pass3.hlsl:46-372) must find
// this frame's pass-1 / pass-2 records of pixels that other shards own.  The halo is recomputed, not exchanged (seeds depend on the pixel only).
__device__ __forceinline__ bool pass_pixel(const DevFrame& f, const uint32_t* __restrict__ pixels, uint32_t i, uint32_t& x, uint32_t& y) {
    if (pixels) { const uint32_t v = pixels[i]; x = v & 0xFFFFu; y = v >> 16; return true; }
    return slot_to_pixel(f, i, x, y);
}

__global__ __launch_bounds__(kBlock) void k_v6_pass1(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, uint32_t sample_id,
                                                     F4* __restrict__ accum, uint32_t* __restrict__ res_di, uint32_t* __restrict__ res_gi, uint32_t* __restrict__ sdata,
                                                     unsigned long long* __restrict__ counters /* primary, extension, shadow */,
                                                     const uint32_t* __restrict__ pixels = nullptr, uint32_t npixels = 0) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_prim = 0, n_ext = 0, n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t nitems = pixels ? npixels : f.npl;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < nitems; pl += stride) {
        uint32_t x, y;
        if (!pass_pixel(f, pixels, pl, x, y)) continue;
        uint32_t s0, s1; seed_init(x, y, sample_id, f.frame_seed, s0, s1);
        f3 origin, dir; primary_ray(cam, f.width, f.height, x, y, 0.0f, 0.0f, origin, dir);      // jitter = 0, pass1:80-82
        Res rdi; rdi.x2 = mk3(0, 0, 0); rdi.w_sum = 0.0f; rdi.n2 = mk3(0, 0, 0); rdi.W = 0.0f; rdi.L2 = mk3(0, 0, 0); rdi.M = 0;
        Res rgi = rdi;
        f3 x1 = mk3(0, 0, 0), n1 = mk3(0, 0, 0), ov = mk3(0, 0, 0), debug = mk3(0, 0, 0), L1 = mk3(0, 0, 0), out = mk3(0, 0, 0);
        uint32_t mID = kMissMat, objID = 0;
        P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
        Surf pay;
        n_prim++;
        if (p1_hit(C, origin, dir, kTMinCam, pay)) {
            mID = pay.mat; objID = pay.inst;
            const MatGPU& m = sc.mats[mID];
            L1 = mk3(m.Ke[0], m.Ke[1], m.Ke[2]);
            if (!(m.KeFullLen > 0.0f)) {                                                         // performSampling, pass1:102-106
                const f3 outgoing = -dir;
                sample_ris_dev(C, sc.nlights ? f.nee_samples : 0u, outgoing, rdi, pay, s0, s1);
                x1 = pay.pos; n1 = normalize(pay.normal); ov = outgoing;
                const float f_g = length(reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov));
                const f3 dv = rdi.x2 - x1;
                const float vis = p1_any(C, x1 + normalize(n1) * kSBias, normalize(dv), 0.0f, maxf_(length(dv) - 10.0f * kSBias, 2.0f * kSBias)) ? 0.0f : 1.0f;
                const float p_hat = f_g * vis;
                rdi.W = p_hat > kEps ? rdi.w_sum / p_hat : 0.0f;
                debug = sample_path_simple_dev(C, f, rgi, pay.pos, pay.normal, outgoing, mID, s0, s1);
                const f3 rc = reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov);
                debug = debug + rc * rdi.W;
                gi_finish(m, f.flags, x1, n1, ov, rgi);
                out = debug;
            } else out = L1;
        }
        n_ext += C.cnt_ext; n_sh += C.cnt_sh;
        const size_t slot = map_pixel_id(f.width, x, y);
        store_res(res_di + slot * 10, rdi);
        store_res(res_gi + slot * 10, rgi);
        uint32_t* d = sdata + slot * 15;                                                        // 60 bytes: Reservoir_v6.hlsl:2-11
        d[0] = f2u(x1.x); d[1] = f2u(x1.y); d[2] = f2u(x1.z);
        d[3] = (mID & 0xFFFFu) | (half_bits_dev(L1.x) << 16); d[4] = half_bits_dev(L1.y) | (half_bits_dev(L1.z) << 16);
        d[5] = f2u(n1.x); d[6] = f2u(n1.y); d[7] = f2u(n1.z); d[8] = f2u(ov.x); d[9] = f2u(ov.y); d[10] = f2u(ov.z);
        d[11] = objID; d[12] = f2u(debug.x); d[13] = f2u(debug.y); d[14] = f2u(debug.z);
        if (finite3(out)) { F4 a = accum[(size_t)y * f.width + x]; a.x = a.x + out.x; a.y = a.y + out.y; a.z = a.z + out.z; a.w = a.w + 1.0f; accum[(size_t)y * f.width + x] = a; }
    }
    atomicAdd(&counters[0], (unsigned long long)n_prim); atomicAdd(&counters[1], (unsigned long long)n_ext); atomicAdd(&counters[2], (unsigned long long)n_sh);
}

// ---------------------------------------------------------------------------------------------
// ReSTIR temporal reuse (pass 2, RayGen_v6_pass2.hlsl:46-204) and spatial reuse + final shade (pass 3,
// RayGen_v6_pass3.hlsl:46-441) with the pairwise MIS of MIS_v6.hlsl / MIS_GI_v6.hlsl, on the reference's packed
// buffers; one thread per pixel like the reference's raygen shaders.  Statement order = oracle/rt_oracle.c.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_to_float_dev(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) { const float f = (float)m * (1.0f / 16777216.0f); return u2f(f2u(f) | sign); }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
struct SData { f3 x1; uint32_t mID; f3 L1; f3 n1; f3 o; uint32_t objID; };
__device__ __forceinline__ Res load_res_dev(const uint32_t* p) {
    Res r;
    r.x2 = mk3(u2f(p[0]), u2f(p[1]), u2f(p[2])); r.w_sum = u2f(p[3]); r.n2 = mk3(u2f(p[4]), u2f(p[5]), u2f(p[6])); r.W = u2f(p[7]);
    r.L2 = mk3(half_to_float_dev(p[8] & 0xFFFFu), half_to_float_dev(p[8] >> 16), half_to_float_dev(p[9] & 0xFFFFu)); r.M = p[9] >> 16;
    return r;
}
__device__ __forceinline__ Res zero_res() { Res r; r.x2 = mk3(0, 0, 0); r.w_sum = 0.0f; r.n2 = mk3(0, 0, 0); r.W = 0.0f; r.L2 = mk3(0, 0, 0); r.M = 0; return r; }
__device__ __forceinline__ SData load_sd_dev(const uint32_t* d) {
    SData s;
    s.x1 = mk3(u2f(d[0]), u2f(d[1]), u2f(d[2])); s.mID = d[3] & 0xFFFFu;
    s.L1 = mk3(half_to_float_dev(d[3] >> 16), half_to_float_dev(d[4] & 0xFFFFu), half_to_float_dev(d[4] >> 16));
    s.n1 = mk3(u2f(d[5]), u2f(d[6]), u2f(d[7])); s.o = mk3(u2f(d[8]), u2f(d[9]), u2f(d[10])); s.objID = d[11];
    return s;
}
__device__ __forceinline__ SData zero_sd() { SData s; s.x1 = mk3(0, 0, 0); s.mID = 0; s.L1 = mk3(0, 0, 0); s.n1 = mk3(0, 0, 0); s.o = mk3(0, 0, 0); s.objID = 0; return s; }
__device__ __forceinline__ float minf_u(float cap, uint32_t m) { return (float)(m < (uint32_t)cap ? m : (uint32_t)cap); }

// GetP_Hat / GetP_Hat_GI (Sampler_v6.hlsl:163-171, MIS_GI_v6.hlsl) with the visibility term behind a functor: vis(k, x1, n1, x2) -> 1.0f (visible) or 0.0f, k = the
// number of this ray among the pixel's rays of the pass (k < 0: no visibility term).  The thread-per-pixel kernels trace inside the functor (VisTrace); the
// wavefront stages trace all rays of a pass in one persistent launch beforehand and look the answer up (rtx_restir_wave.hpp: VisLookup).
template <class V>
__device__ __forceinline__ float p_hat_di(uint32_t flags, const MatGPU& m, f3 x1, f3 n1, f3 x2, f3 n2, f3 L2, f3 o, int k, V&& vis) {
    const float f_g = length(reconnect_di_dev(m, flags, x1, n1, x2, n2, L2, o));
    float v = 1.0f;
    if (k >= 0) v = vis(k, x1, n1, x2);
    return f_g * v;
}
// the GI target function without its visibility factor (lobes, cosine, radiance; non-finite -> 0)
__device__ __forceinline__ f3 p_hat_gi_fr(uint32_t flags, const MatGPU& m, f3 x1, f3 n1, f3 x2, f3 L, f3 o) {
    const f3 dir = x2 - x1;
    const float cos1 = fabsf(dot(n1, normalize(dir)));
    f3 f0, f1; float q0, q1, pd, ps;
    lobes_dev(m, flags, n1, normalize(dir), normalize(o), normalize(o), f0, f1, q0, q1, pd, ps);
    const f3 Fx = safe_mul(pd, f0) + safe_mul(ps, f1);
    f3 fr = mk3(Fx.x * cos1 * L.x, Fx.y * cos1 * L.y, Fx.z * cos1 * L.z);
    if (!finite3(fr)) fr = mk3(0, 0, 0);
    return fr;
}
__device__ __forceinline__ f3 p_hat_gi_fr_v(uint32_t flags, const MatGPU& m, const MixView& mv, f3 x1, f3 n1, f3 x2, f3 L) {      // ... with mv = mix_view(m, flags, n1, normalize(o))
    const f3 dir = x2 - x1;
    const float cos1 = fabsf(dot(n1, normalize(dir)));
    f3 f0, f1; float q0, q1;
    lobes_dev_v(m, flags, mv, n1, normalize(dir), f0, f1, q0, q1);
    const f3 Fx = safe_mul(mv.pd, f0) + safe_mul(mv.ps, f1);
    f3 fr = mk3(Fx.x * cos1 * L.x, Fx.y * cos1 * L.y, Fx.z * cos1 * L.z);
    if (!finite3(fr)) fr = mk3(0, 0, 0);
    return fr;
}
template <class V>
__device__ __forceinline__ f3 p_hat_gi(uint32_t flags, const MatGPU& m, f3 x1, f3 n1, f3 x2, f3 L, f3 o, int k, V&& vis) {
    const f3 fr = p_hat_gi_fr(flags, m, x1, n1, x2, L, o);
    float v = 1.0f;
    if (k >= 0) v = vis(k, x1, n1, x2);
    return fr * v;
}
struct VisTrace {            // the literal form: one any-hit traversal per call, inside the pixel's thread
    P1Ctx& C;
    __device__ __forceinline__ float operator()(int, f3 x1, f3 n1, f3 x2) const {
        F4 so, sd; vis_ray(x1, n1, x2, so, sd);
        return p1_any(C, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, sd.w) ? 0.0f : 1.0f;
    }
};
__device__ __forceinline__ float get_w_dev(float w_sum, float p_hat) { return p_hat > kEps ? w_sum / p_hat : 0.0f; }
__device__ __forceinline__ float jacobian_dev(const SData& r, const SData& q, f3 x2q, f3 n2q) {
    const f3 vq = x2q - q.x1, vr = x2q - r.x1;
    const float cq = fabsf(dot(normalize(-vq), normalize(n2q))), cr = fabsf(dot(normalize(-vr), normalize(n2q)));
    return (cq / cr) * (dot(vr, vr) / dot(vq, vq));
}
__device__ __forceinline__ bool valid_res_dev(const Res& r) { return length(r.n2) > 0.0f && length(r.L2) > 0.0f && r.w_sum > 0.0f && r.M > 0; }
__device__ __forceinline__ bool valid_res_gi_dev(const Res& r) { return r.w_sum > 0.0f && r.M > 0; }
__device__ __forceinline__ bool reject_distance_dev(f3 x1, f3 x2, f3 cam, float thr) {
    const float d1 = length(x1 - cam), d2 = length(x2 - cam);
    return fabsf(d1 - d2) / maxf_(d1, d2) > thr;
}
__device__ __forceinline__ bool reject_jacobian_dev(float J, float thr) { return J > thr || J < 1.0f / thr || is_nan(J) || is_inf(J); }
__device__ __forceinline__ f3 mul44_dev(const float* m, f3 p, float w, float& ow) {
    ow = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * w;
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * w, m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * w, m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * w);
}
__device__ __forceinline__ void random_pixel_dev(uint32_t radius, uint32_t w, uint32_t h, uint32_t x, uint32_t y, uint32_t& s0, uint32_t& s1, int& ox, int& oy) {
    int nx, ny;
    do {
        const float u = tea_next(s0, s1);
        const float r = (float)radius * u;
        const float ang = tea_next(s0, s1) * 6.2831853f;
        float sn, cs; sincos_(ang, sn, cs);
        nx = (int)x + (int)(cs * r); ny = (int)y + (int)(sn * r);
        while (nx < 0 || nx >= (int)w) { if (nx < 0) nx = -nx; else nx = 2 * (int)w - nx - 2; }
        while (ny < 0 || ny >= (int)h) { if (ny < 0) ny = -ny; else ny = 2 * (int)h - ny - 2; }
    } while (nx == (int)x && ny == (int)y);
    ox = nx; oy = ny;
}

struct RestirBufs { uint32_t *cur_di, *cur_gi, *cur_sd, *last_di, *last_gi, *last_sd; };

// ---- pass 2 (RayGen_v6_pass2.hlsl:46-204) in two parts: what the pixel reads and whether it merges at all (p2_gather), then the two pairwise-MIS merges (p2_merge).
// Its visibility rays: k = 0 DI (x1 -> last frame's x2), k = 1 GI (x1 -> last frame's GI x2); each is cast iff the merge it belongs to runs.
struct P2Pix { size_t slot; Res rc, gc, rl, gl; SData sd; bool acc_di, acc_gi; };
// with_cur = false: only what decides the rays is read (the wavefront emit stage runs before this frame's reservoirs of the pixel are complete)
__device__ __forceinline__ bool p2_gather(const DevScene& sc, const DevFrame& f, const CameraGPU& cam, const RestirBufs& B, uint32_t x, uint32_t y, P2Pix& I, bool with_cur = true) {
    I.slot = map_pixel_id(f.width, x, y);
    I.sd = load_sd_dev(B.cur_sd + I.slot * 15);
    const SData& sd = I.sd;
    if (!(sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f) || sd.mID == 0xFFFEu || sd.mID >= sc.nmat) return false;
    if (with_cur) { I.rc = load_res_dev(B.cur_di + I.slot * 10); I.gc = load_res_dev(B.cur_gi + I.slot * 10); }
    const f3 camo = mk3(cam.viewI[12], cam.viewI[13], cam.viewI[14]);
    int px, py;
    {   // GetBestReprojectedPixel_d, Sampler_v6.hlsl:738-785
        float w0, w1, w2, w3;
        const InstGPU& in = sc.insts[sd.objID < sc.ninst ? sd.objID : 0u];
        const f3 lp = mul44_dev(in.o2w_inv, sd.x1, 1.0f, w0);
        const f3 pw = mul44_dev(in.prev_o2w, lp, w0, w1);
        const f3 vp = mul44_dev(cam.prev_view, pw, w1, w2);
        const f3 cp = mul44_dev(cam.prev_proj, vp, w2, w3);
        if (w3 <= 0.0f) { px = -1; py = -1; }
        else { const float ux = (cp.x / w3) * 0.5f + 0.5f; float uy = (cp.y / w3) * 0.5f + 0.5f; uy = 1.0f - uy; px = (int)rintf(ux * (float)f.width); py = (int)rintf(uy * (float)f.height); }
    }
    const bool inside = px >= 0 && py >= 0 && px < (int)f.width && py < (int)f.height;
    // on shards: a read of the history where this context does not hold it (outside its rectangle + the exchanged halo) is counted, once per pixel (rtx_stats.restir_stale_history_reads)
    if (with_cur && f.hist_stale && inside && ((uint32_t)px < f.hist_x0 || (uint32_t)px >= f.hist_x1 || (uint32_t)py < f.hist_y0 || (uint32_t)py >= f.hist_y1)) atomicAdd(f.hist_stale, 1ull);
    const size_t ts = inside ? map_pixel_id(f.width, (uint32_t)px, (uint32_t)py) : 0;
    I.rl = inside ? load_res_dev(B.last_di + ts * 10) : zero_res(); I.gl = inside ? load_res_dev(B.last_gi + ts * 10) : zero_res();
    const SData sl = inside ? load_sd_dev(B.last_sd + ts * 15) : zero_sd();
    const bool base_ok = (px != -1 && py != -1) && length(sl.L1) == 0.0f && !reject_distance_dev(sd.x1, sl.x1, camo, 0.1f) && sl.mID == sd.mID;
    I.acc_di = base_ok && valid_res_dev(I.rl) && (I.rl.x2.x != 0.0f && I.rl.x2.y != 0.0f && I.rl.x2.z != 0.0f);
    I.acc_gi = base_ok && !(I.gl.w_sum > 5.0f) && valid_res_gi_dev(I.gl);
    return true;
}
template <class V>
__device__ __forceinline__ void p2_merge(const DevScene& sc, const DevFrame& f, const RestirBufs& B, uint32_t x, uint32_t y, P2Pix& I, V&& vis) {
    Res &rc = I.rc, &gc = I.gc; const Res &rl = I.rl, &gl = I.gl; const SData& sd = I.sd;
    const uint32_t flags = f.flags;
    uint32_t s0, s1; seed_init(x, y, 2u, f.frame_seed, s0, s1);
    const MatGPU& m = sc.mats[sd.mID];
    if (I.acc_di) {
        const float mc = minf_u(16.0f, rc.M), ml = minf_u(16.0f, rl.M), M_sum = mc + ml;
        float mi_c = mc / M_sum;
        { const float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
        float mi_t;
        { const float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
        if (length(rl.n2) == 0.0f) { mi_c = 1.0f; mi_t = 0.0f; }
        // (round 4) the target function of the sample that ends up selected was evaluated above — for the current sample as w_c's factor, for the temporal one as w_t's factor
        // before its visibility: kept instead of evaluated a third time (same operands, same bits)
        const float fg_c = length(reconnect_di_dev(m, flags, sd.x1, sd.n1, rc.x2, rc.n2, rc.L2, sd.o));
        const float fg_t = length(reconnect_di_dev(m, flags, sd.x1, sd.n1, rl.x2, rl.n2, rl.L2, sd.o));
        const float w_c = mi_c * (fg_c * 1.0f) * rc.W;
        const float w_t = mi_t * (fg_t * vis(0, sd.x1, sd.n1, rl.x2)) * rl.W;
        rc.M = (uint32_t)mc; rc.w_sum = w_c;
        rc.w_sum += w_t; rc.M = (rc.M + (uint32_t)ml) & 0xFFFFu;
        float p_hat = fg_c;
        if (tea_next(s0, s1) < w_t / rc.w_sum) { rc.x2 = rl.x2; rc.n2 = rl.n2; rc.L2 = rl.L2; p_hat = fg_t; }
        rc.W = get_w_dev(rc.w_sum, p_hat);
    }
    if (I.acc_gi) {
        const float mc = minf_u(16.0f, gc.M), ml = minf_u(16.0f, gl.M), M_sum = mc + ml;
        float mi_c = mc / M_sum;
        { const float m_num = mc, m_den = m_num + (M_sum - mc); if (m_den > 0.0f) mi_c += (ml / M_sum) * (m_num / m_den); }
        float mi_t;
        { const float m_num = M_sum - mc, m_den = m_num + mc; mi_t = m_den > 0.0f ? (ml / M_sum) * m_num / m_den : 0.0f; }
        const f3 fr_c = p_hat_gi_fr(flags, m, sd.x1, sd.n1, gc.x2, gc.L2, sd.o), fr_t = p_hat_gi_fr(flags, m, sd.x1, sd.n1, gl.x2, gl.L2, sd.o);
        const float lf_c = length(fr_c * 1.0f);
        const float w_c = mi_c * lf_c * gc.W;
        const float w_t = mi_t * length(fr_t * vis(1, sd.x1, sd.n1, gl.x2)) * gl.W;
        gc.M = (uint32_t)mc; gc.w_sum = w_c;
        gc.w_sum += w_t; gc.M = (gc.M + (uint32_t)ml) & 0xFFFFu;
        float p_hat = lf_c;
        if (tea_next(s0, s1) < w_t / gc.w_sum) { gc.x2 = gl.x2; gc.n2 = gl.n2; gc.L2 = gl.L2; p_hat = length(fr_t * 1.0f); }
        gc.W = get_w_dev(gc.w_sum, p_hat);
    }
    store_res(B.cur_di + I.slot * 10, rc); store_res(B.cur_gi + I.slot * 10, gc);
}

__global__ __launch_bounds__(kBlock) void k_restir_pass2(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, RestirBufs B,
                                                         unsigned long long* __restrict__ counters, const uint32_t* __restrict__ pixels = nullptr, uint32_t npixels = 0) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t nitems = pixels ? npixels : f.npl;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < nitems; pl += stride) {
        uint32_t x, y;
        if (!pass_pixel(f, pixels, pl, x, y)) continue;
        P2Pix I;
        if (!p2_gather(sc, f, cam, B, x, y, I)) continue;
        P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
        p2_merge(sc, f, B, x, y, I, VisTrace{C});
        n_sh += C.cnt_sh;
    }
    atomicAdd(&counters[2], (unsigned long long)n_sh);
}

// ---- pass 3 (RayGen_v6_pass3.hlsl:46-441) in three parts: the neighbour search (p3_select: all the random numbers but the merges' own), the two generalized
// pairwise-MIS merges (p3_merge) and the final shade (p3_shade).  Visibility rays of a pixel, in the order the reference casts them:
//   k = 0..2  DI, canonical sample seen from neighbour j      (neighbour's x1 -> my x2)         k = 3..5  the same for GI
//   k = 6..8  GI, neighbour v's sample seen from this pixel   (my x1 -> neighbour's GI x2)      k = 9     the selected DI sample (my x1 -> its x2)
struct P3Cand { uint32_t di[3], gi[3]; int n_di, n_gi; float M_sum_DI, M_sum_GI; };
__device__ __forceinline__ void p3_select(const DevScene& sc, const DevFrame& f, const RestirBufs& B, f3 camo, uint32_t x, uint32_t y, const SData& sd, const MatGPU& m,
                                          const Res& rcur, const Res& gcur, uint32_t& s0, uint32_t& s1, P3Cand& K) {
    const uint32_t W = f.width, H = f.height;
    K.n_di = 0; K.n_gi = 0;
    K.M_sum_DI = minf_u(128.0f, rcur.M); K.M_sum_GI = minf_u(128.0f, gcur.M);
    for (int a = 0; a < 9 && K.n_di < 3; a++) {
        int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
        const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        const SData sn = load_sd_dev(B.cur_sd + pr * 15); const Res rn = load_res_dev(B.cur_di + pr * 10);
        const bool ok = !(dot(sd.n1, sn.n1) < 0.9f) && !reject_distance_dev(sd.x1, sn.x1, camo, 0.1f) && valid_res_dev(rn) && length(sn.L1) == 0.0f && sn.mID == sd.mID;
        if (ok) { K.di[K.n_di++] = (uint32_t)pr; K.M_sum_DI += minf_u(128.0f, rn.M); }
    }
    for (int a = 0; a < 9 && K.n_gi < 3; a++) {
        int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
        const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        const SData sn = load_sd_dev(B.cur_sd + pr * 15); const Res gn = load_res_dev(B.cur_gi + pr * 10);
        const bool ok = m.Pr > 0.3f && !reject_distance_dev(sd.x1, sn.x1, camo, 0.1f) && !(dot(normalize(gn.x2 - sd.x1), sd.n1) < 0.0f) &&
                        !(gn.w_sum > 5.0f) && valid_res_gi_dev(gn) && !reject_jacobian_dev(jacobian_dev(sn, sd, gn.x2, gn.n2), 5.0f) &&
                        length(sn.L1) == 0.0f && sn.mID == sd.mID;
        if (ok) { K.gi[K.n_gi++] = (uint32_t)pr; K.M_sum_GI += minf_u(128.0f, gn.M); }
    }
}
// The two merges are independent but for the random-number stream: the DI merge draws one number per DI candidate, then the GI merge one per GI candidate that takes part.
// rcur / gcur: in = this pixel's reservoir after pass 2, out = merged (w_sum, M, selected sample; W still to be set by p3_shade)
template <class V>
__device__ __forceinline__ void p3_merge_di(const DevScene& sc, const DevFrame& f, const RestirBufs& B, const SData& sd, const MatGPU& m, const P3Cand& K,
                                            Res& rcur, uint32_t& s0, uint32_t& s1, V&& vis) {
    const uint32_t flags = f.flags;
    const float M_sum_DI = K.M_sum_DI;
    const int n_di = K.n_di;
    const Res can = rcur;
    const float cMmin = minf_u(128.0f, can.M), cMmax = M_sum_DI - cMmin;
    // (round 4) every value below that the reference's text evaluates more than once from the same operands is evaluated ONCE and kept: the canonical sample's own target
    // function (p_c: three more times in the text) and, per neighbour, the canonical sample's unshadowed contribution seen from it (f_g: once with the visibility factor in the
    // first loop, once without in the second; f_g * 1.0f is f_g).  Pure functions of the same operands give the same bits; the second loop no longer reads the neighbour's
    // 60-B sample record at all.
    const MixView mv = mix_view(m, flags, sd.n1, normalize(sd.o));                    // the view terms of this pixel's own 1 + n_di evaluations, once
    const float p_c = length(reconnect_di_dev_v(m, flags, mv, sd.x1, sd.n1, can.x2, can.n2, can.L2)) * 1.0f;
    const float c_m_num = cMmin * p_c; float mi_c = cMmin / M_sum_DI;
    float fg0 = 0.0f, fg1 = 0.0f, fg2 = 0.0f;
    for (int j = 0; j < n_di; j++) {
        const SData sn = load_sd_dev(B.cur_sd + (size_t)K.di[j] * 15); const Res rn = load_res_dev(B.cur_di + (size_t)K.di[j] * 10);
        const float nM = minf_u(128.0f, rn.M);
        const float f_g = length(reconnect_di_dev(m, flags, sn.x1, sn.n1, can.x2, can.n2, can.L2, sn.o));
        if (j == 0) fg0 = f_g; else if (j == 1) fg1 = f_g; else fg2 = f_g;
        const float p_from = f_g * vis(j, sn.x1, sn.n1, can.x2);
        const float m_den = c_m_num + (cMmax * p_from);
        if (m_den > 0.0f) mi_c += (nM / M_sum_DI) * (c_m_num / m_den);
    }
    const float w_c = mi_c * p_c * can.W;
    rcur.M = (uint32_t)cMmin; rcur.w_sum = w_c;
    for (int v = 0; v < n_di; v++) {
        const Res rn = load_res_dev(B.cur_di + (size_t)K.di[v] * 10);
        const float pc2 = p_c;
        const float p_from = v == 0 ? fg0 : v == 1 ? fg1 : fg2;
        const float m_num = (M_sum_DI - cMmin) * p_from, m_den = m_num + (cMmin * pc2);
        const float mi_s = m_den > 0.0f ? (minf_u(128.0f, rn.M) / M_sum_DI) * (m_num / m_den) : 0.0f;
        const float w_s = mi_s * (length(reconnect_di_dev_v(m, flags, mv, sd.x1, sd.n1, rn.x2, rn.n2, rn.L2)) * 1.0f) * rn.W;
        rcur.w_sum += w_s; rcur.M = (rcur.M + (uint32_t)minf_u(128.0f, rn.M)) & 0xFFFFu;
        if (tea_next(s0, s1) < w_s / rcur.w_sum) { rcur.x2 = rn.x2; rcur.n2 = rn.n2; rcur.L2 = rn.L2; }
    }
}
template <class V>
__device__ __forceinline__ void p3_merge_gi(const DevScene& sc, const DevFrame& f, const RestirBufs& B, const SData& sd, const MatGPU& m, const P3Cand& K,
                                            Res& gcur, uint32_t& s0, uint32_t& s1, V&& vis) {
    const uint32_t flags = f.flags;
    const float M_sum_GI = K.M_sum_GI;
    const int n_gi = K.n_gi;
    const Res can_gi = gcur;
    const float gMmin = minf_u(128.0f, can_gi.M), gMmax = M_sum_GI - gMmin;
    // (round 4) evaluated once and kept, as in the DI merge: pg_c (twice more in the text) and, per neighbour, the canonical GI sample's unshadowed target function seen from
    // it and the Jacobian of that shift (lf, jac: the second loop's p_from is length(fr * 1.0f) * jj with the same fr and jj).  The second loop reads x1 of the neighbour only.
    const MixView mv = mix_view(m, flags, sd.n1, normalize(sd.o));
    const float pg_c = length(p_hat_gi_fr_v(flags, m, mv, sd.x1, sd.n1, can_gi.x2, can_gi.L2) * 1.0f);
    const float g_m_num = gMmin * pg_c; float mi_c_gi = gMmin / M_sum_GI;
    float lf0 = 0.0f, lf1 = 0.0f, lf2 = 0.0f, jc0 = 0.0f, jc1 = 0.0f, jc2 = 0.0f;
    for (int j = 0; j < n_gi; j++) {
        const SData sn = load_sd_dev(B.cur_sd + (size_t)K.gi[j] * 15); const Res gn = load_res_dev(B.cur_gi + (size_t)K.gi[j] * 10);
        const float nM = minf_u(128.0f, gn.M);
        const float j_gi = jacobian_dev(sd, sn, can_gi.x2, can_gi.n2);
        const f3 fr = p_hat_gi_fr(flags, m, sn.x1, sn.n1, can_gi.x2, can_gi.L2, sn.o);
        const float lf = length(fr);
        if (j == 0) { lf0 = lf; jc0 = j_gi; } else if (j == 1) { lf1 = lf; jc1 = j_gi; } else { lf2 = lf; jc2 = j_gi; }
        const float p_from = length(fr * vis(3 + j, sn.x1, sn.n1, can_gi.x2)) * j_gi;
        const float m_den = g_m_num + (gMmax * p_from);
        if (m_den > 0.0f) mi_c_gi += (nM / M_sum_GI) * (g_m_num / m_den);
    }
    mi_c_gi = minf_(maxf_(mi_c_gi, 0.0f), 1.0f);
    const float w_c_gi = mi_c_gi * pg_c * can_gi.W;
    gcur.M = (uint32_t)gMmin; gcur.w_sum = w_c_gi;
    for (int v = 0; v < n_gi; v++) {
        const SData sn = load_sd_dev(B.cur_sd + (size_t)K.gi[v] * 15); const Res gn = load_res_dev(B.cur_gi + (size_t)K.gi[v] * 10);
        const float pc2 = pg_c;
        const float p_from = (v == 0 ? lf0 : v == 1 ? lf1 : lf2) * (v == 0 ? jc0 : v == 1 ? jc1 : jc2);
        const float m_num = (M_sum_GI - gMmin) * p_from, m_den = m_num + (gMmin * pc2);
        const float mi_s = m_den > 0.0f ? minf_(maxf_((minf_u(128.0f, gn.M) / M_sum_GI) * (m_num / m_den), 0.0f), 1.0f) : 0.0f;
        const float j_gi = jacobian_dev(sn, sd, gn.x2, gn.n2);
        const f3 f_gi = p_hat_gi_fr_v(flags, m, mv, sd.x1, sd.n1, gn.x2, gn.L2) * vis(6 + v, sd.x1, sd.n1, gn.x2);
        const float w_s = mi_s * length(f_gi) * gn.W * j_gi;
        if (j_gi != 0.0f) {
            gcur.w_sum += w_s; gcur.M = (gcur.M + (uint32_t)minf_u(128.0f, gn.M)) & 0xFFFFu;
            if (tea_next(s0, s1) < w_s / gcur.w_sum) { gcur.x2 = gn.x2; gcur.n2 = gn.n2; gcur.L2 = gn.L2; }
        }
    }
}
template <class V>
__device__ __forceinline__ void p3_merge(const DevScene& sc, const DevFrame& f, const RestirBufs& B, const SData& sd, const MatGPU& m, const P3Cand& K,
                                         Res& rcur, Res& gcur, uint32_t& s0, uint32_t& s1, V&& vis) {
    p3_merge_di(sc, f, B, sd, m, K, rcur, s0, s1, vis);
    p3_merge_gi(sc, f, B, sd, m, K, gcur, s0, s1, vis);
}
// final W of both reservoirs and the pixel's radiance ReconnectDI * W + f_gi * W_gi (pass3:353-372)
template <class V>
__device__ __forceinline__ f3 p3_shade(const DevFrame& f, const SData& sd, const MatGPU& m, Res& rcur, Res& gcur, V&& vis) {
    const MixView mv = mix_view(m, f.flags, sd.n1, normalize(sd.o));
    const f3 rc = reconnect_di_dev_v(m, f.flags, mv, sd.x1, sd.n1, rcur.x2, rcur.n2, rcur.L2);       // once: the target function's length and the radiance term
    const float p_hat = length(rc) * vis(9, sd.x1, sd.n1, rcur.x2);
    rcur.W = get_w_dev(rcur.w_sum, p_hat);
    f3 acc = rc * rcur.W;
    const f3 f_fin = p_hat_gi_fr_v(f.flags, m, mv, sd.x1, sd.n1, gcur.x2, gcur.L2) * 1.0f;
    gcur.W = get_w_dev(gcur.w_sum, length(f_fin));
    acc = acc + f_fin * gcur.W;
    return acc;
}

__global__ __launch_bounds__(kBlock) void k_restir_pass3(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, const CameraGPU* __restrict__ cam_p, RestirBufs B,
                                                         F4* __restrict__ accum, unsigned long long* __restrict__ counters) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t n_sh = 0;
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t W = f.width;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        const size_t slot = map_pixel_id(W, x, y);
        const SData sd = load_sd_dev(B.cur_sd + slot * 15);
        f3 out = mk3(0, 0, 0);
        if (!(sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f)) out = sd.L1;                      // pass3:457-462
        else if (!(sd.mID == 0xFFFEu || sd.mID >= sc.nmat)) {
            P1Ctx C; C.sc = &sc; C.small = small; C.L = &L; C.flags = f.flags; C.cnt_ext = 0; C.cnt_sh = 0;
            const f3 camo = mk3(cam.viewI[12], cam.viewI[13], cam.viewI[14]);
            uint32_t s0, s1; seed_init(x, y, 3u, f.frame_seed, s0, s1);
            const MatGPU& m = sc.mats[sd.mID];
            Res rcur = load_res_dev(B.cur_di + slot * 10), gcur = load_res_dev(B.cur_gi + slot * 10);
            P3Cand K;
            p3_select(sc, f, B, camo, x, y, sd, m, rcur, gcur, s0, s1, K);
            p3_merge(sc, f, B, sd, m, K, rcur, gcur, s0, s1, VisTrace{C});
            out = p3_shade(f, sd, m, rcur, gcur, VisTrace{C});
            store_res(B.last_di + slot * 10, rcur); store_res(B.last_gi + slot * 10, gcur);
            for (int k = 0; k < 15; k++) B.last_sd[slot * 15 + k] = B.cur_sd[slot * 15 + k];
            n_sh += C.cnt_sh;
        }
        if (finite3(out)) { F4 a = accum[(size_t)y * W + x]; a.x = a.x + out.x; a.y = a.y + out.y; a.z = a.z + out.z; a.w = a.w + 1.0f; accum[(size_t)y * W + x] = a; }
    }
    atomicAdd(&counters[2], (unsigned long long)n_sh);
}

// ReSTIR on shards: the history a frame leaves behind (u3 / u5 / u7 = g_Reservoirs_last, g_Reservoirs_last_gi, g_sample_last: 40 + 40 + 60 B per pixel) is written by the
// spatial pass for the shard's own pixels only, but the next frame's temporal pass reprojects to ARBITRARY pixels (RayGen_v6_pass2.hlsl:46-204).  So after every frame the
// shards exchange their own tiles' records: pack -> ONE all-gather -> unpack, exactly like the framebuffer tiles (slab: [local slot][35 dwords]).
constexpr uint32_t kStateDwords = 35;       // 10 + 10 + 15
__global__ __launch_bounds__(kBlock) void k_restir_pack_state(DevFrame f, RestirBufs B, uint32_t* __restrict__ slab) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        uint32_t* o = slab + (size_t)pl * kStateDwords;
        if (!slot_to_pixel(f, pl, x, y)) { for (uint32_t k = 0; k < kStateDwords; k++) o[k] = 0u; continue; }
        const size_t slot = map_pixel_id(f.width, x, y);
        for (int k = 0; k < 10; k++) { o[k] = B.last_di[slot * 10 + k]; o[10 + k] = B.last_gi[slot * 10 + k]; }
        for (int k = 0; k < 15; k++) o[20 + k] = B.last_sd[slot * 15 + k];
    }
}
__global__ __launch_bounds__(kBlock) void k_restir_unpack_state(DevFrame f, uint32_t nshards, const uint32_t* __restrict__ slabs, RestirBufs B) {
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t total = f.npl * nshards;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        DevFrame g = f; g.shard_rank = i / f.npl; g.shard_count = nshards;
        uint32_t x, y;
        if (!slot_to_pixel(g, i - g.shard_rank * f.npl, x, y)) continue;
        const size_t slot = map_pixel_id(f.width, x, y);
        const uint32_t* in = slabs + (size_t)i * kStateDwords;
        for (int k = 0; k < 10; k++) { B.last_di[slot * 10 + k] = in[k]; B.last_gi[slot * 10 + k] = in[10 + k]; }
        for (int k = 0; k < 15; k++) B.last_sd[slot * 15 + k] = in[20 + k];
    }
}

// HALO EXCHANGE (rtx.h: rtx_restir_pack_halo): the history records of up to kHaloPeers pixel rectangles, row-major, 35 dwords each, packed back to back into one buffer
// (and scattered from one).  first[k] = index of rectangle k's first record; rectangles are disjoint parts of the own rectangle (pack) or of the peers' (unpack).
struct HaloRects { uint32_t n; uint32_t x0[kHaloPeers], y0[kHaloPeers], w[kHaloPeers]; uint32_t first[kHaloPeers + 1]; };
template <bool PACK>
__global__ __launch_bounds__(kBlock) void k_restir_halo(uint32_t width, HaloRects R, RestirBufs B, uint32_t* __restrict__ buf) {
    const uint32_t stride = gridDim.x * kBlock, total = R.first[R.n];
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t j = 1; j < kHaloPeers; j++) if (j < R.n && i >= R.first[j]) k = j;
        const uint32_t r = i - R.first[k], x = R.x0[k] + r % R.w[k], y = R.y0[k] + r / R.w[k];
        const size_t slot = map_pixel_id(width, x, y);
        uint32_t* o = buf + (size_t)i * kStateDwords;
        if (PACK) {
            for (int q = 0; q < 10; q++) { o[q] = B.last_di[slot * 10 + q]; o[10 + q] = B.last_gi[slot * 10 + q]; }
            for (int q = 0; q < 15; q++) o[20 + q] = B.last_sd[slot * 15 + q];
        } else {
            for (int q = 0; q < 10; q++) { B.last_di[slot * 10 + q] = o[q]; B.last_gi[slot * 10 + q] = o[10 + q]; }
            for (int q = 0; q < 15; q++) B.last_sd[slot * 15 + q] = o[20 + q];
        }
    }
}

}  // namespace rtx
