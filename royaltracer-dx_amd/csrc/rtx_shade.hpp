// rtx_shade.hpp — surface reconstruction and the per-bounce shading functions shared by k_shade and k_bounce_small
#pragma once
#include "rtx_traverse.hpp"

namespace rtx {

// ---------------------------------------------------------------------------------------------
// surface reconstruction: ClosestHit, Hit_v6.hlsl:12-61, from the pre-gathered TriShade record
// ---------------------------------------------------------------------------------------------
struct Surf { f3 pos; f3 normal; uint32_t mat; uint32_t inst; float area; f3 flat; bool near_hull; };   // near_hull: tiny scenes, see TriShade::guard_tau
__device__ __forceinline__ Surf surface(const DevScene& sc, f3 o, f3 d, float t, float u, float v, uint32_t gid) {
    Surf s;
    const F4* rec = (const F4*)(sc.shade + gid);
    const F4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    s.mat = f2u(r0.x); s.inst = f2u(r0.y);
    const f3 flat = mk3(r0.z, r0.w, r1.x);
    const f3 n0 = mk3(r1.y, r1.z, r1.w), n1 = mk3(r2.x, r2.y, r2.z), n2 = mk3(r2.w, r3.x, r3.y);
    s.area = r3.z; s.flat = flat;
    s.near_hull = minf_(1.0f - u - v, minf_(u, v)) < r3.w;
    s.pos = madd3(d, t, o);                                                   // :15,60
    const float b0 = 1.0f - u - v;                                            // :18
    const f3 smooth = lincomb3(n0, b0, n1, u, n2, v);                         // :40-46
    // :49-54  length(smooth) > 0.0001f, without the IEEE sqrt (19 issue slots): correctly rounded sqrt is monotonic, so the comparison is
    // EXACTLY dot >= T*, T* = 0x322bcc78 = the smallest float whose square root rounds above 1e-4f (tests/test_oracle_golden.py)
    const f3 n = (dot(smooth, smooth) >= u2f(0x322bcc78u)) ? normalize(smooth) : flat;
    s.normal = normalize(xform_dir(sc.insts[s.inst].nrm, n));                 // :56
    return s;
}

// ---------------------------------------------------------------------------------------------
// shading building blocks.  Loop body of RayGen.hlsl:99-133 + Hit.hlsl:126-174,340-369 with the v6 leaf math,
// in the same statement order as oracle/rt_oracle.c:trace_path.  Shared by k_shade (separate trace / shade /
// shadow kernels: general BVH scenes) and k_bounce_small (one fused kernel per bounce: tiny scenes).
//
// Per-path state in HBM (48 B read + 48 B written per bounce):
//   ray_o = (origin.xyz, seed.y bits)   ray_d = (dir.xyz, pdf of the sampled direction)   thr = (throughput.xyz, seed.x bits)
// tmin is a function of the bounce index (camera rays 1e-4, pass1:94; later rays s_bias, Sampler_v6.hlsl:226),
// rad = (radiance.xyz, -) is only touched when something is added.
// ---------------------------------------------------------------------------------------------
struct PathState { uint32_t pid; f3 o, d; float prev_pdf; f3 thr; uint32_t s0, s1; };

__device__ __forceinline__ float bounce_tmin(uint32_t bounce) { return bounce == 0 ? kTMinCam : kSBias; }

// STREAMING ACCESSES of the general path.  Path state, hit records, the next bounce's state and the shadow entries are written once and read once per bounce, gigabytes of them
// (33 M paths x 48 B); marked non-temporal they do not displace what IS re-used from L2 — the randomly fetched TriShade / material lines of k_shade (L2 hit 0.44) and the BVH of
// k_trace_closest (C5: 0.62).  Same box, three alternating rounds (make VARIANT=x VARFLAGS=-DRTX_NO_NT_STREAMS, tools/lib_ab.sh): k_shade 6.60 -> 6.23 ms (C3), 7.91 -> 7.79
// (C5); the ray loads / hit stores of k_trace_closest: 19.09 -> 18.86 ms on C5, nothing on C3; frames -0.9 % both.  NOT in k_trace_shadow: its entries marked the same way
// cost 2.6 % there (11.35 -> 11.65 ms, C3).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
#ifdef RTX_NO_NT_STREAMS
constexpr bool kNtStreams = false;
#else
constexpr bool kNtStreams = true;
#endif
__device__ __forceinline__ F4 ld_stream(const F4* p) {
    if (kNtStreams) { const nt_f4 v = __builtin_nontemporal_load((const nt_f4*)p); return F4{v.x, v.y, v.z, v.w}; }
    return *p;
}
__device__ __forceinline__ void st_stream(F4* p, F4 v) {
    if (kNtStreams) { nt_f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (nt_f4*)p); }
    else *p = v;
}
__device__ __forceinline__ PathState load_path_stream(const DevPaths& p, uint32_t pid) {
    PathState S; S.pid = pid;
    const F4 ro = ld_stream(p.ray_o + pid), rd = ld_stream(p.ray_d + pid), tv = ld_stream(p.thr + pid);
    S.o = mk3(ro.x, ro.y, ro.z); S.s1 = f2u(ro.w);
    S.d = mk3(rd.x, rd.y, rd.z); S.prev_pdf = rd.w;
    S.thr = mk3(tv.x, tv.y, tv.z); S.s0 = f2u(tv.w);
    return S;
}
__device__ __forceinline__ PathState load_path(const DevPaths& p, uint32_t pid) {
    PathState S; S.pid = pid;
    const F4 ro = p.ray_o[pid], rd = p.ray_d[pid], tv = p.thr[pid];
    S.o = mk3(ro.x, ro.y, ro.z); S.s1 = f2u(ro.w);
    S.d = mk3(rd.x, rd.y, rd.z); S.prev_pdf = rd.w;
    S.thr = mk3(tv.x, tv.y, tv.z); S.s0 = f2u(tv.w);
    return S;
}

// hit on an emissive surface: Hit.hlsl:126-174 with the v6 pdf conventions (Sampler_v6.hlsl:459-465)
// fresh: the path's radiance slot has not been written yet (bounce 0 of the fused path): start from zero instead of loading it
__device__ __forceinline__ void add_emissive(const DevScene& sc, const DevPaths& p, const PathState& S, const Surf& sf, const MatGPU& m, uint32_t bounce, uint32_t nee, bool fresh = false) {
    const f3 Ke = mk3(m.Ke[0], m.Ke[1], m.Ke[2]);
    F4 radv = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!fresh) radv = p.rad[S.pid];
    if (bounce == 0) { radv.x = radv.x + Ke.x; radv.y = radv.y + Ke.y; radv.z = radv.z + Ke.z; }   // Hit.hlsl:128-131
    else {
        float mi = 1.0f;
        if (nee) {                                            // Path_Sampler_v6.hlsl:241
            const f3 Lv = sf.pos - S.o;
            const float dist = length(Lv), dist2 = dist * dist;
            const float cos_t = fabsf(dot(sf.normal, -S.d));
            const float pdf_light = (((Ke.x + Ke.y + Ke.z) / 3.0f) / sc.total_weight) * dist2 / maxf_(cos_t, kEps);
            mi = S.prev_pdf / ((float)nee * pdf_light + S.prev_pdf);
            if (S.prev_pdf < 0.0f) mi = 1.0f;                 // the ray came through a transmission lobe (sign of the stored pdf): NEE never samples through an interface
        }
        const f3 e = mk3(Ke.x * S.thr.x * mi, Ke.y * S.thr.y * mi, Ke.z * S.thr.z * mi);   // Hit.hlsl:173
        if (finite3(e)) { radv.x = radv.x + e.x; radv.y = radv.y + e.y; radv.z = radv.z + e.z; }
    }
    p.rad[S.pid] = radv;
}

// one NEE sample: SampleLightNEE_GI, Sampler_v6.hlsl:508-647.  Returns true when a shadow ray is needed.
// mv: the view terms of the shading point (mix_view), shared with the continuation's mixture by k_shade; nullptr = computed here
__device__ __forceinline__ bool nee_sample(const DevScene& sc, const MatGPU& m, uint32_t flags, uint32_t nee, PathState& S, f3 pos, f3 normal, f3 outgoing,
                                           F4& so, F4& sd, f3& con, bool near_hull = false, float eta_p = 0.0f, const MixView* mv = nullptr, const float* lds_cdf = nullptr, const LightGPU* lds_lights = nullptr) {
    const float rv = tea_next(S.s0, S.s1);
    int left = 0, right = (int)sc.nlights - 1, sel = 0;
#ifndef RTX_NO_LDS_CDF
    if (lds_cdf) {                                            // (round 5, uniform) k_shade staged the CDF of a short light list in LDS: the same search, its dependent reads from LDS
        while (left <= right) {
            const int mid = left + (right - left) / 2;
            if (rv < lds_cdf[mid]) { sel = mid; right = mid - 1; } else left = mid + 1;
        }
    } else
#endif
    while (left <= right) {                                   // :523-537
        const int mid = left + (right - left) / 2;
        if (rv < sc.cdf[mid]) { sel = mid; right = mid - 1; } else left = mid + 1;
    }
#ifndef RTX_NO_LDS_LIGHTS
    const LightGPU& lt = lds_lights ? lds_lights[sel] : sc.lights[sel];     // (uniform choice; k_shade stages a list of <= 256 records)
#else
    const LightGPU& lt = sc.lights[sel];
#endif
    const f3 xv = mk3(lt.xv[0], lt.xv[1], lt.xv[2]), yv = mk3(lt.yv[0], lt.yv[1], lt.yv[2]), zv = mk3(lt.zv[0], lt.zv[1], lt.zv[2]);
    float xi1 = tea_next(S.s0, S.s1), xi2 = tea_next(S.s0, S.s1);
    if (xi1 + xi2 > 1.0f) { xi1 = 1.0f - xi1; xi2 = 1.0f - xi2; }
    const float u = 1.0f - xi1 - xi2, v = xi1, w = xi2;
    const f3 sp = lincomb3(xv, u, yv, v, zv, w);
    const f3 Lv = sp - pos;
    const float dist2 = dot(Lv, Lv);
    const float dist = sqrtf(maxf_(dist2, kEps));
    const f3 Ln = normalize(Lv);
    f3 nl = mk3(lt.nl[0], lt.nl[1], lt.nl[2]);
    if (dot(nl, -Ln) < 0.0f) nl = -nl;
    const float cos_x = dot(normal, Ln);
    const float cos_y = fabsf(dot(nl, -Ln));
    if (cos_x < kEps || cos_y < kEps) return false;           // :580-585
    const float pdf_light = lt.pdf_l * dist2 / cos_y;         // :629-630
    f3 F; float P, pd, ps;
    if (mv) bsdf_mixture_v(m, flags, *mv, normal, Ln, outgoing, F, P, eta_p);
    else bsdf_mixture(m, flags, normal, Ln, outgoing, F, P, pd, ps, eta_p);     // (Ln is on the reflection side: cos_x >= EPS; eta_p only rescales p_d)
    const float mi = pdf_light / ((float)nee * pdf_light + P);   // Path_Sampler_v6.hlsl:164
    const float g = cos_x / pdf_light * mi;
    con = mk3(lt.em[0] * (S.thr.x * F.x) * g, lt.em[1] * (S.thr.y * F.y) * g, lt.em[2] * (S.thr.z * F.z) * g);
    if (!finite3(con) || is_zero3(con)) return false;
    const f3 sorg = madd3(normalize(normal), kSBias, pos);    // :616-621
    so = {sorg.x, sorg.y, sorg.z, near_hull ? -0.5f * kSBias : 0.5f * kSBias};     // tmin; its sign carries the hull-guard flag of the tiny-scene path
    sd = {Ln.x, Ln.y, Ln.z, maxf_(kSBias, dist - kSBias * 5.0f)};
    return true;
}

// BSDF sampling + throughput + Russian roulette: Path_Sampler_v6.hlsl:205-229, Sampler_v6.hlsl:423-457,482-497,
// Hit.hlsl:366-369, RayGen.hlsl:118-130.  Returns true when the path continues (state updated in S, smp, P).
__device__ __forceinline__ bool bsdf_continue(const MatGPU& m, const DevFrame& f, uint32_t bounce, PathState& S, f3 normal, f3 outgoing, f3& smp, float& P, float eta_p = 0.0f, const MixView* mv = nullptr) {
    const uint32_t st = mv ? select_strategy_v(m, *mv, f.flags, S.s0, S.s1, eta_p) : select_strategy(m, outgoing, normal, f.flags, S.s0, S.s1, eta_p);
    smp = sample_bsdf(m, st, outgoing, normal, S.s0, S.s1, eta_p);
    if (st == 3u && is_zero3(smp)) return false;              // (extension) total internal reflection ends the path
    f3 F; float pd, ps;
    if (mv) bsdf_mixture_v(m, f.flags, *mv, normal, smp, outgoing, F, P, eta_p);
    else bsdf_mixture(m, f.flags, normal, smp, outgoing, F, P, pd, ps, eta_p);
    float NdotL = dot(normal, smp);                           // unclamped, Sampler_v6.hlsl:455
    if (st == 3u) NdotL = fabsf(NdotL);                       // (extension) the transmitted direction lies on the far side
    if (!(P > 0.0f)) return false;
    const float wgt = NdotL / P;                              // Hit.hlsl:366
    S.thr = mk3(S.thr.x * (F.x * wgt), S.thr.y * (F.y * wgt), S.thr.z * (F.z * wgt));
    if (!finite3(S.thr) || is_zero3(S.thr)) return false;
    if (bounce > f.rr_start) {                                // RayGen.hlsl:118-130
        const float mx = maxf_(S.thr.x, maxf_(S.thr.y, S.thr.z));
        const float q = minf_(maxf_(mx, 0.05f), 1.0f);
        const float r = tea_next(S.s0, S.s1);
        if (r > q) return false;
        const float iq = 1.0f / q;
        S.thr = S.thr * iq;
    }
    if (st == 3u) P = -P;                                     // the stored pdf carries "this ray was transmitted" in its sign (add_emissive)
    return true;
}

__device__ __forceinline__ void store_path_at(F4* ray_o, F4* ray_d, F4* thr, uint32_t idx, const PathState& S, f3 pos, f3 smp, float P) {
    ray_o[idx] = {pos.x, pos.y, pos.z, u2f(S.s1)};            // un-offset origin, Sampler_v6.hlsl:224-227
    ray_d[idx] = {smp.x, smp.y, smp.z, P};                    // pdf for the MIS at the next emissive hit, Hit.hlsl:369
    thr[idx] = {S.thr.x, S.thr.y, S.thr.z, u2f(S.s0)};
}
__device__ __forceinline__ void store_path_at_stream(F4* ray_o, F4* ray_d, F4* thr, uint32_t idx, const PathState& S, f3 pos, f3 smp, float P) {
    st_stream(ray_o + idx, F4{pos.x, pos.y, pos.z, u2f(S.s1)});
    st_stream(ray_d + idx, F4{smp.x, smp.y, smp.z, P});
    st_stream(thr + idx, F4{S.thr.x, S.thr.y, S.thr.z, u2f(S.s0)});
}
__device__ __forceinline__ void store_path(const DevPaths& p, const PathState& S, f3 pos, f3 smp, float P) { store_path_at(p.ray_o, p.ray_d, p.thr, S.pid, S, pos, smp, P); }

}  // namespace rtx
