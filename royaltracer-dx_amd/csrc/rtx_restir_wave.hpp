// rtx_restir_wave.hpp — the reference's shipping frame (three DispatchRays: Renderer.cpp:646-673) as WAVEFRONT STAGES on the machinery of the path tracer.
//
// The thread-per-pixel kernels of rtx_restir.hpp run a pixel's whole raygen shader in one thread: up to 8 + 2 + 10 rays, each traversed inside the thread with
// whatever lanes of its wave happen to be at the same call (k_v6_pass1: 228 VGPRs, 2 waves per SIMD; 10.3 + 1.1 + 7.3 ms per 1080p frame on the 262 k-triangle
// atrium, profiles/r03_restir_base.md).  Here every pass is cut at its ray casts:
//   * the pixel programs become short STAGE kernels (<= 128 VGPRs: 4+ waves per SIMD) that end by writing a ray into a workgroup-private sub-queue,
//   * all rays of a stage are traversed by the persistent-wave kernels of the path tracer (k_trace_closest; k_trace_shadow with an occlusion-byte sink), which
//     keep their lanes filled by refilling from the sub-queue,
//   * state travels between stages by QUEUE POSITION in two buffer sets (a stage writes its survivors densely into the other set: full waves, coalesced streams),
//     rare events (a reservoir selection) go straight to per-pixel records.
// Pass 1 is a chain (its random-number stream runs through all of its rays): raygen | ris | ris_finish | first | loop x bounces | emit_final | finish.
// Passes 2 and 3 need no carried state: a visibility ray there depends on buffer contents only, so an EMIT stage writes all rays of the pass, one launch traces
// them into a byte per ray, and the merge stage — the same code as the literal kernel, p2_merge / p3_merge / p3_shade — looks the answers up.  Pass 2 rides on
// pass 1's last two stages (its rays join pass 1's shadow rays in one traversal launch, its merge follows the pixel's finish in the same thread); pass 3 has one
// dependent ray (the selected DI sample), hence select | merge | shade.
// Per-pixel statement order, random-number order and arithmetic are those of rtx_restir.hpp (shared functions), so the six buffers and the image stay byte-equal
// to the oracle's (tests/test_gpu_parity.py: every ReSTIR test runs both forms).
#pragma once
#include "rtx_restir.hpp"

namespace rtx {

// (the work area RsQ and its constants are declared in rtx_kernels.hpp: the host fills it)

// LOGICAL workgroup (= sub-queue) index of this workgroup.  The hardware deals workgroups to the 8 XCDs round-robin (blockIdx.x % 8), each XCD with its own L2; chunks of 256
// pixels are dealt to LOGICAL workgroups in order (chunk c -> workgroup c mod G), and consecutive chunks are neighbouring screen regions.  Mapping the workgroups of one XCD
// to a contiguous range of logical indices (G is a multiple of 8) gives every XCD contiguous strips of the image instead of every eighth chunk — what the neighbour gathers
// of the spatial pass need: records within 20 px are then mostly in the same L2 (k_rs_p3_select fetched 8.4 GB per frame at an L2 hit rate of 0.41 before, profiles/r03_pmc_restir.md).
__device__ __forceinline__ uint32_t rs_wg(const RsQ& q) { return (blockIdx.x & 7u) * (q.G >> 3) + (blockIdx.x >> 3); }
__device__ __forceinline__ bool rs_item_pixel(const DevFrame& f, const RsQ& q, uint32_t it, uint32_t& x, uint32_t& y) { return pass_pixel(f, q.pixels, it, x, y); }
// push one any-hit ray into the workgroup's ray sub-queue (every lane of the wave calls this)
__device__ __forceinline__ void rs_push_ray(const RsQ& q, uint32_t* s_rn, bool pred, const F4& so, const F4& sd, uint32_t pay) {
    const uint32_t slot = block_push(pred, s_rn);
    if (pred) { const size_t gi = (size_t)rs_wg(q) * q.rcap + slot; q.sh_o[gi] = so; q.sh_d[gi] = sd; q.sh_pay[gi] = pay; }
}
struct VisLookup {           // the answer of ray k, traced before this stage
    const uint8_t* occ;
    __device__ __forceinline__ float operator()(int k, f3, f3, f3) const { return occ[k] ? 0.0f : 1.0f; }
};
__device__ __forceinline__ void rs_accumulate(F4* __restrict__ accum, uint32_t W, uint32_t x, uint32_t y, f3 out) {
    if (finite3(out)) { F4 a = accum[(size_t)y * W + x]; a.x = a.x + out.x; a.y = a.y + out.y; a.z = a.z + out.z; a.w = a.w + 1.0f; accum[(size_t)y * W + x] = a; }
}
__device__ __forceinline__ void store_sdata_head(uint32_t* d, f3 x1, uint32_t mID, f3 L1, f3 n1, f3 ov, uint32_t objID) {     // Reservoir_v6.hlsl:2-11 without `debug`
    d[0] = f2u(x1.x); d[1] = f2u(x1.y); d[2] = f2u(x1.z);
    d[3] = (mID & 0xFFFFu) | (half_bits_dev(L1.x) << 16); d[4] = half_bits_dev(L1.y) | (half_bits_dev(L1.z) << 16);
    d[5] = f2u(n1.x); d[6] = f2u(n1.y); d[7] = f2u(n1.z); d[8] = f2u(ov.x); d[9] = f2u(ov.y); d[10] = f2u(ov.z);
    d[11] = objID;
}
// end of a stage kernel: publish the sub-queue length(s) and the ray statistics (one atomic per workgroup and class); ray_base: entries the ray sub-queue held before this launch
__device__ __forceinline__ void rs_publish(uint32_t* cnt_out, const uint32_t* s_n, uint32_t* shcnt, const uint32_t* s_rn, const RsQ& q, int ray_class, uint32_t ray_base = 0u) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (cnt_out) { cnt_out[rs_wg(q)] = *s_n; if (*s_n) atomicAdd(&q.rays[ray_class], (unsigned long long)*s_n); }
        if (shcnt) { shcnt[rs_wg(q)] = *s_rn; if (*s_rn > ray_base) atomicAdd(&q.rays[2], (unsigned long long)(*s_rn - ray_base)); }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// PASS 1 (RayGen_v6_pass1.hlsl:48-190)
// ---------------------------------------------------------------------------------------------------------------------------------------------------
// stage 0: seed + primary ray of every item (jitter = 0, pass1:80-82); chunks of 256 items are dealt round-robin to the workgroups
__global__ __launch_bounds__(kBlock) void k_rs_raygen(DevFrame f, RsQ q, const CameraGPU* __restrict__ cam_p, uint32_t sample_id, uint32_t* __restrict__ cnt_out) {
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_n;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const size_t qb = (size_t)rs_wg(q) * q.qcap;
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x = 0, y = 0;
        const bool valid = it < q.nitems && rs_item_pixel(f, q, it, x, y);
        const uint32_t slot = block_push(valid, &s_n);
        if (valid) {
            uint32_t s0, s1; seed_init(x, y, sample_id, f.frame_seed, s0, s1);
            f3 o, d; primary_ray(cam, f.width, f.height, x, y, 0.0f, 0.0f, o, d);
            q.st[0][0][qb + slot] = {o.x, o.y, o.z, u2f(s1)};
            q.st[0][1][qb + slot] = {d.x, d.y, d.z, u2f(s0)};
            q.st[0][3][qb + slot] = {0.0f, 0.0f, 0.0f, u2f(it)};
        }
    }
    rs_publish(cnt_out, &s_n, nullptr, nullptr, q, 0);
}

// (round 5) a short material table (<= 16 records) and light list (<= 32 records + CDF) in LDS for the stage kernels: every stage reads the pixel's material record behind its
// sample record, pass 1 also runs NEE (CDF search, light record) per candidate — dependent global reads at 4-5 waves per SIMD.  The kernel sees a COPY of the scene
// descriptor whose table pointers aim at LDS (everything is inlined, so the rest of the code is untouched); longer tables stay in global memory.
#ifndef RTX_RS_P1_WAVES
#define RTX_RS_P1_WAVES 5      // k_rs_p1_first / _loop: 96 VGPRs (5 waves / SIMD) before the LDS tables, 101-105 with them unless asked for 5
#endif
struct RsTables { F4 mats[16 * 10]; F4 lights[32 * 5]; float cdf[32]; };
__device__ __forceinline__ DevScene rs_stage_tables(const DevScene& in, RsTables& T) {
    DevScene sc = in;
    if (in.nmat && in.nmat <= 16u) { for (uint32_t i = threadIdx.x; i < in.nmat * 10u; i += kBlock) T.mats[i] = ((const F4*)in.mats)[i]; sc.mats = (const MatGPU*)T.mats; }
    if (in.nlights && in.nlights <= 32u) {
        for (uint32_t i = threadIdx.x; i < in.nlights * 5u; i += kBlock) T.lights[i] = ((const F4*)in.lights)[i];
        if (threadIdx.x < in.nlights) T.cdf[threadIdx.x] = in.cdf[threadIdx.x];
        sc.lights = (const LightGPU*)T.lights; sc.cdf = T.cdf;
    }
    return sc;
}
#ifndef RTX_NO_RS_LDS_TABLES
#define RS_STAGE_SCENE(sc_in) __shared__ RsTables rs_tab_; const DevScene sc = rs_stage_tables(sc_in, rs_tab_); __syncthreads()
#else
#define RS_STAGE_SCENE(sc_in) const DevScene& sc = sc_in
#endif

// stage 1: primary hit -> pixels that sample nothing are finished here; the others run SampleRIS up to its BSDF-candidate ray (set 0 -> set 1)
__global__ __launch_bounds__(kBlock, 4) void k_rs_p1_ris(DevScene sc_in, DevFrame f, RsQ q, const uint32_t* __restrict__ cnt_in, uint32_t* __restrict__ cnt_out,
                                                         F4* __restrict__ accum, uint32_t* __restrict__ res_di, uint32_t* __restrict__ res_gi, uint32_t* __restrict__ sdata) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n = cnt_in[rs_wg(q)];
    const size_t qb = (size_t)rs_wg(q) * q.qcap;
    const uint32_t M1 = sc.nlights ? f.nee_samples : 0u;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        uint32_t s0 = 0, s1 = 0, item = 0;
        Surf pay; pay.pos = mk3(0, 0, 0); pay.normal = mk3(0, 0, 1); pay.mat = 0; pay.inst = 0;
        f3 outgoing = mk3(0, 0, 1), smp = mk3(0, 0, 1);
        Res rdi = zero_res();
        if (i < n) {
            const F4 ro = q.st[0][0][qb + i], rd = q.st[0][1][qb + i], h = q.hit[qb + i];
            item = f2u(q.st[0][3][qb + i].w); s1 = f2u(ro.w); s0 = f2u(rd.w);
            const f3 o = mk3(ro.x, ro.y, ro.z), d = mk3(rd.x, rd.y, rd.z);
            bool hit = f2u(h.w) != kMissPrim;
            if (hit) { pay = surface(sc, o, d, h.x, h.y, h.z, f2u(h.w)); hit = pay.mat < sc.nmat; }
            const MatGPU& m = sc.mats[hit ? pay.mat : 0u];
            if (hit && !(m.KeFullLen > 0.0f)) {                                                  // performSampling, pass1:102-106
                outgoing = -d;
                smp = ris_front(sc, f.flags, M1, outgoing, rdi, pay.pos, pay.normal, pay.mat, s0, s1);
                alive = true;
            } else {                                                                             // miss, or a light seen directly: the pixel's records are final
                uint32_t x = 0, y = 0; (void)rs_item_pixel(f, q, item, x, y);
                const size_t slot = map_pixel_id(f.width, x, y);
                const f3 L1 = hit ? mk3(m.Ke[0], m.Ke[1], m.Ke[2]) : mk3(0, 0, 0);
                const Res z = zero_res();
                store_res(res_di + slot * 10, z); store_res(res_gi + slot * 10, z);
                uint32_t* dst = sdata + slot * 15;
                store_sdata_head(dst, mk3(0, 0, 0), hit ? pay.mat : kMissMat, L1, mk3(0, 0, 0), mk3(0, 0, 0), hit ? pay.inst : 0u);
                dst[12] = 0u; dst[13] = 0u; dst[14] = 0u;
                q.cls[item] = 0u;
                rs_accumulate(accum, f.width, x, y, L1);
            }
        }
        const uint32_t slot = block_push(alive, &s_n);
        if (alive) {
            const size_t p = qb + slot;
            q.st[1][0][p] = {pay.pos.x, pay.pos.y, pay.pos.z, u2f(s1)};
            q.st[1][1][p] = {smp.x, smp.y, smp.z, u2f(s0)};
            q.st[1][2][p] = {pay.normal.x, pay.normal.y, pay.normal.z, u2f(pay.mat)};
            q.st[1][3][p] = {outgoing.x, outgoing.y, outgoing.z, u2f(item)};
            q.st[1][4][p] = {rdi.x2.x, rdi.x2.y, rdi.x2.z, rdi.w_sum};
            q.st[1][5][p] = {rdi.n2.x, rdi.n2.y, rdi.n2.z, u2f(pay.inst)};
            q.st[1][6][p] = {rdi.L2.x, rdi.L2.y, rdi.L2.z, 0.0f};
        }
    }
    rs_publish(cnt_out, &s_n, nullptr, nullptr, q, 1);
}

// stage 2: the BSDF candidate's hit closes SampleRIS: the DI reservoir and the pixel's sample record are written, the DI visibility ray goes to the ray queue
// (it only decides whether W becomes 0, which k_rs_p1_finish applies: it is traced later, together with the rays of k_rs_p1_emit_final), and the path sampler's first
// BSDF ray starts (set 1 -> set 0)
__global__ __launch_bounds__(kBlock, 4) void k_rs_p1_ris_finish(DevScene sc_in, DevFrame f, RsQ q, const uint32_t* __restrict__ cnt_in, uint32_t* __restrict__ cnt_out, uint32_t* __restrict__ shcnt,
                                                                uint32_t* __restrict__ res_di, uint32_t* __restrict__ sdata) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_n, s_rn;
    if (threadIdx.x == 0) { s_n = 0; s_rn = 0; }
    __syncthreads();
    const uint32_t n = cnt_in[rs_wg(q)];
    const size_t qb = (size_t)rs_wg(q) * q.qcap;
    const uint32_t M1 = sc.nlights ? f.nee_samples : 0u;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const bool act = i < n;
        uint32_t s0 = 0, s1 = 0, item = 0, mat = 0;
        f3 origin = mk3(0, 0, 0), normal = mk3(0, 0, 1), outn = mk3(0, 0, 1), smp2 = mk3(0, 0, 1);
        F4 so = {0, 0, 0, 0}, sd = {0, 0, 1, 0};
        if (act) {
            const size_t p = qb + i;
            const F4 ro = q.st[1][0][p], rd = q.st[1][1][p], a0 = q.st[1][2][p], a1 = q.st[1][3][p], r0 = q.st[1][4][p], r1 = q.st[1][5][p], r2 = q.st[1][6][p], h = q.hit[p];
            origin = mk3(ro.x, ro.y, ro.z); s1 = f2u(ro.w); const f3 smp = mk3(rd.x, rd.y, rd.z); s0 = f2u(rd.w);
            normal = mk3(a0.x, a0.y, a0.z); mat = f2u(a0.w); const f3 outgoing = mk3(a1.x, a1.y, a1.z); item = f2u(a1.w);
            Res rdi; rdi.x2 = mk3(r0.x, r0.y, r0.z); rdi.w_sum = r0.w; rdi.n2 = mk3(r1.x, r1.y, r1.z); rdi.W = 0.0f; rdi.L2 = mk3(r2.x, r2.y, r2.z); rdi.M = 0;
            const uint32_t inst = f2u(r1.w);
            Surf h2; h2.pos = mk3(0, 0, 0); h2.normal = mk3(0, 0, 1); h2.mat = 0; h2.inst = 0;
            bool hit = f2u(h.w) != kMissPrim;
            if (hit) { h2 = surface(sc, origin, smp, h.x, h.y, h.z, f2u(h.w)); hit = h2.mat < sc.nmat; }
            ris_back(sc, f.flags, M1, outgoing, rdi, origin, normal, mat, smp, hit, h2, s0, s1);
            const MatGPU& m = sc.mats[mat];
            const f3 x1 = origin, n1 = normalize(normal), ov = outgoing;
            const float f_g = length(reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov));
            rdi.W = f_g > kEps ? rdi.w_sum / f_g : 0.0f;                           // p_hat = f_g * 1 (visible); an occluded ray makes it 0 -> W = 0 (k_rs_p1_finish)
            vis_ray(x1, n1, rdi.x2, so, sd);
            uint32_t x = 0, y = 0; (void)rs_item_pixel(f, q, item, x, y);
            const size_t slot = map_pixel_id(f.width, x, y);
            store_res(res_di + slot * 10, rdi);
            store_sdata_head(sdata + slot * 15, x1, mat, mk3(m.Ke[0], m.Ke[1], m.Ke[2]), n1, ov, inst);
            q.cls[item] = mat + 1u;
            const F4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            q.cold[(size_t)2 * q.nitems + item] = z; q.cold[(size_t)3 * q.nitems + item] = z; q.cold[(size_t)4 * q.nitems + item] = z;
            GiHot H; gi_begin(H, origin, normal, outgoing, mat);                   // SamplePathSimple starts: first BSDF sample (Path_Sampler_v6.hlsl:37-52)
            outn = H.outgoing;
            smp2 = gi_first_sample(sc, f.flags, H, s0, s1);
        }
        rs_push_ray(q, &s_rn, act, so, sd, item * kRsOcc + 0u);
        const uint32_t slot = block_push(act, &s_n);
        if (act) {
            const size_t p = qb + slot;
            q.st[0][0][p] = {origin.x, origin.y, origin.z, u2f(s1)};
            q.st[0][1][p] = {smp2.x, smp2.y, smp2.z, u2f(s0)};
            q.st[0][2][p] = {normal.x, normal.y, normal.z, u2f(mat)};
            q.st[0][3][p] = {outn.x, outn.y, outn.z, u2f(item)};
        }
    }
    rs_publish(cnt_out, &s_n, shcnt, &s_rn, q, 1);
}

// what a GI reservoir update selects goes to the pixel's cold record (rare: a store, no registers held across the stage)
struct ColdSel {
    const RsQ& q; uint32_t item;
    __device__ __forceinline__ void operator()(f3 L2h, bool light, f3 x1s, f3 x2s) const {
        q.cold[(size_t)4 * q.nitems + item] = {L2h.x, L2h.y, L2h.z, 1.0f};
        if (light) { q.cold[(size_t)2 * q.nitems + item] = {x1s.x, x1s.y, x1s.z, 0.0f}; q.cold[(size_t)3 * q.nitems + item] = {x2s.x, x2s.y, x2s.z, 0.0f}; }
    }
};
__device__ __forceinline__ void rs_store_hot(const RsQ& q, uint32_t set, size_t p, const GiHot& H, f3 smp, uint32_t s0, uint32_t s1, uint32_t item) {
    q.st[set][0][p] = {H.origin.x, H.origin.y, H.origin.z, u2f(s1)};
    q.st[set][1][p] = {smp.x, smp.y, smp.z, u2f(s0)};
    q.st[set][2][p] = {H.normal.x, H.normal.y, H.normal.z, u2f(H.mat)};
    q.st[set][3][p] = {H.outgoing.x, H.outgoing.y, H.outgoing.z, u2f(item)};
    q.st[set][4][p] = {H.acc_f.x, H.acc_f.y, H.acc_f.z, H.acc_pdf};
    q.st[set][5][p] = {H.acc_f_rec.x, H.acc_f_rec.y, H.acc_f_rec.z, 0.0f};
    q.st[set][6][p] = {H.acc_L.x, H.acc_L.y, H.acc_L.z, H.w_sum};
}

// stage 3: first path vertex (set 0 -> set 1).  A path that ends here contributes nothing; the others record (xn, nn) and run the loop body up to its ray
__global__ __launch_bounds__(kBlock, RTX_RS_P1_WAVES) void k_rs_p1_first(DevScene sc_in, DevFrame f, RsQ q, const uint32_t* __restrict__ cnt_in, uint32_t* __restrict__ cnt_out) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n = cnt_in[rs_wg(q)];
    const size_t qb = (size_t)rs_wg(q) * q.qcap;
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        uint32_t s0 = 0, s1 = 0, item = 0;
        GiHot H; gi_begin(H, mk3(0, 0, 0), mk3(0, 0, 1), mk3(0, 0, 1), 0u);
        f3 smp2 = mk3(0, 0, 1);
        if (i < n) {
            const size_t p = qb + i;
            const F4 ro = q.st[0][0][p], rd = q.st[0][1][p], a0 = q.st[0][2][p], a1 = q.st[0][3][p], h = q.hit[p];
            H.origin = mk3(ro.x, ro.y, ro.z); s1 = f2u(ro.w); const f3 smp = mk3(rd.x, rd.y, rd.z); s0 = f2u(rd.w);
            H.normal = mk3(a0.x, a0.y, a0.z); H.mat = f2u(a0.w); H.outgoing = mk3(a1.x, a1.y, a1.z); item = f2u(a1.w);
            Surf hs; hs.pos = mk3(0, 0, 0); hs.normal = mk3(0, 0, 1); hs.mat = 0; hs.inst = 0;
            bool hit = f2u(h.w) != kMissPrim;
            if (hit) { hs = surface(sc, H.origin, smp, h.x, h.y, h.z, f2u(h.w)); hit = hs.mat < sc.nmat; }
            alive = gi_first_hit(sc, f.flags, H, smp, hit, hs);
            if (alive) {
                const f3 xn = H.origin, nn = normalize(H.normal);
                q.cold[item] = {xn.x, xn.y, xn.z, 0.0f}; q.cold[(size_t)q.nitems + item] = {nn.x, nn.y, nn.z, 0.0f};
                if (f.max_bounces > 0u) smp2 = gi_front(sc, f.flags, nee, H, s0, s1, ColdSel{q, item});
                else alive = false;
            }
            if (!alive) q.fin[item] = {H.acc_L.x, H.acc_L.y, H.acc_L.z, H.w_sum};            // (zero, unless `bounces` = 0 let the path end after its first vertex)
        }
        const uint32_t slot = block_push(alive, &s_n);
        if (alive) rs_store_hot(q, 1u, qb + slot, H, smp2, s0, s1, item);
    }
    rs_publish(cnt_out, &s_n, nullptr, nullptr, q, 1);
}

// stage 4, once per loop iteration `iter`: the hit closes iteration `iter`; a surviving path runs iteration iter + 1 up to its ray (set `set` -> the other)
__global__ __launch_bounds__(kBlock, RTX_RS_P1_WAVES) void k_rs_p1_loop(DevScene sc_in, DevFrame f, RsQ q, uint32_t set, uint32_t iter, const uint32_t* __restrict__ cnt_in, uint32_t* __restrict__ cnt_out) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n = cnt_in[rs_wg(q)];
    const size_t qb = (size_t)rs_wg(q) * q.qcap;
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        uint32_t s0 = 0, s1 = 0, item = 0;
        GiHot H; gi_begin(H, mk3(0, 0, 0), mk3(0, 0, 1), mk3(0, 0, 1), 0u);
        f3 smp2 = mk3(0, 0, 1);
        if (i < n) {
            const size_t p = qb + i;
            const F4 ro = q.st[set][0][p], rd = q.st[set][1][p], a0 = q.st[set][2][p], a1 = q.st[set][3][p], a2 = q.st[set][4][p], a3 = q.st[set][5][p], a4 = q.st[set][6][p], h = q.hit[p];
            H.origin = mk3(ro.x, ro.y, ro.z); s1 = f2u(ro.w); const f3 smp = mk3(rd.x, rd.y, rd.z); s0 = f2u(rd.w);
            H.normal = mk3(a0.x, a0.y, a0.z); H.mat = f2u(a0.w); H.outgoing = mk3(a1.x, a1.y, a1.z); item = f2u(a1.w);
            H.acc_f = mk3(a2.x, a2.y, a2.z); H.acc_pdf = a2.w; H.acc_f_rec = mk3(a3.x, a3.y, a3.z); H.acc_L = mk3(a4.x, a4.y, a4.z); H.w_sum = a4.w;
            Surf hs; hs.pos = mk3(0, 0, 0); hs.normal = mk3(0, 0, 1); hs.mat = 0; hs.inst = 0;
            bool hit = f2u(h.w) != kMissPrim;
            if (hit) { hs = surface(sc, H.origin, smp, h.x, h.y, h.z, f2u(h.w)); hit = hs.mat < sc.nmat; }
            alive = gi_back(sc, f.flags, nee, H, smp, hit, hs, s0, s1, ColdSel{q, item});
            if (alive && iter + 1u < f.max_bounces) smp2 = gi_front(sc, f.flags, nee, H, s0, s1, ColdSel{q, item});
            else alive = false;
            if (!alive) q.fin[item] = {H.acc_L.x, H.acc_L.y, H.acc_L.z, H.w_sum};
        }
        const uint32_t slot = block_push(alive, &s_n);
        if (alive) rs_store_hot(q, set ^ 1u, qb + slot, H, smp2, s0, s1, item);
    }
    rs_publish(cnt_out, &s_n, nullptr, nullptr, q, 1);
}

// stage 5: the shadow ray of the selected reconnection (Path_Sampler_v6.hlsl:271-283), for every pixel that sampled — appended to the DI visibility rays of stage 2
// (the ray sub-queues are workgroup-private and a pixel stays with its workgroup, so the counter simply continues).  In a ReSTIR frame (with_p2) the two visibility
// rays of the TEMPORAL pass join them: they depend on this frame's primary hit (written by stages 1 / 2) and on last frame's records only, so all four rays of a
// pixel are traversed by ONE persistent launch.  Occlusion bytes of a pixel: 0 DI, 1 GI reconnection, 2 / 3 temporal DI / GI.
__global__ __launch_bounds__(kBlock, 4) void k_rs_p1_emit_final(DevScene sc_in, DevFrame f, RsQ q, const CameraGPU* __restrict__ cam_p, RestirBufs B, uint32_t with_p2, uint32_t* __restrict__ shcnt) {
    RS_STAGE_SCENE(sc_in);
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_rn;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    const uint32_t ray_base = shcnt[rs_wg(q)];
    if (threadIdx.x == 0) s_rn = ray_base;
    __syncthreads();
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        bool cast = false, r0 = false, r1 = false;
        F4 so = {0, 0, 0, 0}, sd = {0, 0, 1, 0}, so0 = so, sd0 = sd, so1 = so, sd1 = sd;
        uint32_t x, y;
        if (it < q.nitems && rs_item_pixel(f, q, it, x, y)) {
            if (q.cls[it]) {
                const F4 a = q.cold[(size_t)2 * q.nitems + it], b = q.cold[(size_t)3 * q.nitems + it];
                cast = gi_final_ray(nee, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), so, sd);
            }
            if (with_p2) {
                P2Pix I;
                if (p2_gather(sc, f, cam, B, x, y, I, false)) {
                    r0 = I.acc_di; r1 = I.acc_gi;
                    if (r0) vis_ray(I.sd.x1, I.sd.n1, I.rl.x2, so0, sd0);
                    if (r1) vis_ray(I.sd.x1, I.sd.n1, I.gl.x2, so1, sd1);
                }
            }
        }
        rs_push_ray(q, &s_rn, cast, so, sd, it * kRsOcc + 1u);
        if (with_p2) {                                                  // (uniform)
            rs_push_ray(q, &s_rn, r0, so0, sd0, it * kRsOcc + 2u);
            rs_push_ray(q, &s_rn, r1, so1, sd1, it * kRsOcc + 3u);
        }
    }
    rs_publish(nullptr, nullptr, shcnt, &s_rn, q, 1, ray_base);
}

// stage 6: the visibility answers are in: W of the DI reservoir, w_sum / W of the GI reservoir, the pixel's estimate (pass1:140-190) — and, in a ReSTIR frame
// (with_p2), the pixel's TEMPORAL pass right behind it (RayGen_v6_pass2.hlsl:46-204: it reads this pixel's own pass-1 records and last frame's buffers only)
// (round 4) P1 / P2: which halves an instantiation runs.  As ONE kernel (<true, true>, round 3) the two halves' live ranges added up to 128 VGPRs + 55 spilled (148 B of scratch
// per lane) at 0.96 waves per SIMD resident and 75 % of the wave time parked (profiles/r03_pmc_restir.md); as two launches, <true, false> then <false, true>, each half fits
// its registers (profiles/r04_kernel_resources.md).  The temporal pass of a pixel reads that pixel's own pass-1 records: the kernel boundary orders them.
// (round 5) the temporal half alone asks for 3 waves per SIMD: 128 VGPRs + 23 spilled (96 B of scratch per lane) at 4, 0 spilled at 3 — and the launch never had more than one
// resident wave per SIMD anyway (profiles/r04_pmc_restir.md: 0.64)
template <bool P1, bool P2>
__global__ __launch_bounds__(kBlock, (P2 && !P1) ? 3 : 4) void k_rs_p1_finish(DevScene sc_in, DevFrame f, RsQ q, F4* __restrict__ accum, uint32_t* __restrict__ res_di, uint32_t* __restrict__ res_gi,
                                                            uint32_t* __restrict__ sdata, const CameraGPU* __restrict__ cam_p, RestirBufs B, uint32_t with_p2) {
    RS_STAGE_SCENE(sc_in);
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    __syncthreads();
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x, y;
        if (!(it < q.nitems) || !rs_item_pixel(f, q, it, x, y)) continue;
        const uint32_t cl = P1 ? q.cls[it] : 0u;
        const uint8_t* oc = q.occ + (size_t)it * kRsOcc;
        if (P1 && cl) {
            const MatGPU& m = sc.mats[cl - 1u];
            const size_t slot = map_pixel_id(f.width, x, y);
            const uint32_t* sp = sdata + slot * 15;
            const f3 x1 = mk3(u2f(sp[0]), u2f(sp[1]), u2f(sp[2])), n1 = mk3(u2f(sp[5]), u2f(sp[6]), u2f(sp[7])), ov = mk3(u2f(sp[8]), u2f(sp[9]), u2f(sp[10]));
            Res rdi = load_res_dev(res_di + slot * 10);
            if (oc[0]) { rdi.W = 0.0f; res_di[slot * 10 + 7] = 0u; }
            const F4 fv = q.fin[it], c0 = q.cold[it], c1 = q.cold[(size_t)q.nitems + it], c2 = q.cold[(size_t)2 * q.nitems + it], c3 = q.cold[(size_t)3 * q.nitems + it], c4 = q.cold[(size_t)4 * q.nitems + it];
            Res rgi = zero_res();
            rgi.w_sum = fv.w; rgi.L2 = mk3(c4.x, c4.y, c4.z);
            if (c4.w != 0.0f) { rgi.x2 = mk3(c0.x, c0.y, c0.z); rgi.n2 = normalize(mk3(c1.x, c1.y, c1.z)); }
            F4 so, sd;
            if (gi_final_ray(nee, mk3(c2.x, c2.y, c2.z), mk3(c3.x, c3.y, c3.z), so, sd)) {
                if (oc[1]) rgi.w_sum *= 0.0f;
                else rgi.w_sum *= 1.0f;
            }
            f3 debug = mk3(fv.x, fv.y, fv.z);
            const f3 rc = reconnect_di_dev(m, f.flags, x1, n1, rdi.x2, rdi.n2, rdi.L2, ov);
            debug = debug + rc * rdi.W;
            gi_finish(m, f.flags, x1, n1, ov, rgi);
            store_res(res_gi + slot * 10, rgi);
            uint32_t* d = sdata + slot * 15;
            d[12] = f2u(debug.x); d[13] = f2u(debug.y); d[14] = f2u(debug.z);
            rs_accumulate(accum, f.width, x, y, debug);
        }
        if (P2 && with_p2) {                                            // every pixel, as the literal pass does (a pixel that sampled nothing leaves at p2_gather's first test)
            P2Pix I;
            if (p2_gather(sc, f, cam, B, x, y, I, true)) p2_merge(sc, f, B, x, y, I, VisLookup{oc + 2});
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// PASS 3 (RayGen_v6_pass3.hlsl:46-441): select (+ emit rays 0-8) | trace | merge DI (+ emit ray 9) | merge GI | trace | shade
// ---------------------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool p3_samples(const DevScene& sc, const SData& sd) {      // the pixel runs the spatial pass (not a light seen directly, not a miss)
    return (sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f) && !(sd.mID == 0xFFFEu || sd.mID >= sc.nmat);
}
__global__ __launch_bounds__(kBlock, 4) void k_rs_p3_select(DevScene sc_in, DevFrame f, RsQ q, const CameraGPU* __restrict__ cam_p, RestirBufs B, uint32_t* __restrict__ shcnt) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_rn;
    if (threadIdx.x == 0) s_rn = 0;
    __syncthreads();
    const f3 camo = mk3(cam_p->viewI[12], cam_p->viewI[13], cam_p->viewI[14]);
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x = 0, y = 0;
        bool run = it < q.nitems && rs_item_pixel(f, q, it, x, y);
        SData sd = zero_sd(); Res rcur = zero_res(), gcur = zero_res();
        P3Cand K; K.n_di = 0; K.n_gi = 0; K.M_sum_DI = 0.0f; K.M_sum_GI = 0.0f;
        for (int k = 0; k < 3; k++) { K.di[k] = 0xFFFFFFFFu; K.gi[k] = 0xFFFFFFFFu; }
        if (run) {
            const size_t slot = map_pixel_id(f.width, x, y);
            sd = load_sd_dev(B.cur_sd + slot * 15);
            run = p3_samples(sc, sd);
            if (run) {
                uint32_t s0, s1; seed_init(x, y, 3u, f.frame_seed, s0, s1);
                rcur = load_res_dev(B.cur_di + slot * 10); gcur = load_res_dev(B.cur_gi + slot * 10);
                p3_select(sc, f, B, camo, x, y, sd, sc.mats[sd.mID], rcur, gcur, s0, s1, K);
                uint32_t* rec = q.cand + (size_t)it * kRsCand;
                for (int k = 0; k < 3; k++) { rec[k] = k < K.n_di ? K.di[k] : 0xFFFFFFFFu; rec[3 + k] = k < K.n_gi ? K.gi[k] : 0xFFFFFFFFu; }
                rec[6] = s0; rec[7] = s1; rec[8] = f2u(K.M_sum_DI); rec[9] = f2u(K.M_sum_GI);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {                                   // every lane takes part in every push (convergent compaction)
            const int j = k % 3;
            const bool on = run && (k < 3 ? j < K.n_di : j < K.n_gi);
            F4 so = {0, 0, 0, 0}, sdv = {0, 0, 1, 0};
            if (on) {
                if (k < 3) { const SData sn = load_sd_dev(B.cur_sd + (size_t)K.di[j] * 15); vis_ray(sn.x1, sn.n1, rcur.x2, so, sdv); }
                else if (k < 6) { const SData sn = load_sd_dev(B.cur_sd + (size_t)K.gi[j] * 15); vis_ray(sn.x1, sn.n1, gcur.x2, so, sdv); }
                else { const Res gn = load_res_dev(B.cur_gi + (size_t)K.gi[j] * 10); vis_ray(sd.x1, sd.n1, gn.x2, so, sdv); }
            }
            rs_push_ray(q, &s_rn, on, so, sdv, it * kRsOcc + (uint32_t)k);
        }
    }
    rs_publish(nullptr, nullptr, shcnt, &s_rn, q, 1);
}
// ---- (round 4) compact neighbour records of the spatial pass -------------------------------------------------------------------------------------------------
// p3_select tests up to 9 + 9 random neighbours per pixel; a test reads 13 of the 25 dwords of the neighbour's sample record (60 B) and DI / GI reservoir (40 B) — three to five
// 64-B sectors of scattered traffic per attempt (2.2 x what it needs: profiles/r03_pmc_restir.md).  After passes 1 + 2 every pixel of the list writes what a neighbour's
// test needs into two 32-B records (one sector each): A = x1, n1, material id | "L1 is zero" | DI reservoir valid | DI M; B = the GI reservoir's x2, n2, w_sum, M.  The
// flags are evaluated with the literal expressions (length(L1) == 0, valid_res_dev), the floats are copies: the keyed selection returns the same candidates, bit for bit.
struct RsKeys { F4* a; F4* b; };                   // [2 * slot], [2 * slot + 1]; slot = map_pixel_id
__global__ __launch_bounds__(kBlock) void k_rs_p3_keys(DevFrame f, RsQ q, RestirBufs B, RsKeys Ky) {
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x, y;
        if (!(it < q.nitems) || !rs_item_pixel(f, q, it, x, y)) continue;
        const size_t slot = map_pixel_id(f.width, x, y);
        const SData sn = load_sd_dev(B.cur_sd + slot * 15); const Res rn = load_res_dev(B.cur_di + slot * 10), gn = load_res_dev(B.cur_gi + slot * 10);
        const uint32_t meta = sn.mID | (length(sn.L1) == 0.0f ? 1u << 16 : 0u);
        const uint32_t di = (rn.M & 0x7FFFFFFFu) | (valid_res_dev(rn) ? 1u << 31 : 0u);
        Ky.a[2 * slot] = {sn.x1.x, sn.x1.y, sn.x1.z, sn.n1.x}; Ky.a[2 * slot + 1] = {sn.n1.y, sn.n1.z, u2f(meta), u2f(di)};
        Ky.b[2 * slot] = {gn.x2.x, gn.x2.y, gn.x2.z, gn.n2.x}; Ky.b[2 * slot + 1] = {gn.n2.y, gn.n2.z, gn.w_sum, u2f(gn.M)};
    }
}
// p3_select (rtx_restir.hpp) on the records: same draws, same tests in the same arithmetic, same candidates
__device__ __forceinline__ void p3_select_keys(const DevFrame& f, const RsKeys& Ky, f3 camo, uint32_t x, uint32_t y, const SData& sd, const MatGPU& m,
                                               const Res& rcur, const Res& gcur, uint32_t& s0, uint32_t& s1, P3Cand& K) {
    const uint32_t W = f.width, H = f.height;
    K.n_di = 0; K.n_gi = 0;
    K.M_sum_DI = minf_u(128.0f, rcur.M); K.M_sum_GI = minf_u(128.0f, gcur.M);
    for (int a = 0; a < 9 && K.n_di < 3; a++) {
        int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
        const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        const F4 a0 = Ky.a[2 * pr], a1 = Ky.a[2 * pr + 1];
        const f3 nx1 = mk3(a0.x, a0.y, a0.z), nn1 = mk3(a0.w, a1.x, a1.y);
        const uint32_t meta = f2u(a1.z), di = f2u(a1.w);
        const bool ok = !(dot(sd.n1, nn1) < 0.9f) && !reject_distance_dev(sd.x1, nx1, camo, 0.1f) && (di >> 31) != 0u && (meta & 0x10000u) != 0u && (meta & 0xFFFFu) == sd.mID;
        if (ok) { K.di[K.n_di++] = (uint32_t)pr; K.M_sum_DI += minf_u(128.0f, di & 0x7FFFFFFFu); }
    }
    for (int a = 0; a < 9 && K.n_gi < 3; a++) {
        int nx, ny; random_pixel_dev(20u, W, H, x, y, s0, s1, nx, ny);
        const size_t pr = map_pixel_id(W, (uint32_t)nx, (uint32_t)ny);
        const F4 a0 = Ky.a[2 * pr], a1 = Ky.a[2 * pr + 1], b0 = Ky.b[2 * pr], b1 = Ky.b[2 * pr + 1];
        SData sn = zero_sd(); sn.x1 = mk3(a0.x, a0.y, a0.z);                   // (jacobian_dev reads x1 only)
        const uint32_t meta = f2u(a1.z);
        Res gn = zero_res(); gn.x2 = mk3(b0.x, b0.y, b0.z); gn.n2 = mk3(b0.w, b1.x, b1.y); gn.w_sum = b1.z; gn.M = f2u(b1.w);
        const bool ok = m.Pr > 0.3f && !reject_distance_dev(sd.x1, sn.x1, camo, 0.1f) && !(dot(normalize(gn.x2 - sd.x1), sd.n1) < 0.0f) &&
                        !(gn.w_sum > 5.0f) && valid_res_gi_dev(gn) && !reject_jacobian_dev(jacobian_dev(sn, sd, gn.x2, gn.n2), 5.0f) &&
                        (meta & 0x10000u) != 0u && (meta & 0xFFFFu) == sd.mID;
        if (ok) { K.gi[K.n_gi++] = (uint32_t)pr; K.M_sum_GI += minf_u(128.0f, gn.M); }
    }
}
__global__ __launch_bounds__(kBlock, 4) void k_rs_p3_select_keys(DevScene sc_in, DevFrame f, RsQ q, const CameraGPU* __restrict__ cam_p, RestirBufs B, RsKeys Ky, uint32_t* __restrict__ shcnt) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_rn;
    if (threadIdx.x == 0) s_rn = 0;
    __syncthreads();
    const f3 camo = mk3(cam_p->viewI[12], cam_p->viewI[13], cam_p->viewI[14]);
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x = 0, y = 0;
        bool run = it < q.nitems && rs_item_pixel(f, q, it, x, y);
        SData sd = zero_sd(); Res rcur = zero_res(), gcur = zero_res();
        P3Cand K; K.n_di = 0; K.n_gi = 0; K.M_sum_DI = 0.0f; K.M_sum_GI = 0.0f;
        for (int k = 0; k < 3; k++) { K.di[k] = 0xFFFFFFFFu; K.gi[k] = 0xFFFFFFFFu; }
        if (run) {
            const size_t slot = map_pixel_id(f.width, x, y);
            sd = load_sd_dev(B.cur_sd + slot * 15);
            run = p3_samples(sc, sd);
            if (run) {
                uint32_t s0, s1; seed_init(x, y, 3u, f.frame_seed, s0, s1);
                rcur = load_res_dev(B.cur_di + slot * 10); gcur = load_res_dev(B.cur_gi + slot * 10);
                p3_select_keys(f, Ky, camo, x, y, sd, sc.mats[sd.mID], rcur, gcur, s0, s1, K);
                uint32_t* rec = q.cand + (size_t)it * kRsCand;
                for (int k = 0; k < 3; k++) { rec[k] = k < K.n_di ? K.di[k] : 0xFFFFFFFFu; rec[3 + k] = k < K.n_gi ? K.gi[k] : 0xFFFFFFFFu; }
                rec[6] = s0; rec[7] = s1; rec[8] = f2u(K.M_sum_DI); rec[9] = f2u(K.M_sum_GI);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {                                   // every lane takes part in every push (convergent compaction)
            const int j = k % 3;
            const bool on = run && (k < 3 ? j < K.n_di : j < K.n_gi);
            F4 so = {0, 0, 0, 0}, sdv = {0, 0, 1, 0};
            if (on) {
                if (k < 6) {
                    const size_t pr = k < 3 ? K.di[j] : K.gi[j];
                    const F4 a0 = Ky.a[2 * pr], a1 = Ky.a[2 * pr + 1];
                    vis_ray(mk3(a0.x, a0.y, a0.z), mk3(a0.w, a1.x, a1.y), k < 3 ? rcur.x2 : gcur.x2, so, sdv);
                } else { const F4 b0 = Ky.b[2 * (size_t)K.gi[j]]; vis_ray(sd.x1, sd.n1, mk3(b0.x, b0.y, b0.z), so, sdv); }
            }
            rs_push_ray(q, &s_rn, on, so, sdv, it * kRsOcc + (uint32_t)k);
        }
    }
    rs_publish(nullptr, nullptr, shcnt, &s_rn, q, 1);
}
// GI = false: the DI merge (+ the ray of the selected DI sample); GI = true: the GI merge, whose random numbers continue behind the n_di the DI merge drew.  Two launches
// of half the register pressure each: as one kernel the merge needed 128 VGPRs with 51 of them spilled (148 B of scratch per lane).
template <bool GI>
__global__ __launch_bounds__(kBlock, 4) void k_rs_p3_merge(DevScene sc_in, DevFrame f, RsQ q, RestirBufs B, uint32_t* __restrict__ shcnt) {
    RS_STAGE_SCENE(sc_in);
    __shared__ uint32_t s_rn;
    if (threadIdx.x == 0) s_rn = 0;
    __syncthreads();
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x = 0, y = 0;
        bool run = it < q.nitems && rs_item_pixel(f, q, it, x, y);
        F4 so = {0, 0, 0, 0}, sdv = {0, 0, 1, 0};
        if (run) {
            const size_t slot = map_pixel_id(f.width, x, y);
            const SData sd = load_sd_dev(B.cur_sd + slot * 15);
            run = p3_samples(sc, sd);
            if (run) {
                const uint32_t* rec = q.cand + (size_t)it * kRsCand;
                P3Cand K; K.n_di = 0; K.n_gi = 0;
                for (int k = 0; k < 3; k++) { K.di[k] = rec[k]; K.gi[k] = rec[3 + k]; if (rec[k] != 0xFFFFFFFFu) K.n_di = k + 1; if (rec[3 + k] != 0xFFFFFFFFu) K.n_gi = k + 1; }
                uint32_t s0 = rec[6], s1 = rec[7]; K.M_sum_DI = u2f(rec[8]); K.M_sum_GI = u2f(rec[9]);
                if (!GI) {
                    Res rcur = load_res_dev(B.cur_di + slot * 10);
                    p3_merge_di(sc, f, B, sd, sc.mats[sd.mID], K, rcur, s0, s1, VisLookup{q.occ + (size_t)it * kRsOcc});
                    store_res(B.last_di + slot * 10, rcur);                                                  // W still to come (k_rs_p3_shade)
                    vis_ray(sd.x1, sd.n1, rcur.x2, so, sdv);
                } else {
                    for (int k = 0; k < K.n_di; k++) (void)tea_next(s0, s1);                                 // the DI merge's draws
                    Res gcur = load_res_dev(B.cur_gi + slot * 10);
                    p3_merge_gi(sc, f, B, sd, sc.mats[sd.mID], K, gcur, s0, s1, VisLookup{q.occ + (size_t)it * kRsOcc});
                    store_res(B.last_gi + slot * 10, gcur);
                }
            }
        }
        if (!GI) rs_push_ray(q, &s_rn, run, so, sdv, it * kRsOcc + 9u);
    }
    if (!GI) rs_publish(nullptr, nullptr, shcnt, &s_rn, q, 1);
}
__global__ __launch_bounds__(kBlock, 4) void k_rs_p3_shade(DevScene sc_in, DevFrame f, RsQ q, RestirBufs B, F4* __restrict__ accum) {
    RS_STAGE_SCENE(sc_in);
    const uint32_t nchunks = (q.nitems + kBlock - 1) / kBlock;
    for (uint32_t c = rs_wg(q); c < nchunks; c += gridDim.x) {
        const uint32_t it = c * kBlock + threadIdx.x;
        uint32_t x, y;
        if (!(it < q.nitems) || !rs_item_pixel(f, q, it, x, y)) continue;
        const size_t slot = map_pixel_id(f.width, x, y);
        const SData sd = load_sd_dev(B.cur_sd + slot * 15);
        f3 out = mk3(0, 0, 0);
        if (!(sd.L1.x == 0.0f && sd.L1.y == 0.0f && sd.L1.z == 0.0f)) out = sd.L1;                      // pass3:457-462
        else if (p3_samples(sc, sd)) {
            Res rcur = load_res_dev(B.last_di + slot * 10), gcur = load_res_dev(B.last_gi + slot * 10);
            out = p3_shade(f, sd, sc.mats[sd.mID], rcur, gcur, VisLookup{q.occ + (size_t)it * kRsOcc});
            B.last_di[slot * 10 + 7] = f2u(rcur.W); B.last_gi[slot * 10 + 7] = f2u(gcur.W);
            for (int k = 0; k < 15; k++) B.last_sd[slot * 15 + k] = B.cur_sd[slot * 15 + k];
        }
        rs_accumulate(accum, f.width, x, y, out);
    }
}

}  // namespace rtx
