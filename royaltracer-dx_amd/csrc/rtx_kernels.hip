// rtx_kernels.hip — the wavefront path-tracing kernels for gfx950 (CDNA4, wave64).
//
// One sample batch is a set of paths with fixed slots ("pid"); per-path state lives in SoA float4 arrays in HBM.
// General scenes: each bounce runs  k_trace_closest -> k_shade -> k_trace_shadow[j]  over workgroup-private index
// sub-queues that k_shade re-compacts with a wave ballot + prefix sum (one LDS atomic per wave, no global atomics);
// the traversal kernels are persistent waves with dynamic ray fetch, the top of the BVH and the per-lane stack live
// in LDS.  Tiny scenes (<= 64 triangles, the Cornell Box): no BVH, a packed-FP32 plane/edge pre-test with
// scalar-loaded coefficients, and ONE fused kernel per bounce (k_bounce_small).  The reference's own passes
// (k_v6_pass1, k_restir_pass2/3) are thread-per-pixel kernels.  MFMA is unused on purpose: nothing here is a dense
// contraction.
//
// Parity-critical arithmetic (ray/triangle test, surface reconstruction, BSDF, light sampling, path
// throughput) follows rtx_math.hpp / rtx_bsdf.hpp with the library-wide -ffp-contract=off.  Ray/box
// tests are NOT parity-critical (closest hit is defined as the minimum over all triangles with a
// lowest-id tie break, any-hit as existence), they only have to be conservative.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "rtx_kernels.hpp"
#include "rtx_dev_common.hpp"   // wave helpers, compaction, slot -> pixel, primary ray
#include "rtx_traverse.hpp"     // triangle test, 8-wide BVH traversal (simple + persistent), tiny-scene pre-test
#include "rtx_shade.hpp"        // surface reconstruction, NEE, BSDF continuation
#include "rtx_restir.hpp"       // k_v6_pass1, k_restir_pass2 / 3 (thread per pixel) + the shared per-pixel math
#include "rtx_restir_wave.hpp"  // the same three passes as wavefront stages

namespace rtx {

#ifdef RTX_PROFILE_SECTIONS
__device__ unsigned long long g_sec[36];          // [0,12) cycles, [12,24) active lanes summed, [24,36) calls
#define PF_BEGIN Prof pfv; pfv.begin(); Prof* pf = &pfv
#define PF_MARK(i) pf->mark(i)
#define PF_COUNT(i) pf->count(i)
#define PF_FLUSH do { if (lane_id() == 0) for (int i = 0; i < 12; i++) { atomicAdd(&g_sec[i], pfv.acc[i]); atomicAdd(&g_sec[12 + i], pfv.lanes[i]); atomicAdd(&g_sec[24 + i], pfv.calls[i]); } } while (0)
#else
#define PF_BEGIN Prof* pf = nullptr; (void)pf
#define PF_MARK(i) do { } while (0)
#define PF_COUNT(i) do { } while (0)
#define PF_FLUSH do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// raygen: one thread per path slot of the batch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_raygen(DevFrame f, DevPaths p, const CameraGPU* __restrict__ cam_p, uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount, uint32_t compact) {
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_n;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t* myq = queue + (size_t)blockIdx.x * f.qcap;
    // chunk c = 256 consecutive path slots of ONE sample; chunks are dealt round-robin to workgroups so that
    // every workgroup's sub-queue holds a representative sample of the image (load balance across bounces)
    const uint32_t nchunks = f.chunks_per_sample * f.batch_spp;
    // even deal: chunks b, b + G, b + 2 G, ...; tapered deal (f.taper_levels > 0): row k hands chunks row0 .. row0 + n_k - 1 to the sub-queues 0 .. n_k - 1 (taper_row_width)
    for (uint32_t k = 0, row0 = 0; row0 < nchunks; k++) {
        const uint32_t nk = f.taper_levels ? taper_row_width(k, gridDim.x, f.taper_levels) : gridDim.x;
        uint32_t pos = blockIdx.x;
        if (f.taper_levels && blockIdx.x < nk) { pos += (k * 2654435761u) % nk; if (pos >= nk) pos -= nk; }      // rotate the row: every sub-queue sees every part of the image over its rows
        const uint32_t c = row0 + pos;
        row0 += nk;
        if (blockIdx.x >= nk || c >= nchunks) continue;                                   // wave-uniform
        uint32_t sl = c / f.chunks_per_sample, cl = c - sl * f.chunks_per_sample;         // wave-uniform (SALU)
        uint32_t pl = cl * kBlock + threadIdx.x;
        if (f.interleave) {                                                               // RTX_OPT_SAMPLE_INTERLEAVE: chunk = 256 / S pixel slots x S consecutive samples, the samples of a pixel in neighbouring
            const uint32_t sh = f.interleave, S = 1u << sh;                                  // lanes (S = 2 .. 16 divides batch_spp).  Changes no path (seeds come from pixel and sample id) and no sum (rad[] is per path)
            const uint32_t per = S * f.chunks_per_sample, sg = c / per, r = c - sg * per;
            sl = sg * S + (threadIdx.x & (S - 1u));
            pl = r * (kBlock >> sh) + (threadIdx.x >> sh);
        }
        const uint32_t pid = sl * f.npl + pl;
        uint32_t x = 0, y = 0;
        const bool valid = slot_to_pixel(f, pl, x, y);
        const uint32_t slot = block_push(valid, &s_n);
        if (valid) {
            uint32_t s0, s1; seed_init(x, y, f.sample_first + sl, f.frame_seed, s0, s1);
            float jx = 0.0f, jy = 0.0f;
            if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }   // RayGen.hlsl:84-85
            f3 o, d; primary_ray(cam, f.width, f.height, x, y, jx, jy, o, d);
            const uint32_t dst = compact ? blockIdx.x * f.qcap + slot : pid;      // compact state: indexed by the queue position
            p.ray_o[dst] = {o.x, o.y, o.z, u2f(s1)};
            p.ray_d[dst] = {d.x, d.y, d.z, 1.0f};
            p.thr[dst] = {1.0f, 1.0f, 1.0f, u2f(s0)};
            p.rad[pid] = {0.0f, 0.0f, 0.0f, 0.0f};
            myq[slot] = pid;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) qcount[blockIdx.x] = s_n;
}

// packet-culling masks of the primary rays, one per 8x8 pixel block of the shard (slot order): they depend on the camera and the scene, not on the
// sample, so they are computed once per render call (one wave per block, lane = record) instead of once per block AND sample inside the raygen
// kernel, where they were more than half of its instructions (at 17 of 64 lanes)
__global__ __launch_bounds__(kBlock) void k_packet_masks(DevScene sc, DevFrame f, const CameraGPU* __restrict__ cam_p, unsigned long long* __restrict__ masks) {
    __shared__ CameraGPU cam;
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    __syncthreads();
    const uint32_t blk = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);      // wave-uniform
    if (blk >= f.npl / 64u) return;
    uint32_t x = 0, y = 0;
    (void)slot_to_pixel(f, blk * 64u, x, y);                                    // slot 0 of the block = its top-left pixel
    const unsigned long long keep = packet_keep_mask(sc, cam, f, x & ~7u, y & ~7u);
    if (lane_id() == 0) masks[blk] = keep;
}

__global__ __launch_bounds__(kBlock) void k_raygen_trace_small(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f, DevPaths p, const CameraGPU* __restrict__ cam_p,
                                                               uint32_t* __restrict__ queue, uint32_t* __restrict__ qcount, uint32_t* __restrict__ gencount,
                                                               const unsigned long long* __restrict__ masks /* k_packet_masks */) {
    extern __shared__ F4 lds[];
    __shared__ CameraGPU cam;
    __shared__ uint32_t s_n[2];
    if (threadIdx.x < 64) ((float*)&cam)[threadIdx.x] = ((const float*)cam_p)[threadIdx.x];
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    uint32_t* myq = queue + (size_t)blockIdx.x * f.qcap;
    const uint32_t nchunks = f.chunks_per_sample * f.batch_spp;
    uint32_t generated = 0;
    for (uint32_t k = 0, row0 = 0; row0 < nchunks; k++) {                                  // the deal of k_raygen: even, or tapered rows (taper_row_width)
        const uint32_t nk = f.taper_levels ? taper_row_width(k, gridDim.x, f.taper_levels) : gridDim.x;
        uint32_t pos = blockIdx.x;
        if (f.taper_levels && blockIdx.x < nk) { pos += (k * 2654435761u) % nk; if (pos >= nk) pos -= nk; }
        const uint32_t c = row0 + pos;
        row0 += nk;
        if (blockIdx.x >= nk || c >= nchunks) continue;                                   // wave-uniform
        const uint32_t sl = c / f.chunks_per_sample, cl = c - sl * f.chunks_per_sample;
        const uint32_t pl = cl * kBlock + threadIdx.x;
        const uint32_t pid = sl * f.npl + pl;
        uint32_t x = 0, y = 0, s0 = 0, s1 = 0;
        const bool valid = slot_to_pixel(f, pl, x, y);
        f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        if (valid) {
            seed_init(x, y, f.sample_first + sl, f.frame_seed, s0, s1);
            float jx = 0.0f, jy = 0.0f;
            if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }
            primary_ray(cam, f.width, f.height, x, y, jx, jy, o, d);
            generated++;
        }
        // pixel block of this wave: lane 0's pixel (slot_to_pixel lays 8x8 blocks out per wave)
        // records that some ray of this wave's 8x8 pixel block can touch (slot_to_pixel lays one block out per wave): precomputed per block
        const unsigned long long km = masks[pl >> 6];
        const unsigned long long keep = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(km >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)km);
        float t, u, v; uint32_t prim;
        traverse_small<false>(sc, small, L, o, d, kTMinCam, valid ? kTMax : 0.0f, t, u, v, prim, sc.nsmall, keep);
        const bool hit = valid && prim != kMissPrim;
        // 43 % of the Cornell camera rays leave the box: their radiance stays zero, so only a hit bit is recorded for them
        const unsigned long long hm = __ballot(hit);
        if (lane_id() == 0) p.hitmask[pid >> 6] = hm;
        if (hit) {
            if (f.max_bounces == 0u) p.rad[pid] = {0.0f, 0.0f, 0.0f, 0.0f};   // otherwise the bounce-0 kernel writes every hit path's radiance slot
            p.ray_o[pid] = {o.x, o.y, o.z, u2f(s1)};
            p.ray_d[pid] = {d.x, d.y, d.z, 1.0f};
            p.thr[pid] = {1.0f, 1.0f, 1.0f, u2f(s0)};
            p.hit[pid] = {t, u, v, u2f(prim)};
        }
        const uint32_t slot = block_push(hit, &s_n[0]);
        if (hit) myq[slot] = pid;
    }
    atomicAdd(&s_n[1], generated);
    __syncthreads();
    if (threadIdx.x == 0) { qcount[blockIdx.x] = s_n[0]; gencount[blockIdx.x] = s_n[1]; }
}

// RTX_OPT_TRACE_COUNTERS: a wave adds its lanes' tallies of node steps and triangle tests to two 64-bit counters (one atomic pair per wave, at its exit)
__device__ __forceinline__ void trace_count_flush(unsigned long long* cnt, uint32_t nodes, uint32_t tris) {
    unsigned long long a = nodes, b = tris;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if (lane_id() == 0) { atomicAdd(cnt, a); atomicAdd(cnt + 1, b); }
}
#ifndef RTX_TRACE_WAVES
#define RTX_TRACE_WAVES 8          // waves per SIMD the DEFAULT-schedule instantiations (SCHED >= 0) of the persistent traversal kernels are compiled for: 62 VGPRs, no spills.  (Uncapped, the
                                  // closest-hit kernel took 69 VGPRs = 7 waves once the 6-B stack entries let eight workgroups fit a CU's LDS.)  The generic instantiations (SCHED -1: experiment
                                  // knobs, work counters) stay uncapped: capped they spill
#endif
// closest hit for every path in this workgroup's sub-queue: reads ray_o/ray_d, writes hit
template <int STK, bool STEAL, int SCHED>   // traversal stack: 0 = LDS column, 1 = private (scratch); STEAL: work stealing between sub-queues (refill_steal);
                                            // SCHED: the wave schedule as a compile-time constant (the default, 6), or -1 = the run-time parameter (experiment knobs)
__global__ __launch_bounds__(kBlock, (SCHED >= 0 ? RTX_TRACE_WAVES : 1)) void k_trace_closest(DevScene sc, const SmallRecPair* __restrict__ small, DevPaths p, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount, uint32_t qcap, float tmin, uint32_t refill_min, uint32_t sched, uint32_t* heads,
                                                                       uint32_t nq, uint32_t merge) {           // nq sub-queues in the launch, `merge` of them per workgroup (MergedQ; 1 with STEAL and on the tiny-scene test path)
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_head;
#ifdef RTX_WAVE_CLOCK
    #define RTX_WAVE_STAMP(K) do { const uint32_t w_ = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); if (tmin != kTMinCam && lane_id() == 0 && w_ < 65536u) g_wgt[2u * w_ + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RTX_WAVE_STAMP(0u);
#endif
    MergedQ M; M.init(qcount, nq, merge);
    const uint32_t n = M.n;
    if (STEAL ? all_exhausted(heads, gridDim.x) : n == 0) return;      // (work stealing: nothing left in the whole launch)
    if (threadIdx.x == 0) s_head = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const bool sorted = !STEAL && p.perm != nullptr && p.oct_in != nullptr && p.out_o != nullptr;      // RTX_OPT_OCTANT_SORT: every sub-queue of this workgroup grouped by direction octant
    if (sorted) for (uint32_t t = 0; t < merge && M.q0 + t < nq; t++) sort_by_key(p.oct_in + (size_t)(M.q0 + t) * qcap, p.perm + (size_t)(M.q0 + t) * qcap, qcount[M.q0 + t], L.stack);
    const uint32_t* myq = queue + (size_t)blockIdx.x * qcap;
    if (SCHED < 0 && sc.nsmall) {                          // tiny scene, un-fused kernels (test path)
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const uint32_t pid = p.out_o ? blockIdx.x * qcap + i : myq[i];       // compact state: the queue position is the index
            const F4 ro = p.ray_o[pid], rd = p.ray_d[pid];
            float t, u, v; uint32_t prim;
            traverse_small<false>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, kTMax, t, u, v, prim, sc.nsmall);
            p.hit[pid] = {t, u, v, u2f(prim)};
        }
        return;
    }
    typename std::conditional<STK == 1, StackPriv, StackLdsT<STK == 2>>::type stk;
    if constexpr (STK != 1) stk.init(L);
    RayLane R; ray_idle(R);
    bool drained = false;
    RaySource W{heads, qcount, gridDim.x, blockIdx.x, n, 0u, 0u};
    uint32_t rng = steal_seed();
    uint32_t cnt_nodes = 0, cnt_tris = 0;
    auto fetch = [&](uint32_t q, uint32_t idx) {
        if (sorted) idx = p.perm[(size_t)q * qcap + idx];
        const uint32_t pid = p.out_o ? q * qcap + idx : queue[(size_t)q * qcap + idx];
        const F4 ro = ld_stream(p.ray_o + pid), rd = ld_stream(p.ray_d + pid);
        ray_begin(R, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, kTMax, pid, true);
    };
    while (STEAL ? refill_steal<true>(R, W, drained, refill_min, rng, fetch) : refill<true>(R, &s_head, n, drained, refill_min, [&](uint32_t idx) { uint32_t q, off; M.locate(idx, q, off); fetch(q, off); })) {
        if (SCHED >= 5) spec_step<false>(sc, L, R, stk, (uint32_t)SCHED);
        else if (sched >= 5u) spec_step<false, decltype(stk), true>(sc, L, R, stk, sched, &cnt_nodes, &cnt_tris);
        else if (sched) voted_step<false>(sc, L, R, stk, sched);
        else { walk_internal<false>(sc, L, R, stk); process_leaf<false>(sc, L, R, stk); }
        if (R.has && R.done) { st_stream(p.hit + R.item, F4{R.bt, R.bu, R.bv, u2f(R.bprim)}); R.has = false; }
    }
    if (SCHED < 0 && sc.trace_cnt) trace_count_flush(sc.trace_cnt, cnt_nodes, cnt_tris);      // RTX_OPT_TRACE_COUNTERS (generic instantiation only)
#ifdef RTX_WAVE_CLOCK
    RTX_WAVE_STAMP(1u);
#endif
}

// any-hit for NEE slot j: visible contributions are added to the path's radiance (a path appears at most once
// per slot, so the read-modify-write needs no atomic and the order of additions per path is fixed)
// SINK 0: the path tracer's NEE rays (visible contributions are added to the path's radiance).  SINK 1: visibility rays of the ReSTIR stages (rtx_restir_wave.hpp):
// the answer goes to occ[pay[entry]] as a byte, 1 = occluded; end points may be anywhere (last frame's samples), so the tiny-scene path tests every record.
template <int STK, bool STEAL, int SCHED, int SINK = 0>
__global__ __launch_bounds__(kBlock, (SCHED >= 0 ? RTX_TRACE_WAVES : 1)) void k_trace_shadow(DevScene sc, const SmallRecPair* __restrict__ small, DevPaths p, const F4* __restrict__ sh_o, const F4* __restrict__ sh_d,
                                                         const F4* __restrict__ sh_c, const uint32_t* __restrict__ shcount, uint32_t qcap, uint32_t refill_min, uint32_t sched, uint32_t* heads,
                                                         uint32_t nq, uint32_t merge, const uint32_t* __restrict__ pay = nullptr, uint8_t* __restrict__ occ = nullptr) {
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_head;
#ifdef RTX_WAVE_CLOCK        // tooling build: the FIRST any-hit launch after a reset (bounce 0's shadow rays, which overlap the stamped closest-hit launch of bounce 1), waves 32768 ...
    #define RTX_WAVE_STAMP_S(K) do { const uint32_t w_ = 32768u + blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); if (!SINK && lane_id() == 0 && w_ < 65536u && g_wgt[2u * w_ + (K)] == 0ull) g_wgt[2u * w_ + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RTX_WAVE_STAMP_S(0u);
#endif
    MergedQ M; M.init(shcount, nq, merge);
    const uint32_t n = M.n;
    if (STEAL ? all_exhausted(heads, gridDim.x) : n == 0) return;
    if (threadIdx.x == 0) s_head = 0;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const size_t qb = (size_t)blockIdx.x * qcap;
    auto finish = [&](size_t gi, bool occluded) {         // gi: index into the launch's shadow-ray arrays (sub-queue * qcap + entry)
        if (SINK) { occ[pay[gi]] = occluded ? 1 : 0; return; }
        if (!occluded) {
            const F4 c = sh_c[gi];
            const uint32_t pid = f2u(c.w);
            F4 r = p.rad[pid];
            r.x = r.x + c.x; r.y = r.y + c.y; r.z = r.z + c.z;
            p.rad[pid] = r;
        }
    };
    if (SCHED < 0 && sc.nsmall) {
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const F4 so = sh_o[qb + i], sd = sh_d[qb + i];
            float t, u, v; uint32_t prim;
            const uint32_t nrec_sh = (SINK || __builtin_amdgcn_ballot_w64(so.w < 0.0f) != 0ull) ? sc.nsmall : sc.nsmall_occ;                        // hull guard, as in k_bounce_small
            traverse_small<true>(sc, small, L, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), SINK ? so.w : fabsf(so.w), sd.w, t, u, v, prim, nrec_sh);   // the short list: NEE segments only
            finish(qb + i, prim != kMissPrim);
        }
        return;
    }
    typename std::conditional<STK == 1, StackPriv, StackLdsT<STK == 2>>::type stk;
    if constexpr (STK != 1) stk.init(L);
    RayLane R; ray_idle(R);
    bool drained = false;
    RaySource W{heads, shcount, gridDim.x, blockIdx.x, n, 0u, 0u};
    uint32_t rng = steal_seed();
    uint32_t cnt_nodes = 0, cnt_tris = 0;
    auto fetch = [&](uint32_t q, uint32_t idx) {
        const uint32_t gi = q * qcap + idx;                               // (< 2^32: the batch cap)
        const F4 so = sh_o[gi], sd = sh_d[gi];
        ray_begin(R, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), so.w, sd.w, gi, false, sc.occluder_cache != 0u, sc.any_order);
    };
    while (STEAL ? refill_steal<false>(R, W, drained, refill_min, rng, fetch) : refill<false>(R, &s_head, n, drained, refill_min, [&](uint32_t idx) { uint32_t q, off; M.locate(idx, q, off); fetch(q, off); })) {
        if (SCHED >= 5) spec_step<true>(sc, L, R, stk, (uint32_t)SCHED);
        else if (sched >= 5u) spec_step<true, decltype(stk), true>(sc, L, R, stk, sched, &cnt_nodes, &cnt_tris);
        else if (sched) voted_step<true>(sc, L, R, stk, sched);
        else { walk_internal<true>(sc, L, R, stk); process_leaf<true>(sc, L, R, stk); }
        if (R.has && R.done) { finish(R.item, R.bprim != kMissPrim); R.has = false; }
    }
    if (SCHED < 0 && sc.trace_cnt) trace_count_flush(sc.trace_cnt + 2, cnt_nodes, cnt_tris);
#ifdef RTX_WAVE_CLOCK
    RTX_WAVE_STAMP_S(1u);
#endif
}

// shade: one thread per queued path (general path: hits come from k_trace_closest, shadow rays go to queues).
// SORT = material-sorted shading: the workgroup's sub-queue is consumed in chunks of kSortChunk entries; each chunk is
// counting-sorted in LDS by the material id of the hit (misses last), so that a wave shades ONE material and its
// branches (emissive / Lambert / GGX strategy, miss) are wave-uniform.  The sort never leaves LDS: the queue index and
// the hit record it reads are needed by the shading anyway.  Results do not depend on the order (per-path state only).
// MEASURED (MI355X, 1080p 16 spp 8 bounces, ms per 2 frames in k_shade): Bistro-class (30 % GGX) 18.1 unsorted vs 23.3
// sorted, Sponza-class 16.1 vs 32.1 — k_shade is HBM-bound, not divergence-bound, and the permutation turns its
// coalesced per-path state streams into gathers; k_trace_shadow gains 4-8 % from the more coherent shadow rays, the
// frame loses 1-10 %.  Round 2, with the path state kept by queue position (the permutation then stays inside a 2048-entry window of
// each stream): still slower, k_shade per frame 7.4 -> 10.6 ms (Sponza-class), 8.6 -> 9.8 ms (Bistro-class, where k_shade is VALU-bound at
// 33 of 64 lanes).  Hence RTX_OPT_SORT_MATERIALS defaults to 0.
// One item of k_shade / k_shade_dense: entry `qi` of the workgroup's sub-queue (valid = the lane has one).  Every lane of the wave goes through the compactions.
template <bool LAMBERT>
__device__ __forceinline__ void shade_item(const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce, uint32_t nee, bool last, size_t qb,
                                           const uint32_t* __restrict__ myq, uint32_t* __restrict__ mynext, uint32_t* s_cnt, bool valid, uint32_t qi, Prof* pf, const float* lds_cdf = nullptr, const LightGPU* lds_lights = nullptr, const MatGPU* lds_mats = nullptr) {
    PathState S; S.pid = 0; S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.thr = mk3(0, 0, 0); S.prev_pdf = 1.0f; S.s0 = S.s1 = 0;
    Surf sf; sf.mat = 0; sf.normal = mk3(0, 0, 1); sf.pos = mk3(0, 0, 0);
    bool shading = false;
#ifndef RTX_NO_LDS_MATS
    const MatGPU* mats = lds_mats ? lds_mats : sc.mats;      // (uniform; k_shade stages a short material table beside the light list)
#else
    const MatGPU* mats = sc.mats;
#endif
    if (valid) {
        const uint32_t pid = myq[qi];
        const uint32_t src = p.out_o ? (uint32_t)qb + qi : pid;               // compact state: hit and path state live at the queue position
        const F4 h = ld_stream(p.hit + src);
        const uint32_t prim = f2u(h.w);
        if (prim != kMissPrim) {                                          // miss: Miss.hlsl:3-11 -> black, terminate
            S = load_path_stream(p, src); S.pid = pid;
            PF_MARK(0); PF_COUNT(1);
            sf = surface(sc, S.o, S.d, h.x, h.y, h.z, prim);
            PF_MARK(1);
            if (sf.mat < sc.nmat) {
                const MatGPU& m = mats[sf.mat];
                if (m.Ke_len > 0.0f) add_emissive(sc, p, S, sf, m, bounce, nee);   // Hit.hlsl:126, Sampler_v6.hlsl:457
                else shading = true;
            }
        }
    }
    const f3 outgoing = -S.d, pos = sf.pos;
    const MatGPU* mp = mats + (shading ? sf.mat : 0u);
    f3 normal = sf.normal;
    const float eta_p = LAMBERT ? 0.0f : transmission_eta(*mp, f.flags, outgoing, normal);          // (extension) hits from behind a dielectric flip the shading normal
    // the view-dependent terms of the mixture BSDF, once per shading point: the NEE samples and the continuation share them (rtx_bsdf.hpp: MixView)
    MixView mvs; const MixView* mv = nullptr;
#ifndef RTX_NO_MIXVIEW          // (A/B build: make VARIANT=nomv VARFLAGS=-DRTX_NO_MIXVIEW)
    if (!LAMBERT) { mvs = mix_view(*mp, f.flags, normal, outgoing, eta_p); mv = &mvs; }
#endif
    PF_MARK(2);
    for (uint32_t j = 0; j < nee; j++) {                                  // NEE: visibility deferred to k_trace_shadow
        bool push = false;
        F4 so = {0, 0, 0, 0}, sd = {0, 0, 0, 0}; f3 con = mk3(0, 0, 0);
        if (shading) { PF_COUNT(3); }
        if (shading) push = nee_sample(sc, *mp, f.flags, nee, S, pos, normal, outgoing, so, sd, con, sc.nsmall != 0u && sf.near_hull, eta_p, mv, lds_cdf, lds_lights);
        PF_MARK(3);
        if (push) { PF_COUNT(4); }
        const size_t seg = (size_t)j * f.qcap * gridDim.x + qb;           // NEE slot j, this workgroup's sub-queue
        const uint32_t slot = block_push(push, &s_cnt[1 + j]);
        if (push) { st_stream(p.sh_o + seg + slot, so); st_stream(p.sh_d + seg + slot, sd); st_stream(p.sh_c + seg + slot, F4{con.x, con.y, con.z, u2f(S.pid)}); }
    }
    PF_MARK(4);
    bool alive = false;
    f3 smp = mk3(0, 0, 1); float P = 0.0f;
    if (shading && !last) { PF_COUNT(5); }
    if (shading && !last) alive = bsdf_continue(*mp, f, bounce, S, normal, outgoing, smp, P, eta_p, mv);
    PF_MARK(5);
    if (alive) { PF_COUNT(6); }
    const uint32_t slot = block_push(alive, &s_cnt[0]);
    if (alive) {
        if (p.out_o) store_path_at_stream(p.out_o, p.out_d, p.out_thr, (uint32_t)qb + slot, S, pos, smp, P);     // densely, at its place in the next queue
        else store_path(p, S, pos, smp, P);
        if (p.oct_out) {                                              // RTX_OPT_OCTANT_SORT: the key the next bounce's closest-hit kernel groups its fetches by
            uint32_t key = (f2u(smp.x) >> 31) | ((f2u(smp.y) >> 31) << 1) | ((f2u(smp.z) >> 31) << 2);            // 1: the direction octant (what ray_octant() will see: sign bits)
            if (p.key_mode == 3u) {                                   // 3: the cell of the ray's ORIGIN on a grid over the scene's box (sc.cell_*: 8 bits in all, split by the box's extents)
                const uint32_t bx = sc.cell_bits & 15u, by = (sc.cell_bits >> 4) & 15u, bz = (sc.cell_bits >> 8) & 15u;
                const uint32_t cx = (uint32_t)fminf(fmaxf((pos.x - sc.cell_o[0]) * sc.cell_s[0], 0.0f), (float)((1u << bx) - 1u));
                const uint32_t cy = (uint32_t)fminf(fmaxf((pos.y - sc.cell_o[1]) * sc.cell_s[1], 0.0f), (float)((1u << by) - 1u));
                const uint32_t cz = (uint32_t)fminf(fmaxf((pos.z - sc.cell_o[2]) * sc.cell_s[2], 0.0f), (float)((1u << bz) - 1u));
                key = cx | (cy << bx) | (cz << (bx + by));
            }
            if (p.key_mode == 5u) key = ((uint32_t)slot * 2654435761u) >> 24;      // 5 (tooling): a hashed key — the scattered fetch without any grouping, to price the fetch alone
            p.oct_out[qb + slot] = (uint8_t)key;
        }
        mynext[slot] = S.pid;
    }
}

constexpr uint32_t kSortChunk = 2048, kSortKeys = 64, kLdsCdf = 256;
// k_shade's dynamic LDS (launch and kernel agree through this one function): the light list (records + CDF) when it has <= 256 entries, the material table behind it while
// both stay within kShadeLds bytes.  RTX_SHADE_LDS (A/B builds) moves the budget; 0 = lights only.
#ifndef RTX_SHADE_LDS
#define RTX_SHADE_LDS 24576
#endif
constexpr uint32_t kShadeLds = RTX_SHADE_LDS;
__host__ __device__ inline void shade_lds_plan(uint32_t nlights, uint32_t nmat, bool sort, uint32_t& lights_bytes, uint32_t& mats_bytes) {
    lights_bytes = (!sort && nlights <= kLdsCdf) ? ((nlights * 84u + 15u) & ~15u) : 0u;
    mats_bytes = (!sort && nmat && lights_bytes + nmat * 160u <= kShadeLds) ? nmat * 160u : 0u;
}
static size_t shade_lds_bytes(const DevScene& sc, bool sort) { uint32_t lb, mb; shade_lds_plan(sc.nlights, sc.nmat, sort, lb, mb); return (size_t)lb + mb; }
#ifndef RTX_SHADE_WAVES
#define RTX_SHADE_WAVES 7          // waves per SIMD k_shade is compiled for: 7 = 72 VGPRs + 1 spilled (GGX) / 66 (Lambert); uncapped: 94 VGPRs, 5 waves; 6: 80, no spills; 8: 64, 9 spilled.
                                   // k_shade per frame, C3 / C5: 7.42 / 8.77 ms uncapped, 6.93 / 8.54 at 6, 7.50 / 8.83 at 8 (round 2); round 4: 6.27 / 6.85 at 6, 6.28 / 6.72 at 7
#endif
template <bool SORT, bool LAMBERT>     // LAMBERT: RTX_FLAG_LAMBERT_ONLY as a compile-time constant (no GGX / transmission code in that instantiation)
__global__ __launch_bounds__(kBlock, RTX_SHADE_WAVES) void k_shade(DevScene sc, DevFrame f_in, DevPaths p, uint32_t bounce,
                                                  const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount,
                                                  uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_count,
                                                  uint32_t* __restrict__ shcounts /* [nee][gridDim.x] */) {
    DevFrame f = f_in;
    f.flags = LAMBERT ? (f_in.flags | 1u) : (f_in.flags & ~1u);
    __shared__ uint32_t s_cnt[1 + kMaxNee];                 // [0] next-queue length, [1 + j] shadow queue j length
    __shared__ uint32_t s_pid[SORT ? kSortChunk : 1], s_sorted[SORT ? kSortChunk : 1], s_hist[SORT ? kSortKeys : 1];
    __shared__ uint8_t s_key[SORT ? kSortChunk : 1];
    // (round 5) a light list of <= 256 entries in LDS — the records (80 B each) and behind them the CDF: NEE's binary search is 1-8 DEPENDENT reads per sample (street scene:
    // 204 lights), the record one more.  Dynamic LDS, sized by the list at launch (shade_lds_bytes): a scene with two lights pays 168 bytes, not a workgroup per CU
    extern __shared__ F4 s_lights[];
    float* s_cdf = (float*)(s_lights + (size_t)sc.nlights * 5u);
    uint32_t lights_bytes, mats_bytes; shade_lds_plan(sc.nlights, sc.nmat, SORT, lights_bytes, mats_bytes);
    const bool cdf_in_lds = lights_bytes != 0u;
    if (cdf_in_lds) {
        for (uint32_t i = threadIdx.x; i < sc.nlights * 5u; i += kBlock) s_lights[i] = ((const F4*)sc.lights)[i];
        for (uint32_t i = threadIdx.x; i < sc.nlights; i += kBlock) s_cdf[i] = sc.cdf[i];
    }
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const float* lds_cdf = cdf_in_lds ? s_cdf : nullptr;
    const LightGPU* lds_lights = cdf_in_lds ? (const LightGPU*)s_lights : nullptr;
    // ... and the material table (160 B each) behind the CDF when launch_shade found room for it (mats_in_lds: what it sized the dynamic LDS for)
    const MatGPU* lds_mats = nullptr;
    if (mats_bytes) {
        F4* dst = (F4*)((char*)s_lights + lights_bytes);
        for (uint32_t i = threadIdx.x; i < sc.nmat * 10u; i += kBlock) dst[i] = ((const F4*)sc.mats)[i];
        lds_mats = (const MatGPU*)dst;
    }
    const uint32_t n = qcount[blockIdx.x];
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const bool last = (bounce + 1u == f.max_bounces);
    const size_t qb = (size_t)blockIdx.x * f.qcap;
    const uint32_t* myq = queue + qb;
    uint32_t* mynext = next_queue + qb;
    const uint32_t chunk = SORT ? kSortChunk : n;
    PF_BEGIN;                                               // (PROFILE build: sections 0 load, 1 surface, 2 emissive / setup, 3 NEE sample, 4 shadow push, 5 BSDF sample, 6 store)
    for (uint32_t cb = 0; cb < n; cb += chunk) {
        const uint32_t cn = (n - cb < chunk) ? n - cb : chunk;
        if (SORT) {
            if (threadIdx.x < kSortKeys) s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cn; i += kBlock) {
                const uint32_t prim = f2u(p.hit[p.out_o ? (uint32_t)qb + cb + i : myq[cb + i]].w);
                const uint32_t key = prim == kMissPrim ? kSortKeys - 1u : (sc.shade[prim].mat % (kSortKeys - 1u));
                s_pid[i] = i; s_key[i] = (uint8_t)key;                 // (the entry's place in the chunk: the queue position is needed too)
                atomicAdd(&s_hist[key], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {                             // exclusive scan of the 64 bucket counts by one wave
                const uint32_t c = s_hist[threadIdx.x];
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += t; }
                s_hist[threadIdx.x] = incl - c;
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cn; i += kBlock) s_sorted[atomicAdd(&s_hist[s_key[i]], 1u)] = s_pid[i];
            __syncthreads();
        }
        for (uint32_t base = threadIdx.x & ~63u; base < cn; base += kBlock) {
            const uint32_t i = base + (threadIdx.x & 63u);
            shade_item<LAMBERT>(sc, f, p, bounce, nee, last, qb, myq, mynext, s_cnt, i < cn, cb + (SORT ? s_sorted[i] : i), pf, lds_cdf, lds_lights, lds_mats);
        }
        if (SORT) __syncthreads();                              // the next chunk overwrites the LDS buffers
    }
    PF_MARK(6);
    PF_FLUSH;
    __syncthreads();
    if (threadIdx.x == 0) next_count[blockIdx.x] = s_cnt[0];
    if (threadIdx.x >= 1 && threadIdx.x <= nee) shcounts[(size_t)(threadIdx.x - 1) * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}

// k_shade with the HITS of the sub-queue compacted before they are shaded (RTX_OPT_SHADE_DENSE).  In an open scene a large part of a bounce's rays leaves the scene
// (the Bistro-class street keeps 86 / 66 / 53 / 44 % of its paths through bounces 1-4) and in k_shade their lanes idle through surface reconstruction, NEE and BSDF
// sampling, which on that scene is VALU-bound work at 33 of 64 lanes (profiles/r02_pmc_bvh.md).  Here the workgroup reads the hit records of 256 entries at a time, pushes
// the entries that hit something into an LDS ring (ballot + one LDS atomic per wave), and shades ring entries 256 at a time — full waves of hits, as the hit ring of the
// fused tiny-scene kernel does.  The permutation stays inside the workgroup's sub-queue, so the state streams stay coalesced (monotone gathers within a 2-KB window; this is
// not the global material sort that was rightly rejected).  Same arithmetic per item; only the order of the entries in the next queue changes, which no result depends on.
// Ring bookkeeping as in k_bounce_small: hits of pass k are counted in s_blk[k % 3] and summed in a register after the pass's barrier, a word is cleared one pass later.
template <bool LAMBERT>
__global__ __launch_bounds__(kBlock, RTX_SHADE_WAVES) void k_shade_dense(DevScene sc, DevFrame f_in, DevPaths p, uint32_t bounce,
                                                        const uint32_t* __restrict__ queue, const uint32_t* __restrict__ qcount,
                                                        uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_count, uint32_t* __restrict__ shcounts) {
    DevFrame f = f_in;
    f.flags = LAMBERT ? (f_in.flags | 1u) : (f_in.flags & ~1u);
    constexpr uint32_t kRing = 512u;
    __shared__ uint32_t s_cnt[1 + kMaxNee], s_list[kRing], s_blk[3];
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < 3) s_blk[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = qcount[blockIdx.x];
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const bool last = (bounce + 1u == f.max_bounces);
    const size_t qb = (size_t)blockIdx.x * f.qcap;
    const uint32_t* myq = queue + qb;
    uint32_t* mynext = next_queue + qb;
    Prof* pf = nullptr;
    uint32_t prod = 0, head = 0, rk = 0;                    // hits pushed / shaded so far, pass number mod 3 (all uniform, in registers)
    for (uint32_t base = 0; base < n; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        bool is_hit = false;
        if (i < n) is_hit = f2u(p.hit[p.out_o ? (uint32_t)qb + i : myq[i]].w) != kMissPrim;        // miss: Miss.hlsl:3-11 -> black, the path ends: nothing to do
        const uint32_t slot = prod + block_push(is_hit, &s_blk[rk]);
        if (is_hit) s_list[slot & (kRing - 1u)] = i;
        __syncthreads();
        prod += s_blk[rk];
        if (threadIdx.x == 0) s_blk[rk == 0u ? 2u : rk - 1u] = 0;
        rk = rk == 2u ? 0u : rk + 1u;
        const bool flush = base + kBlock >= n;
        while (prod - head >= (uint32_t)kBlock || (flush && prod != head)) {                      // uniform
            const uint32_t take = prod - head < (uint32_t)kBlock ? prod - head : (uint32_t)kBlock;
            const bool valid = threadIdx.x < take;
            const uint32_t qi = valid ? s_list[(head + threadIdx.x) & (kRing - 1u)] : 0u;
            shade_item<LAMBERT>(sc, f, p, bounce, nee, last, qb, myq, mynext, s_cnt, valid, qi, pf);
            head += take;
            __syncthreads();                                // every lane has read its ring slot before the next pass overwrites it
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) next_count[blockIdx.x] = s_cnt[0];
    if (threadIdx.x >= 1 && threadIdx.x <= nee) shcounts[(size_t)(threadIdx.x - 1) * gridDim.x + blockIdx.x] = s_cnt[threadIdx.x];
}

// Fused bounce kernel for tiny scenes (sc.nsmall != 0): trace the extension ray, shade, trace the NEE shadow
// rays and add their contributions, sample the BSDF, compact — all in one pass over the workgroup's sub-queue.
// Nothing but the 48-B path state and the queue index moves through HBM; hit records and shadow-ray entries
// stay in registers.  Radiance additions happen in the oracle's order (emissive, then NEE slot 0, 1, ...).
// LAMBERT: RTX_FLAG_LAMBERT_ONLY is a launch constant, so it is a template parameter too: the Lambert-only instantiation carries no GGX code
// (fewer live registers, fewer SGPR spills through v_writelane / v_readlane in the loop).
template <int WAVES, bool HAVE_HIT, bool LAMBERT, bool RING>
__global__ __launch_bounds__(kBlock, WAVES) void k_bounce_small(DevScene sc, const SmallRecPair* __restrict__ small, DevFrame f_in, DevPaths p,
                                                         uint32_t bounce_first, uint32_t bounce_end,
                                                         uint32_t* __restrict__ queue_a, uint32_t* __restrict__ queue_b /* bounce b reads (b & 1 ? b : a), writes the other */,
                                                         uint32_t* __restrict__ qrows /* [bounce][gridDim.x] sub-queue lengths entering each bounce */,
                                                         uint32_t* __restrict__ srows /* [bounce][nee][gridDim.x]: shadow rays traced (statistics) */,
                                                         const uint32_t* __restrict__ order /* workgroup -> sub-queue, longest first (k_order_queues); may be null */) {
    // BOUNCE RANGE: sub-queues are workgroup-private, so bounce b + 1 of sub-queue q depends on bounce b of the SAME sub-queue only.  One
    // launch therefore runs the bounces [bounce_first, bounce_end) of its sub-queue back to back, with a workgroup barrier in between
    // (workgroup-scope release / acquire: the path state and queue entries a bounce writes are read by the same workgroup).  A frame has
    // two fused launches (bounce 0, which reads the primary hits, and bounces 1 .. max_bounces - 1) instead of eight, and no drain /
    // ramp-up between the bounces; the sparsely populated late bounces cost a few loop trips instead of a launch each.
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_cnt[1 + kMaxNee];
    DevFrame f = f_in;
    f.flags = LAMBERT ? (f_in.flags | 1u) : (f_in.flags & ~1u);      // bit 0 known at compile time
    const uint32_t qid = order ? order[blockIdx.x] : blockIdx.x;      // the sub-queue this workgroup owns (input and output)
#ifdef RTX_WAVE_CLOCK        // tooling build: wave start / end stamps of the launch of bounces >= 1 (tools/wave_timeline.py cornell)
    #define RTX_WAVE_STAMP_B(K) do { const uint32_t w_ = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6); if (!HAVE_HIT && lane_id() == 0 && w_ < 65536u) g_wgt[2u * w_ + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RTX_WAVE_STAMP_B(0u);
#endif
    // NEE shadow rays of the workgroup's 256 current items are compacted through LDS, so that the shadow traversal runs on
    // ceil(rays / 64) full waves instead of on every wave at ~2/3 occupancy (only ~65 % of the items get a shadow ray)
    __shared__ F4 s_sho[kBlock], s_shd[kBlock];
    __shared__ uint32_t s_shn[2];                          // ray count, double-buffered by iteration parity
    __shared__ uint8_t s_occ[kBlock];
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_shn[threadIdx.x] = 0;
    // (round 5) the light list (<= 32 records + CDF) and the material table (<= 16 records) of a tiny scene in LDS: with five workgroups per CU the dependent global reads of NEE's
    // CDF search, the light record and the material record are not hidden by other waves (k_shade gained 17 % from the same on the street scene)
#ifndef RTX_NO_SMALL_LDS_TABLES
    constexpr uint32_t kSmallLights = 32, kSmallMats = 16;
    __shared__ F4 s_lt[kSmallLights * 5]; __shared__ float s_ltcdf[kSmallLights]; __shared__ F4 s_mt[kSmallMats * 10];
    const bool lt_lds = sc.nlights && sc.nlights <= kSmallLights, mt_lds = sc.nmat && sc.nmat <= kSmallMats;
    if (lt_lds) { for (uint32_t i = threadIdx.x; i < sc.nlights * 5u; i += kBlock) s_lt[i] = ((const F4*)sc.lights)[i]; if (threadIdx.x < sc.nlights) s_ltcdf[threadIdx.x] = sc.cdf[threadIdx.x]; }
    if (mt_lds) for (uint32_t i = threadIdx.x; i < sc.nmat * 10u; i += kBlock) s_mt[i] = ((const F4*)sc.mats)[i];
    const MatGPU* mats = mt_lds ? (const MatGPU*)s_mt : sc.mats;
    const LightGPU* lds_lights = lt_lds ? (const LightGPU*)s_lt : nullptr; const float* lds_cdf = lt_lds ? s_ltcdf : nullptr;
    // (the shade records and normal matrices as well, 4.3 KB more, cost the fifth workgroup per CU: 18.09 vs 18.03 ms without any table on the same box — not kept)
#else
    const MatGPU* mats = sc.mats; const LightGPU* lds_lights = nullptr; const float* lds_cdf = nullptr;
#endif
    const uint32_t G = gridDim.x;
    uint32_t n = qrows[(size_t)bounce_first * G + qid];
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const uint32_t nee1 = nee ? nee : 1u;
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const size_t qb = (size_t)qid * f.qcap;
    uint32_t par = 0;
    // HIT RING (bounces >= 1).  20-25 % of the extension rays of a bounce leave the open front of the box or end on the light, and their lanes used to idle
    // through surface reconstruction, NEE and BSDF sampling — over half of the kernel's instructions (46-47 of 64 lanes active per VALU instruction,
    // profiles/r02_pmc_cornell.md).  Tracing and shading are therefore decoupled inside the workgroup: the trace phase takes 256 queue entries at a time
    // and pushes only the HITS (path slot + hit record, 20 B) into an LDS ring (ballot + one LDS atomic per wave); whenever the ring holds a full 256 (or
    // the input is exhausted) the shading phase runs on ring entries, i.e. on full waves.  A miss costs nothing beyond its trace.  The price: the path state
    // is read twice (origin / direction for the trace, all 48 B for the shading), which a VALU-bound kernel at 2.4 of 8 TB/s does not notice.
    // All four waves walk through the same phases (every decision is read from LDS after a barrier), so there is no producer / consumer protocol.
    constexpr uint32_t kRing = 512u;                       // < 256 waiting + <= 256 pushed by one trace pass
    __shared__ uint32_t s_rpid[(HAVE_HIT || !RING) ? 1 : kRing];
    __shared__ F4 s_rhit[(HAVE_HIT || !RING) ? 1 : kRing];
    // Ring counters.  s_ring[0] = entries consumed (written by thread 0 between two barriers).  Entries PRODUCED are counted per trace pass in one of three
    // rotating LDS words (pass k pushes into s_rc[k % 3]) and summed in a register (`prod`) after the barrier that ends the pass: the loop condition below then
    // depends on registers and on s_ring[0] only, never on a word another wave may be pushing into.  (With ONE produced-counter read in the condition, a wave
    // that evaluated it late could see a push of the current pass, leave the loop and strand the others at the barrier — a formal race)  A word is
    // cleared by thread 0 after the barrier of pass k + 1 and next pushed into in pass k + 3; it was last read before that barrier.
    __shared__ uint32_t s_ring[1], s_rc[3];
    PF_BEGIN;
    for (uint32_t bounce = bounce_first; bounce < bounce_end; bounce++) {
    const bool last = (bounce + 1u == f.max_bounces);
    const float tmin = bounce_tmin(bounce);
    const uint32_t* myq = ((bounce & 1u) ? queue_b : queue_a) + qb;
    uint32_t* mynext = ((bounce & 1u) ? queue_a : queue_b) + qb;
    if (!HAVE_HIT && RING) { if (threadIdx.x < 3) s_rc[threadIdx.x] = 0; if (threadIdx.x == 3) s_ring[0] = 0; __syncthreads(); }
    uint32_t next_in = 0;                                   // trace phase: next input entry (uniform)
    uint32_t prod = 0, rk = 0;                              // hits pushed by the completed trace passes of this bounce; pass number mod 3 (both uniform)
    for (uint32_t base = 0; (HAVE_HIT || !RING) ? base < n : true; base += kBlock) {     // one trip per 256 queue entries; with the ring: until input and ring are empty
        PathState S; S.pid = 0; S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.thr = mk3(0, 0, 0); S.prev_pdf = 1.0f; S.s0 = S.s1 = 0;
        float t = 0.0f, u = 0.0f, v = 0.0f; uint32_t prim = kMissPrim;
        bool active;
        PF_MARK(0);
        if (HAVE_HIT) {                                   // bounce 0: the primary hit comes from k_raygen_trace_small; every queue entry is a hit
            const uint32_t i = base + threadIdx.x;
            active = i < n;
            if (active) { S = load_path(p, myq[i]); const F4 h = p.hit[S.pid]; t = h.x; u = h.y; v = h.z; prim = f2u(h.w); }
        } else if (!RING) {                               // RTX_OPT_BOUNCE_VARIANT = 1: trace and shade the same 256 entries (lanes whose ray missed idle through the shading)
            const uint32_t i = base + threadIdx.x;
            active = i < n;
            if (active) S = load_path(p, myq[i]);
            traverse_small<false>(sc, small, L, S.o, S.d, tmin, active ? kTMax : 0.0f, t, u, v, prim, sc.nsmall, ~0ull, pf, 1);   // inactive lanes: empty interval
        } else {
            // ---- trace phase: fill the ring until it holds a full workgroup of hits (or the input runs out) ----
            while (next_in < n && prod - s_ring[0] < kBlock) {                // uniform: `prod` is a register, s_ring[0] was written before the last barrier
                const uint32_t i = next_in + threadIdx.x;
                const bool act = i < n;
                uint32_t pid = 0; f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);
                if (act) { pid = myq[i]; const F4 a = p.ray_o[pid], b = p.ray_d[pid]; ro = mk3(a.x, a.y, a.z); rd = mk3(b.x, b.y, b.z); }
                float ht, hu, hv; uint32_t hp;
                traverse_small<false>(sc, small, L, ro, rd, tmin, act ? kTMax : 0.0f, ht, hu, hv, hp, sc.nsmall, ~0ull, pf, 1);   // inactive lanes: empty interval
                const bool hit = act && hp != kMissPrim;
                const uint32_t slot = prod + block_push(hit, &s_rc[rk]);
                if (hit) { s_rpid[slot & (kRing - 1u)] = pid; s_rhit[slot & (kRing - 1u)] = {ht, hu, hv, u2f(hp)}; }
                next_in += kBlock;
                __syncthreads();
                prod += s_rc[rk];                                               // this pass's pushes are complete; the word stays untouched for two more passes
                if (threadIdx.x == 0) s_rc[rk == 0u ? 2u : rk - 1u] = 0;        // the word of the PREVIOUS pass: every wave read it before the barrier above; next used two passes from now
                rk = rk == 2u ? 0u : rk + 1u;
            }
            const uint32_t head = s_ring[0], avail = prod - head;
            if (avail == 0u) break;                                             // input exhausted and ring drained: this bounce is done (uniform)
            const uint32_t take = avail < (uint32_t)kBlock ? avail : (uint32_t)kBlock;
            active = threadIdx.x < take;
            if (active) {
                const uint32_t e = (head + threadIdx.x) & (kRing - 1u);
                S = load_path(p, s_rpid[e]);
                const F4 h = s_rhit[e]; t = h.x; u = h.y; v = h.z; prim = f2u(h.w);
            }
            __syncthreads();                                                    // every lane has read its entry before the slots are released
            if (threadIdx.x == 0) s_ring[0] = head + take;
        }
        PF_MARK(2);
        Surf sf; sf.mat = 0; sf.normal = mk3(0, 0, 1); sf.pos = mk3(0, 0, 0);
        bool shading = false;
        if (active && prim != kMissPrim) {
            PF_COUNT(3);
            sf = surface(sc, S.o, S.d, t, u, v, prim);
            if (sf.mat < sc.nmat) {
                const MatGPU& m = mats[sf.mat];
                if (m.Ke_len > 0.0f) add_emissive(sc, p, S, sf, m, bounce, nee, HAVE_HIT);
                else shading = true;
            } else if (HAVE_HIT) p.rad[S.pid] = {0.0f, 0.0f, 0.0f, 0.0f};
        }
        const f3 outgoing = -S.d, pos = sf.pos;
        const MatGPU* mp = mats + (shading ? sf.mat : 0u);
        f3 normal = sf.normal;
        const float eta_p = LAMBERT ? 0.0f : transmission_eta(*mp, f.flags, outgoing, normal);
        // bounce 0 (HAVE_HIT): nothing has written this path's radiance slot yet: it starts from zero here and is always stored
        bool loaded = HAVE_HIT && shading; F4 radv = {0, 0, 0, 0};
        PF_MARK(3);
        for (uint32_t j = 0; j < nee; j++) {
            bool push = false;
            F4 so = {0, 0, 0, 0}, sd = {0, 0, 1, 0}; f3 con = mk3(0, 0, 0);
            if (shading) { PF_COUNT(4); push = nee_sample(sc, *mp, f.flags, nee, S, pos, normal, outgoing, so, sd, con, sf.near_hull, eta_p, nullptr, lds_cdf, lds_lights); }
            PF_MARK(4);
            const uint32_t slot = block_push(push, &s_shn[par]);
            if (push) { s_sho[slot] = so; s_shd[slot] = sd; }
            __syncthreads();
            PF_MARK(5);
            const uint32_t ns = s_shn[par];
            if ((threadIdx.x & ~63u) < ns) {                                   // wave-uniform: this wave has rays to trace
                const bool mine = threadIdx.x < ns;
                const F4 ro = mine ? s_sho[threadIdx.x] : F4{0, 0, 0, 0}, rd = mine ? s_shd[threadIdx.x] : F4{0, 0, 1, 0};
                float st_, su_, sv_; uint32_t sprim;
                // hull-face shortcut only if no ray of this wave starts near a hull plane (flag in the sign of tmin, TriShade::guard_tau)
                const uint32_t nrec_sh = __builtin_amdgcn_ballot_w64(mine && ro.w < 0.0f) != 0ull ? sc.nsmall : sc.nsmall_occ;
                traverse_small<true>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), fabsf(ro.w), mine ? rd.w : 0.0f, st_, su_, sv_, sprim, nrec_sh, ~0ull, pf, 6);
                if (mine) s_occ[threadIdx.x] = sprim != kMissPrim ? 1 : 0;
            }
            PF_MARK(7);
            if (threadIdx.x == 0) { s_shn[par ^ 1u] = 0; s_cnt[1 + j] += ns; }
            __syncthreads();
            PF_MARK(8);
            if (push && !s_occ[slot]) {
                if (!loaded) { radv = p.rad[S.pid]; loaded = true; }
                radv.x = radv.x + con.x; radv.y = radv.y + con.y; radv.z = radv.z + con.z;
            }
            par ^= 1u;
        }
        if (loaded) p.rad[S.pid] = radv;
        bool alive = false;
        f3 smp = mk3(0, 0, 1); float P = 0.0f;
        if (shading && !last) { PF_COUNT(9); alive = bsdf_continue(*mp, f, bounce, S, normal, outgoing, smp, P, eta_p); }
        if (alive) { PF_COUNT(10); store_path(p, S, pos, smp, P); }
        const uint32_t slot = block_push(alive, &s_cnt[0]);
        if (alive) mynext[slot] = S.pid;
        PF_MARK(9);
        if (!HAVE_HIT && RING) __syncthreads();             // the released ring slots (s_ring[0]) are visible to the next trip's trace phase
    }
    // end of this bounce of the sub-queue: publish its counters; what it wrote (path state, next queue) becomes visible to the workgroup
    __syncthreads();
    n = s_cnt[0];
    if (threadIdx.x == 0) qrows[(size_t)(bounce + 1u) * G + qid] = n;
    if (threadIdx.x >= 1 && threadIdx.x <= nee) srows[((size_t)bounce * nee1 + (threadIdx.x - 1)) * G + qid] = s_cnt[threadIdx.x];
    __syncthreads();
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    }
    PF_FLUSH;
#ifdef RTX_WAVE_CLOCK
    RTX_WAVE_STAMP_B(1u);
#endif
}

// Fused bounce kernel of the GENERAL (BVH) path: for its private sub-queue a workgroup runs, bounce after bounce in ONE launch,
//   phase 1  closest-hit traversal (persistent waves with refill, as k_trace_closest) — hits go to p.hit and, compacted, to a hit list
//   phase 2  shading of the hit list (as k_shade: surface, emissive MIS, NEE sample -> shadow entries, BSDF sample, RR, compaction)
//   phase 3  any-hit traversal of the shadow entries, radiance added per NEE slot in order (as k_trace_shadow)
// with workgroup barriers in between (sub-queues are workgroup-private: bounce b + 1 of a sub-queue depends on bounce b of the same sub-queue
// only, exactly as in k_bounce_small).  The idea: the traversal phases are VALU-bound (VALU busy 1.00 / 0.80, profiles/r02_pmc_sponza.md) and the
// shading phase is HBM-bound (3.6-4 TB/s at VALU busy 0.46), so with the workgroups of a launch in different phases at any moment the two resources
// would be used at the same time; shading iterates over HITS only; 1 launch per frame instead of 24-25.
// MEASURED (MI355X, round 2): bit-identical to the separate kernels (every general-path test runs both), and SLOWER — C3 51.3 vs 47.7 ms, C5 46.2 vs
// 44.4 ms per frame (58.5 / 53.0 before the kernel was built for 5 waves per SIMD and reduced to one wave schedule).  Why (rocprofv3 --pmc): the
// same VALU work (+7 % instructions from SGPR spill traffic in the loops) runs at 79 % VALU-busy instead of 100 %: waves are parked 56 % of the
// time, because a wave that has finished its share of a phase keeps its SIMD slot while it waits at the barrier for the slowest wave of its
// workgroup (in the separate kernels it retires and the next workgroup's wave takes the slot), and 86 VGPRs (75 for the traversal kernels) leave
// fewer waves to cover that.  Hence RTX_OPT_FUSED_BVH defaults to 0; the kernel stays as the measured alternative.
// Arithmetic and the order of radiance additions per path are those of the separate kernels.
template <int STK>
__global__ __launch_bounds__(kBlock, 5) void k_bounce_bvh(DevScene sc, DevFrame f, DevPaths p, uint32_t bounce_first, uint32_t bounce_end,
                                                        uint32_t* __restrict__ queue_a, uint32_t* __restrict__ queue_b, uint32_t* __restrict__ hitq,
                                                        uint32_t* __restrict__ qrows, uint32_t* __restrict__ srows, const uint32_t* __restrict__ order) {
    extern __shared__ F4 lds[];
    __shared__ uint32_t s_head, s_nh;
    __shared__ uint32_t s_cnt[1 + kMaxNee];
    const uint32_t G = gridDim.x;
    const uint32_t qid = order ? order[blockIdx.x] : blockIdx.x;
    const uint32_t nee = sc.nlights ? f.nee_samples : 0u;
    const uint32_t nee1 = nee ? nee : 1u;
    const TraceLds L = stage_lds(sc, lds);
    if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_head = 0; s_nh = 0; }
    __syncthreads();
    const size_t qb = (size_t)qid * f.qcap;
    uint32_t* myhits = hitq + qb;
    uint32_t n = qrows[(size_t)bounce_first * G + qid];
    typename std::conditional<STK == 1, StackPriv, StackLdsT<STK == 2>>::type stk;
    if constexpr (STK != 1) stk.init(L);
    for (uint32_t bounce = bounce_first; bounce < bounce_end; bounce++) {
        const bool last = (bounce + 1u == f.max_bounces);
        const float tmin = bounce_tmin(bounce);
        const uint32_t* myq = ((bounce & 1u) ? queue_b : queue_a) + qb;
        uint32_t* mynext = ((bounce & 1u) ? queue_a : queue_b) + qb;
        // ---- phase 1: closest hit for every entry of the sub-queue ----
        if (n) {
            RayLane R; ray_idle(R);
            bool drained = false;
            while (refill<true>(R, &s_head, n, drained, sc.refill_min, [&](uint32_t idx) {
                       const uint32_t pid = myq[idx];
                       const F4 ro = p.ray_o[pid], rd = p.ray_d[pid];
                       ray_begin(R, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, kTMax, pid, true);
                   })) {
                spec_step<false>(sc, L, R, stk, sc.trace_sched);               // (the fused kernel carries the speculative voted schedule only: RTX_OPT_TRACE_SCHED 5-7)
                const bool fin = R.has && R.done;
                const bool hit = fin && R.bprim != kMissPrim;
                if (__ballot(fin) != 0ull) {                                   // wave-uniform
                    const uint32_t slot = block_push(hit, &s_nh);              // only hits reach the shading phase (a miss ends the path: Miss.hlsl:3-11)
                    if (hit) { p.hit[R.item] = {R.bt, R.bu, R.bv, u2f(R.bprim)}; myhits[slot] = R.item; }
                    if (fin) R.has = false;
                }
            }
        }
        __syncthreads();
        const uint32_t nh = s_nh;
        // ---- phase 2: shade the hits ----
        for (uint32_t base = threadIdx.x & ~63u; base < nh; base += kBlock) {
            const uint32_t i = base + (threadIdx.x & 63u);
            PathState S; S.pid = 0; S.o = mk3(0, 0, 0); S.d = mk3(0, 0, 1); S.thr = mk3(0, 0, 0); S.prev_pdf = 1.0f; S.s0 = S.s1 = 0;
            Surf sf; sf.mat = 0; sf.normal = mk3(0, 0, 1); sf.pos = mk3(0, 0, 0);
            bool shading = false;
            if (i < nh) {
                const uint32_t pid = myhits[i];
                const F4 h = p.hit[pid];
                S = load_path(p, pid);
                sf = surface(sc, S.o, S.d, h.x, h.y, h.z, f2u(h.w));
                if (sf.mat < sc.nmat) {
                    const MatGPU& m = sc.mats[sf.mat];
                    if (m.Ke_len > 0.0f) add_emissive(sc, p, S, sf, m, bounce, nee);
                    else shading = true;
                }
            }
            const f3 outgoing = -S.d, pos = sf.pos;
            const MatGPU* mp = sc.mats + (shading ? sf.mat : 0u);
            f3 normal = sf.normal;
            const float eta_p = transmission_eta(*mp, f.flags, outgoing, normal);
            for (uint32_t j = 0; j < nee; j++) {
                bool push = false;
                F4 so = {0, 0, 0, 0}, sd = {0, 0, 0, 0}; f3 con = mk3(0, 0, 0);
                if (shading) push = nee_sample(sc, *mp, f.flags, nee, S, pos, normal, outgoing, so, sd, con, false, eta_p);
                const size_t seg = (size_t)j * f.qcap * G + qb;
                const uint32_t slot = block_push(push, &s_cnt[1 + j]);
                if (push) { p.sh_o[seg + slot] = so; p.sh_d[seg + slot] = sd; p.sh_c[seg + slot] = {con.x, con.y, con.z, u2f(S.pid)}; }
            }
            bool alive = false;
            f3 smp = mk3(0, 0, 1); float P = 0.0f;
            if (shading && !last) alive = bsdf_continue(*mp, f, bounce, S, normal, outgoing, smp, P, eta_p);
            if (alive) store_path(p, S, pos, smp, P);
            const uint32_t slot = block_push(alive, &s_cnt[0]);
            if (alive) mynext[slot] = S.pid;
        }
        __syncthreads();
        // ---- phase 3: NEE visibility, slot by slot (a path appears at most once per slot: plain read-modify-write, fixed order of additions) ----
        for (uint32_t j = 0; j < nee; j++) {
            const uint32_t ns = s_cnt[1 + j];
            if (threadIdx.x == 0) s_head = 0;
            __syncthreads();
            if (ns) {
                const size_t sb = (size_t)j * f.qcap * G + qb;
                RayLane R; ray_idle(R);
                bool drained = false;
                while (refill<false>(R, &s_head, ns, drained, sc.refill_min, [&](uint32_t idx) {
                           const F4 so = p.sh_o[sb + idx], sd = p.sh_d[sb + idx];
                           ray_begin(R, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), fabsf(so.w), sd.w, idx, false, false, sc.any_order);
                       })) {
                    spec_step<true>(sc, L, R, stk, sc.trace_sched);
                    if (R.has && R.done) {
                        if (R.bprim == kMissPrim) {                            // visible
                            const F4 c = p.sh_c[sb + R.item];
                            const uint32_t pid = f2u(c.w);
                            F4 r = p.rad[pid];
                            r.x = r.x + c.x; r.y = r.y + c.y; r.z = r.z + c.z;
                            p.rad[pid] = r;
                        }
                        R.has = false;
                    }
                }
            }
            __syncthreads();
        }
        // ---- end of this bounce: publish the counters, reset for the next one ----
        n = s_cnt[0];
        if (threadIdx.x == 0) { qrows[(size_t)(bounce + 1u) * G + qid] = n; s_head = 0; s_nh = 0; }
        if (threadIdx.x >= 1 && threadIdx.x <= nee) srows[((size_t)bounce * nee1 + (threadIdx.x - 1)) * G + qid] = s_cnt[threadIdx.x];
        __syncthreads();
        if (threadIdx.x <= kMaxNee) s_cnt[threadIdx.x] = 0;
        __syncthreads();
    }
}

// Longest sub-queue first.  The sub-queues of a batch differ in length by ~12 % (std; each is a sample of ~84 of the image's 8100
// 256-pixel regions, 43 % of which are background on the Cornell view) and a launch has only ~6 workgroups per resident slot, so in
// blockIdx order the last dispatch round is ragged: 3.4 of 4 waves per SIMD resident on average.  The hardware dispatches
// workgroups in blockIdx order, so handing the longest sub-queues out first (LPT list scheduling) lets the short ones fill the end.
// One workgroup, counting sort by length into 1024 buckets (descending); the order inside a bucket is arbitrary (LDS atomics) and
// never matters: every sub-queue is processed independently, results and statistics do not depend on the dispatch order.
__global__ __launch_bounds__(1024) void k_order_queues(const uint32_t* __restrict__ qcount, uint32_t G, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_max, s_hist[1024], s_scan[1024];
    if (threadIdx.x == 0) s_max = 0;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t m = 0;
    for (uint32_t g = threadIdx.x; g < G; g += 1024u) m = max(m, qcount[g]);
    atomicMax(&s_max, m);
    __syncthreads();
    const uint64_t mx = s_max ? s_max : 1u;
    for (uint32_t g = threadIdx.x; g < G; g += 1024u) atomicAdd(&s_hist[1023u - (uint32_t)((uint64_t)qcount[g] * 1023u / mx)], 1u);
    __syncthreads();
    uint32_t v = s_hist[threadIdx.x];
    s_scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                 // inclusive scan (Hillis-Steele)
        const uint32_t t = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += t;
        __syncthreads();
    }
    s_hist[threadIdx.x] = s_scan[threadIdx.x] - v;            // bucket start
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < G; g += 1024u) order[atomicAdd(&s_hist[1023u - (uint32_t)((uint64_t)qcount[g] * 1023u / mx)], 1u)] = g;
}

// ---------------------------------------------------------------------------------------------
// accumulate: gPermanentData running sum + count, RayGen_v6_pass3.hlsl:383-405.  Fixed order: the
// batch's samples are added in sample order, batches run in order on the stream.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accumulate(DevFrame f, DevPaths p, F4* __restrict__ accum) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        if (!slot_to_pixel(f, pl, x, y)) continue;
        F4 a = accum[(size_t)y * f.width + x];
        for (uint32_t s = 0; s < f.batch_spp; s++) {
            const size_t pid = (size_t)s * f.npl + pl;
            F4 r = {0.0f, 0.0f, 0.0f, 0.0f};
            if (!p.hitmask || ((p.hitmask[pid >> 6] >> (pl & 63u)) & 1ull)) r = p.rad[pid];
            const f3 rv = mk3(r.x, r.y, r.z);
            if (finite3(rv)) { a.x = a.x + rv.x; a.y = a.y + rv.y; a.z = a.z + rv.z; a.w = a.w + 1.0f; }
        }
        accum[(size_t)y * f.width + x] = a;
    }
}

// sRGB8 output: RayGen_v6_pass3.hlsl:405,428-441 + Common_v6.hlsl:353-376
__global__ __launch_bounds__(kBlock) void k_srgb8(const F4* __restrict__ accum, uint32_t npix, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    const F4 a = accum[i];
    const float cnt = maxf_(a.w, 1.0f);
    float c[3] = {a.x / cnt, a.y / cnt, a.z / cnt};
    if (is_nan(c[0]) || is_nan(c[1]) || is_nan(c[2])) { c[0] = 1.0f; c[1] = 0.0f; c[2] = 1.0f; }
    if (is_inf(c[0]) || is_inf(c[1]) || is_inf(c[2])) { c[0] = 0.0f; c[1] = 1.0f; c[2] = 1.0f; }
    uint32_t px = 0xFF000000u;
    for (int k = 0; k < 3; k++) {
        float v = c[k] <= 0.0031308f ? 12.92f * c[k] : 1.055f * pow_(c[k], 1.0f / 2.4f) - 0.055f;
        v = saturate(v);
        px |= ((uint32_t)(int)(v * 255.0f + 0.5f)) << (8 * k);
    }
    out[i] = px;
}

// Debug output layers (the reference's gOutput is a 30-layer texture array and 'C' cycles m_displayLevels = {0, 10..17, 20..28}: Renderer.h:298-299,
// Renderer.cpp:690-698, 748-754; its live shaders only ever write layer 0, the others show whatever was left there).  Here layers 10-17 are
// DEFINED: first-hit attributes of the pixel-corner primary ray (jitter-free, RayGen_v6_pass1.hlsl:80-95), one thread per pixel:
//   10 shading normal n/2 + 1/2   11 depth t / (1 + t)   12 material id (hashed colour)   13 Kd (fp16-rounded, as shaded)
//   14 instance id (hashed colour)   15 barycentrics (1-u-v, u, v)   16 Ke / (1 + Ke)   17 (roughness, metallic, dissolve)
// a miss is black; layers 20-28 stay black (never written by the reference either).  Linear values, quantised like layer 0's alpha: v * 255 + 0.5.
__device__ __forceinline__ uint32_t hash_colour(uint32_t id) {
    uint32_t h = id * 2654435761u + 0x9E3779B9u; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    return 0xFF000000u | (0x404040u + (h & 0x00BFBFBFu));
}
__device__ __forceinline__ uint32_t pack_rgb8(float r, float g, float b) {
    const float c[3] = {saturate(r), saturate(g), saturate(b)};
    uint32_t px = 0xFF000000u;
    for (int k = 0; k < 3; k++) px |= ((uint32_t)(int)(c[k] * 255.0f + 0.5f)) << (8 * k);
    return px;
}
__global__ __launch_bounds__(kBlock) void k_debug_layer(DevScene sc, const SmallRecPair* __restrict__ small, uint32_t width, uint32_t height, const CameraGPU* __restrict__ cam, uint32_t layer, uint32_t* __restrict__ out) {
    extern __shared__ F4 lds[];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < width * height; i += stride) {
        const uint32_t x = i % width, y = i / width;
        f3 o, d; primary_ray(*cam, width, height, x, y, 0.0f, 0.0f, o, d);
        float t, u, v; uint32_t prim;
        trace_ray<false>(sc, small, L, o, d, kTMinCam, kTMax, t, u, v, prim);
        uint32_t px = 0xFF000000u;
        if (prim != kMissPrim && layer >= 10u && layer <= 17u) {
            const Surf sf = surface(sc, o, d, t, u, v, prim);
            const bool hm = sf.mat < sc.nmat;
            const MatGPU& m = sc.mats[hm ? sf.mat : 0u];
            switch (layer) {
            case 10u: px = pack_rgb8(sf.normal.x * 0.5f + 0.5f, sf.normal.y * 0.5f + 0.5f, sf.normal.z * 0.5f + 0.5f); break;
            case 11u: { const float z = t / (1.0f + t); px = pack_rgb8(z, z, z); break; }
            case 12u: px = hash_colour(sf.mat); break;
            case 13u: px = hm ? pack_rgb8(m.Kd[0], m.Kd[1], m.Kd[2]) : px; break;
            case 14u: px = hash_colour(sf.inst + 0x51ED27u); break;
            case 15u: px = pack_rgb8(1.0f - u - v, u, v); break;
            case 16u: px = hm ? pack_rgb8(m.Ke[0] / (1.0f + m.Ke[0]), m.Ke[1] / (1.0f + m.Ke[1]), m.Ke[2] / (1.0f + m.Ke[2])) : px; break;
            default: px = hm ? pack_rgb8(m.Pr, m.Pm, m.alpha) : px; break;
            }
        }
        out[i] = px;
    }
}

// tile slabs for the multi-GPU gather
__global__ __launch_bounds__(kBlock) void k_pack_tiles(DevFrame f, const F4* __restrict__ accum, F4* __restrict__ slab) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t pl = blockIdx.x * kBlock + threadIdx.x; pl < f.npl; pl += stride) {
        uint32_t x, y;
        F4 v = {0, 0, 0, 0};
        if (slot_to_pixel(f, pl, x, y)) v = accum[(size_t)y * f.width + x];
        slab[pl] = v;
    }
}
__global__ __launch_bounds__(kBlock) void k_unpack_tiles(DevFrame f, uint32_t nshards, const F4* __restrict__ slabs, F4* __restrict__ accum) {
    const uint32_t stride = gridDim.x * kBlock;
    const uint32_t total = f.npl * nshards;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        DevFrame g = f; g.shard_rank = i / f.npl; g.shard_count = nshards;
        uint32_t x, y;
        if (slot_to_pixel(g, i - g.shard_rank * f.npl, x, y)) accum[(size_t)y * f.width + x] = slabs[i];
    }
}

// ---------------------------------------------------------------------------------------------
// kernel-level debug entry points (parity tests): same device functions as the render loop
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_dbg_trace(DevScene sc, const SmallRecPair* __restrict__ small, const F4* __restrict__ rays, uint32_t n, int any, F4* __restrict__ hits) {
    extern __shared__ F4 lds[];
    const TraceLds L = stage_lds(sc, lds);
    __syncthreads();
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const F4 ro = rays[2 * i], rd = rays[2 * i + 1];
        float t, u, v; uint32_t prim;
        if (any == 2) traverse_stats<false>(sc, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        else if (any == 3) traverse_stats<true>(sc, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);      // any-hit in the order sc.any_order, counted (u = node steps, v = triangle tests)
        else if (any) trace_ray<true>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        else trace_ray<false>(sc, small, L, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w, t, u, v, prim);
        hits[i] = {t, u, v, u2f(prim)};
    }
}
__global__ __launch_bounds__(kBlock) void k_dbg_surface(DevScene sc, const F4* __restrict__ rays, const F4* __restrict__ hits, uint32_t n, F4* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const F4 h = hits[i];
    F4 z = {0, 0, 0, 0};
    out[4 * i] = z; out[4 * i + 1] = z; out[4 * i + 2] = z; out[4 * i + 3] = z;
    if (f2u(h.w) == kMissPrim) { out[4 * i].w = u2f(kMissMat); return; }
    const F4 ro = rays[2 * i], rd = rays[2 * i + 1];
    const Surf s = surface(sc, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), h.x, h.y, h.z, f2u(h.w));
    out[4 * i] = {s.pos.x, s.pos.y, s.pos.z, u2f(s.mat)};
    out[4 * i + 1] = {s.normal.x, s.normal.y, s.normal.z, s.area};
    out[4 * i + 2] = {u2f(s.inst), s.flat.x, s.flat.y, s.flat.z};
}
__global__ __launch_bounds__(kBlock) void k_dbg_bsdf_eval(DevScene sc, uint32_t mat, uint32_t flags, const float* __restrict__ in9, uint32_t n, float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* q = in9 + (size_t)i * 9; float* o = out8 + (size_t)i * 8;
    f3 F; float P, pd, ps;
    f3 nrm = mk3(q[0], q[1], q[2]); const f3 wo = mk3(q[3], q[4], q[5]);
    const float eta_p = transmission_eta(sc.mats[mat], flags, wo, nrm);
    if (flags & 0x80000000u) {                       // the form k_shade runs: view terms computed once (MixView), mixture evaluated against them — must give the same bits
        const uint32_t fl = flags & 0x7FFFFFFFu;
        const MixView mv = mix_view(sc.mats[mat], fl, nrm, wo, eta_p);
        bsdf_mixture_v(sc.mats[mat], fl, mv, nrm, mk3(q[6], q[7], q[8]), wo, F, P, eta_p); pd = mv.pd; ps = mv.ps;
    } else bsdf_mixture(sc.mats[mat], flags, nrm, mk3(q[6], q[7], q[8]), wo, F, P, pd, ps, eta_p);
    o[0] = F.x; o[1] = F.y; o[2] = F.z; o[3] = P; o[4] = pd; o[5] = ps; o[6] = eta_p; o[7] = 0.0f;
}
__global__ __launch_bounds__(kBlock) void k_dbg_bsdf_sample(DevScene sc, uint32_t mat, uint32_t flags, const float* __restrict__ in8, uint32_t n, float* __restrict__ out8) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* q = in8 + (size_t)i * 8; float* o = out8 + (size_t)i * 8;
    uint32_t s0 = f2u(q[6]), s1 = f2u(q[7]);
    f3 nrm = mk3(q[0], q[1], q[2]); const f3 wo = mk3(q[3], q[4], q[5]);
    const float eta_p = transmission_eta(sc.mats[mat], flags, wo, nrm);
    const uint32_t fl = flags & 0x7FFFFFFFu;
    const uint32_t st = (flags & 0x80000000u) ? select_strategy_v(sc.mats[mat], mix_view(sc.mats[mat], fl, nrm, wo, eta_p), fl, s0, s1, eta_p) : select_strategy(sc.mats[mat], wo, nrm, flags, s0, s1, eta_p);
    const f3 wi = sample_bsdf(sc.mats[mat], st, wo, nrm, s0, s1, eta_p);
    o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = u2f(st); o[4] = u2f(s0); o[5] = u2f(s1); o[6] = 0.0f; o[7] = 0.0f;
}
__global__ void k_dbg_tea(uint32_t s0, uint32_t s1, uint32_t n, float* __restrict__ out, uint32_t* __restrict__ seed_out) {
    if (threadIdx.x || blockIdx.x) return;
    for (uint32_t i = 0; i < n; i++) out[i] = tea_next(s0, s1);
    seed_out[0] = s0; seed_out[1] = s1;
}
__global__ __launch_bounds__(kBlock) void k_dbg_primary(DevFrame f, const CameraGPU* __restrict__ cam, uint32_t sample_id, F4* __restrict__ rays) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= f.width * f.height) return;
    const uint32_t x = i % f.width, y = i / f.width;
    uint32_t s0, s1; seed_init(x, y, sample_id, f.frame_seed, s0, s1);
    float jx = 0.0f, jy = 0.0f;
    if (f.flags & 2u) { jx = tea_next(s0, s1); jy = tea_next(s0, s1); }
    f3 o, d; primary_ray(*cam, f.width, f.height, x, y, jx, jy, o, d);
    rays[2 * i] = {o.x, o.y, o.z, kTMinCam};
    rays[2 * i + 1] = {d.x, d.y, d.z, kTMax};
}


// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// GPU refit of the wide BVH after a transform-only commit (the reference refits its TLAS every frame: Renderer.cpp:594,
// TopLevelASGenerator.cpp:149-250).  Topology, slot assignment and triangle order stay; k_refit_tris re-derives the world-space
// triangles from the object-space vertices with the host's operation order (xform_point: bit-identical TriGPU records, so the
// triangle tests still match the oracle's), k_refit_nodes re-derives and re-quantises the child boxes level by level, deepest
// first.  Quantisation is conservative by construction: lo - p is rounded DOWN before floor(), hi - p UP before ceil(), and
// 2^e is chosen with 255 * 2^e >= extent, so the decoded planes bracket the float boxes exactly as the host's double-checked
// build does.
// ---------------------------------------------------------------------------------------------
// directed-rounding stand-ins (HIP has no __fsub_rd / __fsub_ru here): the neighbours of the round-to-nearest result bracket the
// exact difference (|exact - fl| <= half a spacing), at the price of at most one extra spacing of slack
__device__ __forceinline__ float next_below(float x) { uint32_t b = f2u(x); if (x > 0.0f) b--; else if (x < 0.0f) b++; else b = 0x80000001u; return u2f(b); }
__device__ __forceinline__ float next_above(float x) { uint32_t b = f2u(x); if (x > 0.0f) b++; else if (x < 0.0f) b--; else b = 0x00000001u; return u2f(b); }
__device__ __forceinline__ float sub_down(float a, float b) { return next_below(a - b); }
__device__ __forceinline__ float sub_up(float a, float b) { return next_above(a - b); }

// PARTIAL refit (round 4): `moved` != nullptr names the instances whose transform changed since the last commit.  Only their triangles are re-derived (tri_dirty[s] says
// which leaf entries those were), and k_refit_nodes re-quantises only nodes with a dirty triangle or a dirty child (node_dirty), taking the float box of a clean child
// from node_aabb, which the previous refit left there.  A frame that moves one small instance of a large scene (the reference's own loop: Renderer.cpp:444-452) then costs
// the launches, not the scene.  The padding scale only grows in a partial refit (the untouched boxes keep the padding they were built with: still conservative).
__global__ __launch_bounds__(kBlock) void k_refit_tris(TriGPU* __restrict__ tris, uint32_t ntris, const TriShade* __restrict__ shade, const InstGPU* __restrict__ insts,
                                                       const F4* __restrict__ objtris, uint32_t* __restrict__ scale_bits, const uint32_t* __restrict__ moved, uint8_t* __restrict__ tri_dirty) {
    __shared__ uint32_t s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    float amax = 0.0f;
    bool work = s < ntris;
    uint32_t g = 0, inst = 0;
    if (work) { g = f2u(tris[s].v0.w); inst = shade[g].inst; }
    if (work && moved) { work = moved[inst] != 0u; tri_dirty[s] = work ? 1 : 0; }
    if (work) {
        const float* M = insts[inst].o2w;
        const F4 a = objtris[(size_t)g * 3], b = objtris[(size_t)g * 3 + 1], c = objtris[(size_t)g * 3 + 2];
        const f3 w0 = xform_point(M, mk3(a.x, a.y, a.z)), w1 = xform_point(M, mk3(b.x, b.y, b.z)), w2 = xform_point(M, mk3(c.x, c.y, c.z));
        const f3 e1 = w1 - w0, e2 = w2 - w0;
        tris[s].v0 = {w0.x, w0.y, w0.z, u2f(g)};
        tris[s].e1 = {e1.x, e1.y, e1.z, tri_det_floor(e1, e2)};        // (as the host build: same operations, same bits)
        tris[s].e2 = {e2.x, e2.y, e2.z, 0.0f};
        amax = fmaxf(fmaxf(fmaxf(fabsf(w0.x), fabsf(w0.y)), fmaxf(fabsf(w0.z), fabsf(w1.x))), fmaxf(fmaxf(fabsf(w1.y), fabsf(w1.z)), fmaxf(fmaxf(fabsf(w2.x), fabsf(w2.y)), fabsf(w2.z))));
    }
    atomicMax(&s_max, f2u(amax));                      // non-negative floats order like their bit patterns
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(scale_bits, s_max);
}

__global__ __launch_bounds__(kBlock) void k_refit_nodes(Node8GPU* __restrict__ nodes, uint32_t first, uint32_t count, const TriGPU* __restrict__ tris,
                                                        F4* __restrict__ node_aabb /* 2 per node: min, max */, const uint32_t* __restrict__ scale_bits,
                                                        const uint8_t* __restrict__ tri_dirty, uint8_t* __restrict__ node_dirty) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t n = first + i;
    Node8GPU N = nodes[n];
    const float pad = 2e-6f * u2f(*scale_bits);          // the host build's bvh_pad (rtx_scene_host.cpp)
    const uint32_t imask = N.e_imask >> 24;
    if (tri_dirty) {                                      // partial refit: anything below this node touched?
        bool dirty = false;
        const uint32_t nint = (uint32_t)__builtin_popcount(imask);
        for (uint32_t k = 0; k < nint; k++) dirty = dirty || node_dirty[N.child_base + k] != 0;
        const uint32_t nleaf = (uint32_t)__builtin_popcount(N.trivalid);
        for (uint32_t k = 0; k < nleaf; k++) dirty = dirty || tri_dirty[N.tri_base + k] != 0;
        node_dirty[n] = dirty ? 1 : 0;
        if (!dirty) return;                               // node_aabb[n] and the quantised node stay what the last refit made them
    }
    float cmn[8][3], cmx[8][3];
    float bmn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bmx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t rank = 0, tri_at = N.tri_base, used = 0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) {
        const uint32_t nib = (N.trivalid >> (4 * sl)) & 0xfu;
        for (int a = 0; a < 3; a++) { cmn[sl][a] = 0.0f; cmx[sl][a] = 0.0f; }
        if ((imask >> sl) & 1u) {
            const F4 mn = node_aabb[2 * (size_t)(N.child_base + rank)], mx = node_aabb[2 * (size_t)(N.child_base + rank) + 1];
            rank++;
            cmn[sl][0] = mn.x; cmn[sl][1] = mn.y; cmn[sl][2] = mn.z; cmx[sl][0] = mx.x; cmx[sl][1] = mx.y; cmx[sl][2] = mx.z;
        } else if (nib) {
            float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            const uint32_t cnt = (uint32_t)__builtin_popcount(nib);
            for (uint32_t k = 0; k < cnt; k++, tri_at++) {
                const TriGPU T = tris[tri_at];
                const float v[3][3] = {{T.v0.x, T.v0.y, T.v0.z}, {T.v0.x + T.e1.x, T.v0.y + T.e1.y, T.v0.z + T.e1.z}, {T.v0.x + T.e2.x, T.v0.y + T.e2.y, T.v0.z + T.e2.z}};
                for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], fminf(v[0][a], fminf(v[1][a], v[2][a]))); mx[a] = fmaxf(mx[a], fmaxf(v[0][a], fmaxf(v[1][a], v[2][a]))); }
            }
            for (int a = 0; a < 3; a++) { cmn[sl][a] = mn[a] - pad; cmx[sl][a] = mx[a] + pad; }
        } else continue;
        used |= 1u << sl;
        for (int a = 0; a < 3; a++) { bmn[a] = fminf(bmn[a], cmn[sl][a]); bmx[a] = fmaxf(bmx[a], cmx[sl][a]); }
    }
    if (!used) { for (int a = 0; a < 3; a++) { bmn[a] = 0.0f; bmx[a] = 0.0f; } }
    node_aabb[2 * (size_t)n] = {bmn[0], bmn[1], bmn[2], 0.0f}; node_aabb[2 * (size_t)n + 1] = {bmx[0], bmx[1], bmx[2], 0.0f};
    // byte grid: p = box minimum, smallest power of two with 255 steps covering the (upward-rounded) extent
    uint32_t eb[3]; float inv_step[3];
    for (int a = 0; a < 3; a++) {
        const float ext = sub_up(bmx[a], bmn[a]);
        int e = -120;
        if (ext > 0.0f) {
            int k; const float m = frexpf(ext, &k);            // ext = m * 2^k, m in [0.5, 1)
            e = m <= 0.99609375f ? k - 8 : k - 7;               // 255 * 2^(k-8) = 0.99609375 * 2^k
            if (e < -120) e = -120;
            if (e > 120) e = 120;                               // (cannot cover; such coordinates are rejected at commit)
        }
        eb[a] = (uint32_t)(e + 127); inv_step[a] = u2f((uint32_t)(127 - e) << 23);
    }
    N.px = bmn[0]; N.py = bmn[1]; N.pz = bmn[2];
    N.e_imask = eb[0] | eb[1] << 8 | eb[2] << 16 | imask << 24;
    for (int r = 0; r < 12; r++) N.q[r] = 0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) {
        if (!((used >> sl) & 1u)) continue;
        for (int a = 0; a < 3; a++) {
            float qlo = floorf(sub_down(cmn[sl][a], bmn[a]) * inv_step[a]), qhi = ceilf(sub_up(cmx[sl][a], bmn[a]) * inv_step[a]);
            qlo = fminf(255.0f, fmaxf(0.0f, qlo)); qhi = fminf(255.0f, fmaxf(0.0f, qhi));
            N.q[2 * a + (sl >> 2)] |= (uint32_t)qlo << (8 * (sl & 3));
            N.q[2 * (3 + a) + (sl >> 2)] |= (uint32_t)qhi << (8 * (sl & 3));
        }
    }
    nodes[n] = N;
}

// ---------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t items, uint32_t max_blocks) {
    uint32_t b = (items + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    return b < max_blocks ? b : max_blocks;
}
static inline size_t small_planes_bytes(const DevScene& sc) { return sc.nsmall ? (size_t)small_planes_count(sc.nsmall) * 16 : 0; }   // stage_lds
size_t trace_lds_bytes(const DevScene& sc) {      // the LDS column stack is always reserved: debug / pass-1 kernels use it
    return (size_t)sc.lds_nodes * 80 + (size_t)sc.lds_tris * 48 + small_planes_bytes(sc) + (size_t)sc.stack_depth * kBlock * kStackEntryBytes;
}
size_t trace_lds_bytes_queue(const DevScene& sc) {   // queue kernels with a private stack need no LDS stack
    const size_t stack = sc.stack_private == 1 ? 0 : (size_t)sc.stack_depth * kBlock * kStackEntryBytes;
    return (size_t)sc.lds_nodes * 80 + (size_t)sc.lds_tris * 48 + small_planes_bytes(sc) + stack;
}

// workgroups of the two persistent traversal kernels (default instantiations) that fit on a CU with this scene's LDS layout; 0 = query failed
int trace_workgroups_per_cu(const DevScene& sc) {
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_trace_closest<0, false, 6>, (int)kBlock, trace_lds_bytes(sc)) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_trace_shadow<0, false, 6>, (int)kBlock, trace_lds_bytes(sc)) != hipSuccess) return 0;
    return a < b ? a : b;
}
void launch_raygen(hipStream_t st, const DevFrame& f, const DevPaths& p, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount, bool compact) {
    hipLaunchKernelGGL(k_raygen, dim3(f.nblocks), dim3(kBlock), 0, st, f, p, cam, queue, qcount, compact ? 1u : 0u);
}
void launch_packet_masks(hipStream_t st, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, unsigned long long* masks) {
    const uint32_t nblk = f.npl / 64u;
    hipLaunchKernelGGL(k_packet_masks, dim3((nblk + 3u) / 4u), dim3(kBlock), 0, st, sc, f, cam, masks);
}
void launch_raygen_trace_small(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, const CameraGPU* cam, uint32_t* queue, uint32_t* qcount, uint32_t* gencount, const unsigned long long* masks) {
    hipLaunchKernelGGL(k_raygen_trace_small, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, p, cam, queue, qcount, gencount, masks);
}
void launch_trace_closest(hipStream_t st, const DevFrame& f, const DevScene& sc, const DevPaths& p, uint32_t bounce, const uint32_t* queue, const uint32_t* qcount, uint32_t* heads, uint32_t merge) {
    const float tmin = bounce == 0 ? kTMinCam : kSBias;
    if (sc.nsmall) heads = nullptr;                     // (the un-fused tiny-scene test path has no persistent waves)
    if (heads || sc.nsmall || merge < 1u) merge = 1u;
    if (merge > kMaxMerge) merge = kMaxMerge;
    const uint32_t grid = (f.nblocks + merge - 1u) / merge;
#define RTX_LAUNCH_TC(SS, TT, CC, LDSB) hipLaunchKernelGGL((k_trace_closest<SS, TT, CC>), dim3(grid), dim3(kBlock), LDSB, st, sc, sc.small, p, queue, qcount, f.qcap, tmin, sc.refill_min, sc.trace_sched, heads, f.nblocks, merge)
    if (sc.stack_private == 1) { if (heads) RTX_LAUNCH_TC(1, true, -1, trace_lds_bytes_queue(sc)); else RTX_LAUNCH_TC(1, false, -1, trace_lds_bytes_queue(sc)); }
    else if (sc.stack_ovf) {                                // RTX_OPT_STACK_CAP: the tree needs more entries than the LDS column holds (StackLdsT<true>)
        if (heads) RTX_LAUNCH_TC(2, true, -1, trace_lds_bytes(sc));
        else if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TC(2, false, 6, trace_lds_bytes(sc));
        else RTX_LAUNCH_TC(2, false, -1, trace_lds_bytes(sc));
    }
    else if (heads) RTX_LAUNCH_TC(0, true, -1, trace_lds_bytes(sc));
#ifndef RTX_NO_SCHED_SPECIAL
    else if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TC(0, false, 6, trace_lds_bytes(sc));      // the default configuration: schedule compiled in (the work counters live in the generic one)
#endif
    else RTX_LAUNCH_TC(0, false, -1, trace_lds_bytes(sc));
#undef RTX_LAUNCH_TC
}
void launch_bounce_small(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce_first, uint32_t bounce_end,
                         uint32_t* queue_a, uint32_t* queue_b, uint32_t* qrows, uint32_t* srows, const uint32_t* order, bool ring) {
    // general instantiation: 118 VGPRs, 4 waves/SIMD (5 or 6 spill and measured slower); Lambert-only: 85 VGPRs, 5 waves/SIMD (a build for 6 waves, 80 VGPRs
    // with 2 spilled, measured the same: 19.13 vs 19.03 ms).  Bounce 0 (reads the primary hits) is its own instantiation and launch.
    const bool lam = (f.flags & 1u) != 0u, have_hit = bounce_first == 0u;
#define RTX_LAUNCH_BOUNCE(HH, LL, RR) hipLaunchKernelGGL((k_bounce_small<4, HH, LL, RR>), dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, p, bounce_first, bounce_end, queue_a, queue_b, qrows, srows, order)
    if (have_hit) { if (lam) RTX_LAUNCH_BOUNCE(true, true, false); else RTX_LAUNCH_BOUNCE(true, false, false); }
    else if (ring) { if (lam) RTX_LAUNCH_BOUNCE(false, true, true); else RTX_LAUNCH_BOUNCE(false, false, true); }
    else { if (lam) RTX_LAUNCH_BOUNCE(false, true, false); else RTX_LAUNCH_BOUNCE(false, false, false); }
#undef RTX_LAUNCH_BOUNCE
}
void launch_bounce_bvh(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce_first, uint32_t bounce_end,
                       uint32_t* queue_a, uint32_t* queue_b, uint32_t* hitq, uint32_t* qrows, uint32_t* srows, const uint32_t* order) {
    if (sc.stack_private == 1) hipLaunchKernelGGL(k_bounce_bvh<1>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes_queue(sc), st, sc, f, p, bounce_first, bounce_end, queue_a, queue_b, hitq, qrows, srows, order);
    else if (sc.stack_ovf) hipLaunchKernelGGL(k_bounce_bvh<2>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, f, p, bounce_first, bounce_end, queue_a, queue_b, hitq, qrows, srows, order);
    else hipLaunchKernelGGL(k_bounce_bvh<0>, dim3(f.nblocks), dim3(kBlock), trace_lds_bytes(sc), st, sc, f, p, bounce_first, bounce_end, queue_a, queue_b, hitq, qrows, srows, order);
}
void launch_order_queues(hipStream_t st, const uint32_t* qcount, uint32_t G, uint32_t* order) {
    hipLaunchKernelGGL(k_order_queues, dim3(1), dim3(1024), 0, st, qcount, G, order);
}
void launch_trace_shadow(hipStream_t st, const DevFrame& f, const DevScene& sc, const DevPaths& p, uint32_t j, const uint32_t* shcount, uint32_t* heads, uint32_t merge) {
    const size_t seg = (size_t)j * f.qcap * f.nblocks;
    if (sc.nsmall) heads = nullptr;
    if (heads || sc.nsmall || merge < 1u) merge = 1u;
    if (merge > kMaxMerge) merge = kMaxMerge;
    const uint32_t grid = (f.nblocks + merge - 1u) / merge;
#define RTX_LAUNCH_TS(SS, TT, CC, LDSB) hipLaunchKernelGGL((k_trace_shadow<SS, TT, CC>), dim3(grid), dim3(kBlock), LDSB, st, sc, sc.small, p, p.sh_o + seg, p.sh_d + seg, p.sh_c + seg, shcount, f.qcap, sc.refill_min, sc.trace_sched, heads, f.nblocks, merge)
    if (sc.stack_private == 1) { if (heads) RTX_LAUNCH_TS(1, true, -1, trace_lds_bytes_queue(sc)); else RTX_LAUNCH_TS(1, false, -1, trace_lds_bytes_queue(sc)); }
    else if (sc.stack_ovf) {
        if (heads) RTX_LAUNCH_TS(2, true, -1, trace_lds_bytes(sc));
        else if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TS(2, false, 6, trace_lds_bytes(sc));
        else RTX_LAUNCH_TS(2, false, -1, trace_lds_bytes(sc));
    }
    else if (heads) RTX_LAUNCH_TS(0, true, -1, trace_lds_bytes(sc));
#ifndef RTX_NO_SCHED_SPECIAL
    else if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TS(0, false, 6, trace_lds_bytes(sc));
#endif
    else RTX_LAUNCH_TS(0, false, -1, trace_lds_bytes(sc));
#undef RTX_LAUNCH_TS
}
void launch_shade(hipStream_t st, const DevScene& sc, const DevFrame& f, const DevPaths& p, uint32_t bounce,
                  const uint32_t* queue, const uint32_t* qcount, uint32_t* next_queue, uint32_t* next_count, uint32_t* shcounts) {
    // material-sorted variant: measured slower (see k_shade), the permutation un-coalesces the per-path state streams
#define RTX_LAUNCH_SHADE(SS, LL) hipLaunchKernelGGL((k_shade<SS, LL>), dim3(f.nblocks), dim3(kBlock), shade_lds_bytes(sc, SS), st, sc, f, p, bounce, queue, qcount, next_queue, next_count, shcounts)
    const bool lam = (f.flags & 1u) != 0u;
    if (sc.shade_dense && !sc.sort_materials) {
        if (lam) hipLaunchKernelGGL((k_shade_dense<true>), dim3(f.nblocks), dim3(kBlock), 0, st, sc, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
        else hipLaunchKernelGGL((k_shade_dense<false>), dim3(f.nblocks), dim3(kBlock), 0, st, sc, f, p, bounce, queue, qcount, next_queue, next_count, shcounts);
    }
    else if (sc.sort_materials) { if (lam) RTX_LAUNCH_SHADE(true, true); else RTX_LAUNCH_SHADE(true, false); }
    else { if (lam) RTX_LAUNCH_SHADE(false, true); else RTX_LAUNCH_SHADE(false, false); }
#undef RTX_LAUNCH_SHADE
}
void launch_v6_pass1(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t sample_id,
                     F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata, unsigned long long* counters, const uint32_t* pixels, uint32_t npixels) {
    hipLaunchKernelGGL(k_v6_pass1, dim3(grid_for(pixels ? npixels : f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, sample_id, accum, res_di, res_gi, sdata, counters, pixels, npixels);
}
void launch_restir_pass2(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t* const bufs[6], unsigned long long* counters,
                         const uint32_t* pixels, uint32_t npixels) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_pass2, dim3(grid_for(pixels ? npixels : f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, B, counters, pixels, npixels);
}
void launch_restir_pack_state(hipStream_t st, uint32_t max_blocks, const DevFrame& f, uint32_t* const bufs[6], uint32_t* slab) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_pack_state, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), 0, st, f, B, slab);
}
void launch_restir_unpack_state(hipStream_t st, uint32_t max_blocks, const DevFrame& f, uint32_t nshards, const uint32_t* slabs, uint32_t* const bufs[6]) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_unpack_state, dim3(grid_for(f.npl * nshards, max_blocks)), dim3(kBlock), 0, st, f, nshards, slabs, B);
}
// history records of n <= 16 pixel rectangles <-> one buffer (rtx_restir_pack_halo / unpack_halo): rect k = (x0, y0, w, h), records of all rectangles back to back
void launch_restir_halo(hipStream_t st, uint32_t max_blocks, uint32_t width, bool pack, const uint32_t* rects4, uint32_t n, uint32_t* const bufs[6], uint32_t* buf) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    HaloRects R{}; R.n = n; R.first[0] = 0;
    for (uint32_t k = 0; k < n; k++) { R.x0[k] = rects4[4 * k]; R.y0[k] = rects4[4 * k + 1]; R.w[k] = rects4[4 * k + 2]; R.first[k + 1] = R.first[k] + rects4[4 * k + 2] * rects4[4 * k + 3]; }
    if (!R.first[n]) return;
    if (pack) hipLaunchKernelGGL(k_restir_halo<true>, dim3(grid_for(R.first[n], max_blocks)), dim3(kBlock), 0, st, width, R, B, buf);
    else hipLaunchKernelGGL(k_restir_halo<false>, dim3(grid_for(R.first[n], max_blocks)), dim3(kBlock), 0, st, width, R, B, buf);
}
void launch_restir_pass3(hipStream_t st, uint32_t max_blocks, const DevScene& sc, const DevFrame& f, const CameraGPU* cam, uint32_t* const bufs[6], F4* accum, unsigned long long* counters) {
    RestirBufs B = {bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], bufs[5]};
    hipLaunchKernelGGL(k_restir_pass3, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, f, cam, B, accum, counters);
}
// ---- wavefront ReSTIR stages ----
void launch_trace_occ(hipStream_t st, const DevScene& sc_in, const RsQ& q, const uint32_t* shcnt) {
    DevPaths none{};
    // the visibility rays of the ReSTIR stages run between arbitrary scene points (reconnections, last frame's samples), not towards sampled lights: the NEE probe's
    // order does not carry over — slot order measured best on all three scenes (atrium 8.28 vs 8.38 ms, garage 6.74 vs 6.84, street 7.99 vs 8.02-8.12 per frame with orders 1 / 2)
    DevScene sc = sc_in; sc.any_order = sc_in.any_order_occ;
#define RTX_LAUNCH_TO(CC) hipLaunchKernelGGL((k_trace_shadow<SL_, false, CC, 1>), dim3(q.G), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, none, q.sh_o, q.sh_d, (const F4*)nullptr, shcnt, q.rcap, sc.refill_min, sc.trace_sched, (uint32_t*)nullptr, q.G, 1u, q.sh_pay, q.occ)
    if (sc.stack_ovf) { constexpr int SL_ = 2; if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TO(6); else RTX_LAUNCH_TO(-1); }
    else { constexpr int SL_ = 0; if (sc.trace_sched == 6u && !sc.nsmall && !sc.trace_cnt) RTX_LAUNCH_TO(6); else RTX_LAUNCH_TO(-1); }
#undef RTX_LAUNCH_TO
}
static inline RestirBufs rs_bufs(uint32_t* const* b) { return RestirBufs{b[0], b[1], b[2], b[3], b[4], b[5]}; }
void launch_rs_raygen(hipStream_t st, const DevFrame& f, const RsQ& q, const CameraGPU* cam, uint32_t sample_id, uint32_t* cnt_out) {
    hipLaunchKernelGGL(k_rs_raygen, dim3(q.G), dim3(kBlock), 0, st, f, q, cam, sample_id, cnt_out);
}
void launch_rs_p1_ris(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, const uint32_t* cnt_in, uint32_t* cnt_out, F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata) {
    hipLaunchKernelGGL(k_rs_p1_ris, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cnt_in, cnt_out, accum, res_di, res_gi, sdata);
}
void launch_rs_p1_ris_finish(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, const uint32_t* cnt_in, uint32_t* cnt_out, uint32_t* shcnt, uint32_t* res_di, uint32_t* sdata) {
    hipLaunchKernelGGL(k_rs_p1_ris_finish, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cnt_in, cnt_out, shcnt, res_di, sdata);
}
void launch_rs_p1_first(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, const uint32_t* cnt_in, uint32_t* cnt_out) {
    hipLaunchKernelGGL(k_rs_p1_first, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cnt_in, cnt_out);
}
void launch_rs_p1_loop(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, uint32_t set, uint32_t iter, const uint32_t* cnt_in, uint32_t* cnt_out) {
    hipLaunchKernelGGL(k_rs_p1_loop, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, set, iter, cnt_in, cnt_out);
}
void launch_rs_p1_emit_final(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, const CameraGPU* cam, uint32_t* const* bufs, uint32_t* shcnt) {
    hipLaunchKernelGGL(k_rs_p1_emit_final, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cam, bufs ? rs_bufs(bufs) : RestirBufs{}, bufs ? 1u : 0u, shcnt);
}
void launch_rs_p1_finish(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, F4* accum, uint32_t* res_di, uint32_t* res_gi, uint32_t* sdata, const CameraGPU* cam, uint32_t* const* bufs) {
#ifdef RTX_RS_FUSED_FINISH       // (A/B build: make VARIANT=ffin VARFLAGS=-DRTX_RS_FUSED_FINISH — the round-3 form, both halves in one kernel)
    hipLaunchKernelGGL((k_rs_p1_finish<true, true>), dim3(q.G), dim3(kBlock), 0, st, sc, f, q, accum, res_di, res_gi, sdata, cam, bufs ? rs_bufs(bufs) : RestirBufs{}, bufs ? 1u : 0u);
#else
    hipLaunchKernelGGL((k_rs_p1_finish<true, false>), dim3(q.G), dim3(kBlock), 0, st, sc, f, q, accum, res_di, res_gi, sdata, cam, RestirBufs{}, 0u);
    if (bufs) hipLaunchKernelGGL((k_rs_p1_finish<false, true>), dim3(q.G), dim3(kBlock), 0, st, sc, f, q, accum, res_di, res_gi, sdata, cam, rs_bufs(bufs), 1u);
#endif
}
void launch_rs_p3_keys(hipStream_t st, const DevFrame& f, const RsQ& q, uint32_t* const bufs[6], F4* key_a, F4* key_b) {
    hipLaunchKernelGGL(k_rs_p3_keys, dim3(q.G), dim3(kBlock), 0, st, f, q, rs_bufs(bufs), RsKeys{key_a, key_b});
}
void launch_rs_p3_select(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, const CameraGPU* cam, uint32_t* const bufs[6], uint32_t* shcnt, F4* key_a, F4* key_b) {
    if (key_a) { hipLaunchKernelGGL(k_rs_p3_select_keys, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cam, rs_bufs(bufs), RsKeys{key_a, key_b}, shcnt); return; }
    hipLaunchKernelGGL(k_rs_p3_select, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, cam, rs_bufs(bufs), shcnt);
}
void launch_rs_p3_merge(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, uint32_t* const bufs[6], uint32_t* shcnt) {
    hipLaunchKernelGGL(k_rs_p3_merge<false>, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, rs_bufs(bufs), shcnt);
    hipLaunchKernelGGL(k_rs_p3_merge<true>, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, rs_bufs(bufs), shcnt);
}
void launch_rs_p3_shade(hipStream_t st, const DevScene& sc, const DevFrame& f, const RsQ& q, uint32_t* const bufs[6], F4* accum) {
    hipLaunchKernelGGL(k_rs_p3_shade, dim3(q.G), dim3(kBlock), 0, st, sc, f, q, rs_bufs(bufs), accum);
}
void launch_accumulate(hipStream_t st, uint32_t max_blocks, const DevFrame& f, const DevPaths& p, F4* accum) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), 0, st, f, p, accum);
}
void launch_srgb8(hipStream_t st, const F4* accum, uint32_t npix, uint32_t* out) {
    hipLaunchKernelGGL(k_srgb8, dim3((npix + kBlock - 1) / kBlock), dim3(kBlock), 0, st, accum, npix, out);
}
void launch_debug_layer(hipStream_t st, uint32_t max_blocks, const DevScene& sc, uint32_t width, uint32_t height, const CameraGPU* cam, uint32_t layer, uint32_t* out) {
    hipLaunchKernelGGL(k_debug_layer, dim3(grid_for(width * height, max_blocks)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, width, height, cam, layer, out);
}
void launch_pack_tiles(hipStream_t st, uint32_t max_blocks, const DevFrame& f, const F4* accum, F4* slab) {
    hipLaunchKernelGGL(k_pack_tiles, dim3(grid_for(f.npl, max_blocks)), dim3(kBlock), 0, st, f, accum, slab);
}
void launch_unpack_tiles(hipStream_t st, uint32_t max_blocks, const DevFrame& f, uint32_t nshards, const F4* slabs, F4* accum) {
    hipLaunchKernelGGL(k_unpack_tiles, dim3(grid_for(f.npl * nshards, max_blocks)), dim3(kBlock), 0, st, f, nshards, slabs, accum);
}
void launch_refit(hipStream_t st, Node8GPU* nodes, const uint32_t* level_start, uint32_t nlevels, TriGPU* tris, uint32_t ntris, const TriShade* shade,
                  const InstGPU* insts, const F4* objtris, F4* node_aabb, uint32_t* scale_bits, const uint32_t* moved, uint8_t* tri_dirty, uint8_t* node_dirty) {
    if (!moved) tri_dirty = nullptr;                              // full refit
    if (ntris) hipLaunchKernelGGL(k_refit_tris, dim3((ntris + kBlock - 1) / kBlock), dim3(kBlock), 0, st, tris, ntris, shade, insts, objtris, scale_bits, moved, tri_dirty);
    for (uint32_t l = nlevels; l-- > 0;) {                       // deepest level first: children are refitted before their parents
        const uint32_t first = level_start[l], count = level_start[l + 1] - first;
        if (count) hipLaunchKernelGGL(k_refit_nodes, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nodes, first, count, tris, node_aabb, scale_bits, (const uint8_t*)tri_dirty, node_dirty);
    }
}
void launch_dbg_trace(hipStream_t st, const DevScene& sc, const F4* rays, uint32_t n, int any, F4* hits) {
    hipLaunchKernelGGL(k_dbg_trace, dim3(grid_for(n, 2048)), dim3(kBlock), trace_lds_bytes(sc), st, sc, sc.small, rays, n, any, hits);
}
void launch_dbg_surface(hipStream_t st, const DevScene& sc, const F4* rays, const F4* hits, uint32_t n, F4* out) {
    hipLaunchKernelGGL(k_dbg_surface, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, rays, hits, n, out);
}
void launch_dbg_bsdf_eval(hipStream_t st, const DevScene& sc, uint32_t mat, uint32_t flags, const float* in9, uint32_t n, float* out8) {
    hipLaunchKernelGGL(k_dbg_bsdf_eval, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, mat, flags, in9, n, out8);
}
void launch_dbg_bsdf_sample(hipStream_t st, const DevScene& sc, uint32_t mat, uint32_t flags, const float* in8, uint32_t n, float* out8) {
    hipLaunchKernelGGL(k_dbg_bsdf_sample, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, mat, flags, in8, n, out8);
}
void launch_dbg_tea(hipStream_t st, uint32_t s0, uint32_t s1, uint32_t n, float* out, uint32_t* seed_out) {
    hipLaunchKernelGGL(k_dbg_tea, dim3(1), dim3(64), 0, st, s0, s1, n, out, seed_out);
}
void launch_dbg_primary(hipStream_t st, const DevFrame& f, const CameraGPU* cam, uint32_t sample_id, F4* rays) {
    hipLaunchKernelGGL(k_dbg_primary, dim3((f.width * f.height + kBlock - 1) / kBlock), dim3(kBlock), 0, st, f, cam, sample_id, rays);
}

}  // namespace rtx

#ifdef RTX_PROFILE_SECTIONS
// tooling entry point of the PROFILE=1 build only (tools/section_profile.py): read (and optionally clear) the section counters
extern "C" int rtx_debug_traversal(unsigned long long* out8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(rtx::g_trv), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_trv), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
extern "C" int rtx_debug_sections(unsigned long long* out36, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out36 && hipMemcpyFromSymbol(out36, HIP_SYMBOL(rtx::g_sec), sizeof(unsigned long long) * 36) != hipSuccess) return -1;
    if (reset) { unsigned long long z[36] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_sec), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

#ifdef RTX_WAVE_CLOCK
extern "C" int rtx_debug_wave_times(unsigned long long* out, unsigned nwaves, int reset) {      // nwaves <= 65536 (start, end) pairs
    if (hipDeviceSynchronize() != hipSuccess || nwaves > 65536u) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_wgt), sizeof(unsigned long long) * 2 * nwaves) != hipSuccess) return -1;
    if (reset) { void* d = nullptr; if (hipGetSymbolAddress(&d, HIP_SYMBOL(rtx::g_wgt)) != hipSuccess || hipMemset(d, 0, sizeof(unsigned long long) * 2 * 65536) != hipSuccess) return -1; }
    return 0;
}
#endif
