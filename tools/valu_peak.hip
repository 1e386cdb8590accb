// valu_peak.hip — what is the wave64 VALU issue peak of one gfx950 SIMD?  (tooling only; not part of the product library)
//
// MI355X_MICROARCH.md says the CDNA4 SIMD is 32 lanes wide: a wave64 v_fma_f32 occupies it for 2 cycles, while ONE wave's own stream
// issues every 4.  Round 1 priced the VALU roofline at one instruction per 4 cycles per SIMD.  This microbenchmark measures the rate
// directly: W waves per SIMD (W = 1, 2, 3, 4, 5, 8) each run a long stream of independent VALU instructions; reported are
// wave-instructions per cycle per SIMD from s_memtime inside the kernel (median over waves) and from the hipEvent wall time at the
// clock GRBM would report (we print the in-kernel clock as well: s_memtime / s_memrealtime).
//   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f2v __attribute__((ext_vector_type(2)));

// KIND: 0 = v_fma_f32 independent (16 accumulators), 1 = v_pk_fma_f32 independent, 2 = v_fma_f32 one dependent chain,
//       3 = v_rcp_f32 independent, 4 = mix typical of the traversal (cvt_f32_ubyte + fma + max3 + cmp), 5 = v_add_u32 (integer, TEA-like)
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    float b = 1.0000001f, c = 1e-9f;
    unsigned u = threadIdx.x * 2654435761u;
    f2v p[8]; for (int i = 0; i < 8; i++) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
    f2v pb = {b, b}, pc = {c, c};
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 4) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[k & 15]) : "v"(u));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(k + 1) & 15]) : "v"(b), "v"(c));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 2) & 15]) : "v"(b), "v"(c));
                asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[(k + 3) & 15]), "v"(b) : "vcc");
            }
        } else if (KIND == 5) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(threadIdx.x));
        } else {
            // 1:1 mixes of two opcodes on independent registers (calibration of the compute roofline: do the costs of two instruction classes ADD, or do they
            // overlap in separate pipes?): 6 fma + min, 7 fma + add_u32, 8 min + cvt_ubyte, 9 fma + rcp, 10 add_u32 + min, 11 fma + mul (VOP2), 12 fma + mov, 13 add_u32 + cvt
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float& x = a[i]; float& y = a[8 + i];
                    if (KIND == 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(y) : "v"(b)); }
                    if (KIND == 7) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(u)); }
                    if (KIND == 8) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b)); asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(y) : "v"(u)); }
                    if (KIND == 9) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c)); asm volatile("v_rcp_f32 %0, %0" : "+v"(y)); }
                    if (KIND == 10) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(u)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(y) : "v"(b)); }
                    if (KIND == 11) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y) : "v"(b)); }
                    if (KIND == 12) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c)); asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(b)); }
                    if (KIND == 13) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(u)); asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(y) : "v"(u)); }
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int i = 0; i < 16; i++) s += a[i];
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)u;
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; rt[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0; }
}

// per-instruction issue cost: 16 independent destination registers x 4, one opcode per kernel
#define OPK(ID, ASM) template <> __global__ __launch_bounds__(256) void k_op<ID>(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) { \
    float a[16]; for (int i = 0; i < 16; i++) a[i] = 1.0f + (float)(threadIdx.x + i) * 1e-3f; \
    float b = 1.0000001f, c = 1.5f; unsigned u = threadIdx.x * 2654435761u + 12345u; (void)u; \
    __syncthreads(); \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime(); \
    for (int it = 0; it < iters; it++) { \
        _Pragma("unroll") for (int k = 0; k < 4; k++) _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(u) : "vcc", "s20", "s21", "s22", "s23"); } \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    float s = 0.0f; for (int i = 0; i < 16; i++) s += a[i]; out[blockIdx.x * 256 + threadIdx.x] = s; \
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; rt[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0; } }
template <int ID> __global__ void k_op(float* out, unsigned long long* cyc, unsigned long long* rt, int iters);
OPK(0, "v_mul_f32 %0, %0, %1")
OPK(1, "v_add_f32 %0, %0, %2")
OPK(2, "v_max3_f32 %0, %0, %1, %2")
OPK(3, "v_min_f32 %0, %0, %1")
OPK(4, "v_cvt_f32_ubyte0 %0, %3")
OPK(5, "v_cmp_le_f32 vcc, %0, %1")
OPK(6, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(7, "v_sqrt_f32 %0, %0")
OPK(8, "v_rsq_f32 %0, %0")
OPK(9, "v_mul_lo_u32 %0, %0, %3")
OPK(10, "v_lshl_add_u32 %0, %0, 4, %3")
OPK(11, "v_xor_b32 %0, %0, %3")
OPK(22, "v_add3_u32 %0, %0, %3, %3")
OPK(23, "v_xad_u32 %0, %0, %3, %3")
OPK(12, "v_add_u32 %0, %0, %3")
OPK(13, "v_div_scale_f32 %0, vcc, %0, %1, %0")
OPK(14, "v_div_fmas_f32 %0, %0, %1, %2")
OPK(15, "v_div_fixup_f32 %0, %0, %1, %2")
OPK(16, "v_fma_f32 %0, %0, %1, %2")
OPK(17, "v_mov_b32 %0, %1")
OPK(18, "v_and_b32 %0, %0, %3")
OPK(19, "v_lshrrev_b32 %0, 5, %0")
OPK(20, "v_cvt_f32_u32 %0, %0")
OPK(21, "v_mad_u32_u24 %0, %0, %3, %3")
OPK(24, "v_mul_f32_e64 %0, %0, %1")
OPK(25, "v_add_f32_e64 %0, %0, %2")
OPK(26, "v_fmac_f32 %0, %1, %2")
OPK(27, "v_fma_f32 %0, %0, %1, 0")
OPK(28, "v_fma_f32 %0, %0, 1.0, %2")
OPK(29, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
OPK(30, "v_max_f32 %0, %0, %1")
OPK(31, "v_med3_f32 %0, %0, %1, %2")
OPK(32, "v_perm_b32 %0, %0, %3, %3")
OPK(33, "v_pk_fma_f16 %0, %0, %1, %2")
OPK(34, "v_pk_max_f16 %0, %0, %1")
OPK(35, "v_pk_add_f16 %0, %0, %1")
OPK(36, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
OPK(37, "v_bfe_u32 %0, %0, 3, 8")
OPK(38, "v_sub_f32 %0, %0, %1")
OPK(39, "v_and_or_b32 %0, %0, %3, %3")
OPK(40, "v_or_b32 %0, %0, %3")
OPK(41, "v_cvt_f32_ubyte1 %0, %3")
OPK(42, "v_cmp_le_f32_e64 s[22:23], %0, %1")
OPK(43, "v_lshlrev_b32 %0, 3, %0")
OPK(44, "v_mul_legacy_f32 %0, %0, %1")
OPK(45, "v_ldexp_f32 %0, %0, %3")
OPK(46, "v_dot2c_f32_f16 %0, %1, %2")
OPK(47, "v_add_u32_e64 %0, %0, %3")
OPK(48, "v_xor_b32_e64 %0, %0, %3")
OPK(49, "v_and_b32_e64 %0, %0, %3")
OPK(50, "v_lshrrev_b32_e64 %0, 5, %0")
OPK(51, "v_max_f32_e64 %0, %0, %1")
OPK(52, "v_sub_f32_e64 %0, %0, %1")
OPK(53, "v_mov_b32_e64 %0, %1")
OPK(54, "v_or_b32_e64 %0, %0, %3")
OPK(55, "v_fma_mix_f32 %0, %3, %1, %0 op_sel_hi:[1,0,0]")
OPK(56, "v_fma_mix_f32 %0, %3, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
OPK(57, "v_cvt_f32_f16 %0, %3")
OPK(60, "v_fma_mix_f32 %0, %0, %1, %2")
OPK(61, "v_cvt_f32_ubyte3 %0, %3")
OPK(62, "v_mul_f32_sdwa %0, %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")

template <int ID>
static int run_op(const char* name, int ncu, float* d_out, unsigned long long* d_cyc, unsigned long long* d_rt) {
    const int iters = 10000;
    for (int W : {1, 4, 8}) {
        const int blocks = ncu * W;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_op<ID>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, 200);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_op<ID>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, iters);
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(blocks * 4), r(blocks * 4);
        CHK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r.data(), d_rt, r.size() * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
        const double cyc = (double)c[c.size() / 2], rtk = (double)r[r.size() / 2], inst = (double)iters * 64, ghz = cyc / (rtk * 10.0);
        const double rate = (double)blocks * 4 * inst / (ms * 1e-3);                    // wave-instructions per second, whole chip
        printf("op %-36s W=%d  one-wave cycles/inst=%.2f  clock=%.3f GHz  chip rate=%.1f Ginst/s  => SIMD cycles/inst=%.2f\n", name, W, cyc / inst, ghz, rate / 1e9,
               (ncu * 4.0) * ghz * 1e9 / rate);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return 0;
}

template <int KIND>
static int run(const char* name, int ncu, int inst_per_iter, float* d_out, unsigned long long* d_cyc, unsigned long long* d_rt) {
    const int iters = 20000;
    const int Ws[] = {1, 2, 3, 4, 5, 8};
    for (int W : Ws) {
        const int blocks = ncu * W;                 // 256-thread workgroups: one wave per SIMD each, W workgroups per CU
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, 200);   // warm-up
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, iters);
        CHK(hipEventRecord(e1, 0));
        CHK(hipDeviceSynchronize());
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(blocks * 4), r(blocks * 4);
        CHK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(r.data(), d_rt, r.size() * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
        const double cyc = (double)c[c.size() / 2], rtk = (double)r[r.size() / 2];
        const double inst = (double)iters * inst_per_iter;
        const double ghz = cyc / (rtk * 10.0);                       // s_memrealtime ticks at 100 MHz
        printf("%-28s W=%d  cycles/inst/wave=%.3f  wave-inst/cycle/SIMD=%.3f  in-kernel clock=%.3f GHz  wall=%.3f ms  wall-rate=%.1f Ginst/s/chip\n",
               name, W, cyc / inst, W * inst / cyc, ghz, ms, (double)blocks * 4 * inst / (ms * 1e-3) / 1e9);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return 0;
}

// Instruction-mix replay (tools/gen_mix.py writes mix_kernels.inc from a kernel's measured class counts): 128 instructions per block, independent registers
template <int ID> __device__ __forceinline__ void mix_block(float* a, f2v* p, unsigned* w, float b, float c, f2v pb, f2v pc, unsigned u);
#ifdef RTX_HAVE_MIX
#include "mix_kernels.inc"
#endif
template <int ID>
__global__ __launch_bounds__(256) void k_mix(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
    float a[16]; for (int i = 0; i < 16; i++) a[i] = 1.0f + (float)(threadIdx.x + i) * 1e-3f;
    f2v p[8]; for (int i = 0; i < 8; i++) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
    unsigned w[8]; for (int i = 0; i < 8; i++) w[i] = threadIdx.x * 2654435761u + i;
    float b = 1.0000001f, c = 1e-9f; f2v pb = {b, b}, pc = {c, c}; unsigned u = threadIdx.x * 40503u + 7u;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) mix_block<ID>(a, p, w, b, c, pb, pc, u);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f; for (int i = 0; i < 16; i++) s += a[i]; for (int i = 0; i < 8; i++) s += p[i].x + p[i].y + (float)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; rt[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0; }
}

// CALIBRATION of the compute roofline (`valu_peak calib`, tools/valu_calib.sh): ONE launch per kernel, 8 waves per SIMD, so that a `rocprofv3 --pmc` pass of this
// program has exactly one dispatch row per kernel name.  Each line prints what the kernel measured about itself (SIMD cycles per instruction from the in-kernel
// clock); the counter passes then tell how the SQ_INSTS_VALU_* counters classify the opcode and what  sum(n_class x cycles_class) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
// reads for a loop that is known to saturate the pipe — it must be 1.00 +- 0.03 for the formula to be usable as a roofline fraction.
template <class K>
static int calib_one(const char* name, K kernel, int ncu, int iters, int inst_per_iter, float* d_out, unsigned long long* d_cyc, unsigned long long* d_rt) {
    const int W = 8, blocks = ncu * W;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_rt, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(blocks * 4), r(blocks * 4);
    CHK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(r.data(), d_rt, r.size() * 8, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    const double cyc = (double)c[c.size() / 2], rtk = (double)r[r.size() / 2], inst = (double)iters * inst_per_iter, ghz = cyc / (rtk * 10.0);
    printf("calib %-28s wave-inst=%.0f  wave-cycles=%.0f  clock=%.3f GHz  wall=%.3f ms  SIMD-cycles/inst=%.3f\n", name, (double)blocks * 4 * inst, cyc, ghz, ms, (ncu * 4.0) * ghz * 1e6 * ms / ((double)blocks * 4 * inst));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

int main(int argc, char** argv) {
    const bool calib = argc > 1 && !strcmp(argv[1], "calib");
    const int first_op = (argc > 1 && !calib) ? atoi(argv[1]) : 0;          // run only the per-opcode kernels with id >= first_op
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    float* d_out; unsigned long long *d_cyc, *d_rt;
    CHK(hipMalloc(&d_out, (size_t)ncu * 8 * 256 * 4)); CHK(hipMalloc(&d_cyc, (size_t)ncu * 8 * 4 * 8)); CHK(hipMalloc(&d_rt, (size_t)ncu * 8 * 4 * 8));
    if (calib) {
        const int it = 20000;
        if (calib_one("k_valu<0> v_fma_f32", k_valu<0>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<1> v_pk_fma_f32", k_valu<1>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<3> v_rcp_f32", k_valu<3>, ncu, it / 2, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<4> cvt/fma/max3/cmp", k_valu<4>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<6> fma+min", k_valu<6>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<7> fma+add_u32", k_valu<7>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<8> min+cvt", k_valu<8>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<9> fma+rcp", k_valu<9>, ncu, it / 2, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<10> add_u32+min", k_valu<10>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<11> fma+mul", k_valu<11>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<12> fma+mov", k_valu<12>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_valu<13> add_u32+cvt", k_valu<13>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<0> v_mul_f32", k_op<0>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<1> v_add_f32", k_op<1>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<3> v_min_f32", k_op<3>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<4> v_cvt_f32_ubyte0", k_op<4>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<12> v_add_u32", k_op<12>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<29> v_cndmask_b32", k_op<29>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<9> v_mul_lo_u32", k_op<9>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<37> v_bfe_u32", k_op<37>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
        if (calib_one("k_op<17> v_mov_b32", k_op<17>, ncu, it, 64, d_out, d_cyc, d_rt)) return 1;
#ifdef RTX_HAVE_MIX
#define MIX(ID, NAME) if (calib_one("k_mix<" #ID "> " NAME, k_mix<ID>, ncu, it / 2, 128, d_out, d_cyc, d_rt)) return 1;
        RTX_MIX_LIST
#undef MIX
#endif
        return 0;
    }
    if (run<0>("v_fma_f32 independent", ncu, 64, d_out, d_cyc, d_rt)) return 1;
    if (run<1>("v_pk_fma_f32 independent", ncu, 64, d_out, d_cyc, d_rt)) return 1;
    if (run<2>("v_fma_f32 dependent chain", ncu, 64, d_out, d_cyc, d_rt)) return 1;
    if (run<3>("v_rcp_f32 independent", ncu, 64, d_out, d_cyc, d_rt)) return 1;
    if (run<4>("cvt/fma/max3/cmp mix", ncu, 64, d_out, d_cyc, d_rt)) return 1;
    if (run<5>("v_add_u32 dependent chain", ncu, 64, d_out, d_cyc, d_rt)) return 1;
#define RUNOP(ID, NAME) if (ID >= first_op && run_op<ID>(NAME, ncu, d_out, d_cyc, d_rt)) return 1;
    RUNOP(16, "v_fma_f32") RUNOP(0, "v_mul_f32") RUNOP(1, "v_add_f32") RUNOP(2, "v_max3_f32") RUNOP(3, "v_min_f32") RUNOP(4, "v_cvt_f32_ubyte0")
    RUNOP(5, "v_cmp_le_f32 vcc") RUNOP(6, "v_cndmask_b32") RUNOP(7, "v_sqrt_f32") RUNOP(8, "v_rsq_f32") RUNOP(9, "v_mul_lo_u32") RUNOP(10, "v_lshl_add_u32")
    RUNOP(11, "v_xor_b32") RUNOP(22, "v_add3_u32") RUNOP(23, "v_xad_u32") RUNOP(12, "v_add_u32") RUNOP(13, "v_div_scale_f32") RUNOP(14, "v_div_fmas_f32") RUNOP(15, "v_div_fixup_f32") RUNOP(17, "v_mov_b32")
    RUNOP(18, "v_and_b32") RUNOP(19, "v_lshrrev_b32") RUNOP(20, "v_cvt_f32_u32") RUNOP(21, "v_mad_u32_u24")
    RUNOP(24, "v_mul_f32_e64") RUNOP(25, "v_add_f32_e64") RUNOP(26, "v_fmac_f32") RUNOP(27, "v_fma_f32 a,b,0") RUNOP(28, "v_fma_f32 a,1.0,c") RUNOP(29, "v_cndmask_b32_e64 sgpr-cond")
    RUNOP(30, "v_max_f32") RUNOP(31, "v_med3_f32") RUNOP(32, "v_perm_b32") RUNOP(33, "v_pk_fma_f16") RUNOP(34, "v_pk_max_f16") RUNOP(35, "v_pk_add_f16") RUNOP(36, "v_cvt_pkrtz_f16_f32")
    RUNOP(37, "v_bfe_u32") RUNOP(38, "v_sub_f32") RUNOP(39, "v_and_or_b32") RUNOP(40, "v_or_b32") RUNOP(41, "v_cvt_f32_ubyte1") RUNOP(42, "v_cmp_le_f32_e64 sgpr") RUNOP(43, "v_lshlrev_b32")
    RUNOP(44, "v_mul_legacy_f32") RUNOP(45, "v_ldexp_f32") RUNOP(46, "v_dot2c_f32_f16")
    RUNOP(47, "v_add_u32_e64") RUNOP(48, "v_xor_b32_e64") RUNOP(49, "v_and_b32_e64") RUNOP(50, "v_lshrrev_b32_e64") RUNOP(51, "v_max_f32_e64") RUNOP(52, "v_sub_f32_e64") RUNOP(53, "v_mov_b32_e64")
    RUNOP(54, "v_or_b32_e64")
    RUNOP(55, "v_fma_mix_f32 f16lo,f32,f32") RUNOP(56, "v_fma_mix_f32 f16hi,f32,f32") RUNOP(57, "v_cvt_f32_f16") RUNOP(60, "v_fma_mix_f32 all f32") RUNOP(61, "v_cvt_f32_ubyte3") RUNOP(62, "v_mul_f32_sdwa byte1")
    return 0;
}
