// exec_half.hip — does a gfx950 SIMD skip the half (or quarter) of a wave64 VALU instruction whose EXEC bits are all zero?  (tooling only)
// The traversal kernels run their triangle steps with ~24 of 64 lanes: if a pass over 32 idle lanes were free, compacting the busy lanes into one half
// of the wave would halve those steps.  Each wave runs a long stream of independent v_fma_f32 / v_cvt+v_max3 mix under an EXEC mask chosen per run:
//   all 64 | low 32 | low 16 | even lanes (32, scattered) | lanes 0-15 + 32-47
// and the time per instruction is compared.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/exec_half.hip -o gpurun_out/exec_half && gpurun_out/exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_run(float* out, unsigned long long mask, int iters) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float b = 1.0000001f, c = 1e-9f;
    unsigned u = threadIdx.x * 2654435761u;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iters; it++) {
            if (KIND == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[k & 15]) : "v"(u));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(k + 1) & 15]) : "v"(b), "v"(c));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[(k + 2) & 15]) : "v"(b), "v"(c));
                    asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[(k + 3) & 15]) : "v"(b));
                }
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; const int blocks = 256 * 8, iters = 4000;           // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const struct { const char* name; unsigned long long m; } masks[] = {
        {"all 64 lanes", ~0ull}, {"low 32", 0xFFFFFFFFull}, {"high 32", 0xFFFFFFFF00000000ull}, {"low 16", 0xFFFFull}, {"even lanes (32 scattered)", 0x5555555555555555ull},
        {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull}, {"one lane", 1ull}};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int kind = 0; kind < 2; kind++) {
        double base = 0;
        for (auto& mk : masks) {
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_run<0>, dim3(blocks), dim3(256), 0, 0, out, mk.m, iters);
                else hipLaunchKernelGGL(k_run<1>, dim3(blocks), dim3(256), 0, 0, out, mk.m, iters);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) { if (base == 0) base = ms; printf("kind %d (%s)  %-28s %8.3f ms  %.2fx of all-64\n", kind, kind ? "cvt+fma+max3+min" : "v_fma_f32", mk.name, ms, ms / base); }
            }
        }
    }
    return 0;
}
