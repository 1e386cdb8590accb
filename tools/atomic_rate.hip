// Tooling: rate of device-scope global atomics on gfx950 as the work-stealing cursors would use them.
// Every wave's lane 0 does `iters` atomicAdd (with return) on an address chosen by the mode; prints ns per atomic (aggregate) and
// the latency one wave sees.   hipcc --offload-arch=gfx950 -O3 tools/atomic_rate.hip -o /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(uint32_t* a, uint32_t stride_words, uint32_t naddr, uint32_t iters, unsigned long long* cyc, uint32_t* sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        uint32_t* p = a + (size_t)(wave % naddr) * stride_words;
        for (uint32_t i = 0; i < iters; i++) acc += atomicAdd(p, 1u + (acc & 1u));        // dependent chain: one atomic in flight per wave
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { atomicAdd(cyc, t1 - t0); sink[wave] = acc; }
}
int main() {
    uint32_t* a; unsigned long long* cyc; uint32_t* sink;
    const size_t words = 1u << 22;
    hipMalloc(&a, words * 4); hipMalloc(&cyc, 8); hipMalloc(&sink, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Mode { const char* name; uint32_t blocks, naddr, stride; } modes[] = {
        {"1 wave, 1 address", 1, 1, 1},
        {"1280 blocks x 4 waves, ONE address", 1280, 1, 1},
        {"1280 x 4 waves, 4096 adjacent words (16 KB)", 1280, 4096, 1},
        {"1280 x 4 waves, 4096 addresses 128 B apart", 1280, 4096, 32},
        {"1280 x 4 waves, own address per wave, adjacent", 1280, 5120, 1},
        {"1280 x 4 waves, own address per wave, 128 B apart", 1280, 5120, 32},
        {"1280 x 4 waves, 128 addresses 128 B apart", 1280, 128, 32},
        {"1280 x 4 waves, 8 addresses 128 B apart", 1280, 8, 32},
    };
    for (const Mode& m : modes) {
        const uint32_t iters = m.blocks == 1 ? 2000 : (m.naddr == 1 ? 20 : 200);
        hipMemset(a, 0, words * 4); hipMemset(cyc, 0, 8);
        k<<<m.blocks, 256>>>(a, m.stride, m.naddr, 2, cyc, sink);       // warm
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0); k<<<m.blocks, 256>>>(a, m.stride, m.naddr, iters, cyc, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double nw = m.blocks * 4.0, total = nw * iters;
        printf("%-52s %8.3f ms  %8.1f ns/atomic aggregate  %8.0f shader-clock ticks latency per atomic (one wave)\n", m.name, ms, ms * 1e6 / total, (double)c / total);
    }
    return 0;
}
