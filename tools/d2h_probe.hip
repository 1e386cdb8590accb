// d2h_probe — does hipMemcpyAsync(device -> PAGEABLE host) + hipStreamSynchronize always deliver?  (round 5: the once-in-thousands "mismatch" of the fuzz tests turned out
// to sit in the READ-BACKS — a validator fed a partly stale copy of the tree, a tree hash that differed while the tree rendered the right image.)
// A kernel writes a per-iteration pattern into a device buffer; the buffer is copied into a freshly allocated pageable std::vector on a non-blocking stream, the stream is
// synchronised, every word is checked.  Sizes like the library's read-backs (trees of mid-size scenes: 20-300 KB; a 48 x 32 image: 24 KB), fresh contexts' worth of
// allocation churn in between.  Prints the number of incomplete copies per method: async + stream sync into pageable memory, into pinned memory, blocking hipMemcpy.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/d2h_probe tools/d2h_probe.hip      run: tools/d2h_probe [iterations = 200000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k_fill(uint32_t* p, uint32_t n, uint32_t tag) { const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = tag ^ (i * 2654435761u); }
int main(int argc, char** argv) {
    const long iters = argc > 1 ? atol(argv[1]) : 200000;
    hipStream_t st; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { printf("no stream\n"); return 1; }
    const uint32_t sizes[] = {6144, 20000, 65536, 131072, 300000, 1u << 20};       // bytes / 4 = words below
    uint32_t* dev = nullptr; if (hipMalloc(&dev, 4u << 20) != hipSuccess) return 1;
    uint32_t* pinned = nullptr; if (hipHostMalloc((void**)&pinned, 4u << 20, hipHostMallocDefault) != hipSuccess) return 1;
    long bad[3] = {0, 0, 0}, first_bad[3] = {-1, -1, -1};
    for (long it = 0; it < iters; it++) {
        const uint32_t bytes = sizes[it % 6], n = bytes / 4, tag = (uint32_t)it * 747796405u + 1u;
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, st, dev, n, tag);
        for (int method = 0; method < 3; method++) {
            std::vector<uint32_t> host;                       // a fresh pageable buffer every time, as the library's callers have
            uint32_t* dst = pinned;
            if (method != 1) { host.assign(n, 0u); dst = host.data(); } else memset(pinned, 0, bytes);
            if (method == 2) { (void)hipStreamSynchronize(st); (void)hipMemcpy(dst, dev, bytes, hipMemcpyDeviceToHost); }
            else { (void)hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, st); (void)hipStreamSynchronize(st); }
            uint32_t wrong = 0, at = 0;
            for (uint32_t i = 0; i < n; i++) if (dst[i] != (tag ^ (i * 2654435761u))) { if (!wrong) at = i; wrong++; }
            if (wrong) { bad[method]++; if (first_bad[method] < 0) { first_bad[method] = it; printf("method %d iteration %ld: %u of %u words wrong, first at word %u (value %08x)\n", method, it, wrong, n, at, dst[at]); } }
        }
        if (it % 50000 == 49999) { printf("  %ld iterations: incomplete copies async->pageable %ld, async->pinned %ld, blocking hipMemcpy %ld\n", it + 1, bad[0], bad[1], bad[2]); fflush(stdout); }
    }
    printf("d2h_probe: %ld iterations: incomplete copies async->pageable %ld, async->pinned %ld, blocking hipMemcpy %ld\n", iters, bad[0], bad[1], bad[2]);
    return 0;
}
