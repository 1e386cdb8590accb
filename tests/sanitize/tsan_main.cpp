// ThreadSanitizer exercise of the host layer's threaded code (CPU only; built and run by tests/test_host_layer.py::test_parallel_builder_under_tsan): the parallel top-down
// phase of build_bvh (subtree tasks on a thread pool, spliced back in cutting order), the re-insertion passes and the wide collapse behind it, the commit-time any-hit probe —
// with 1, 3 and 16 builder threads; the three trees must be the same tree (the replayed traversal takes the same steps ray for ray).
#include "rtx_host.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
int main() {
    const uint32_t nt = 66000;                                   // >= 65 536: the parallel phase is on
    std::mt19937 rng(11); std::uniform_real_distribution<float> U(-1, 1); std::normal_distribution<float> N(0, 0.03f);
    std::vector<float> w((size_t)nt * 9);
    for (uint32_t i = 0; i < nt; i++) { const float cx = 4 * U(rng), cy = U(rng), cz = 4 * U(rng); for (int k = 0; k < 3; k++) { w[i * 9 + k * 3] = cx + N(rng); w[i * 9 + k * 3 + 1] = cy + N(rng); w[i * 9 + k * 3 + 2] = cz + N(rng); } }
    const uint32_t nr = 2000;
    std::vector<float> rays((size_t)nr * 8);
    for (uint32_t i = 0; i < nr; i++) { float d[3] = {U(rng), U(rng), U(rng)}; const float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-9f;
        float* r = &rays[(size_t)i * 8]; r[0] = 4 * U(rng); r[1] = 0.3f * U(rng); r[2] = 4 * U(rng); r[3] = 1e-5f; r[4] = d[0] / l; r[5] = d[1] / l; r[6] = d[2] / l; r[7] = 1e30f; }
    std::vector<float> first;
    for (int threads : {1, 3, 16}) {
        if (rtxh_bvh_option("threads", threads) != 0) { printf("threads option refused\n"); return 2; }
        uint32_t nodes8 = 0, stack = 0;
        const int rc = threads == 3 ? rtxh_bvh8_check(w.data(), nt, &nodes8, &stack) : 0;      // (the validators once: they are serial and slow under the sanitizer)
        std::vector<float> out((size_t)nr * 4); uint32_t refs = 0;
        const int rr = rtxh_bvh_replay(w.data(), nt, rays.data(), nr, 0, 0, out.data(), &refs);
        printf("threads %d: bvh8 rc %d nodes %u stack %u | replay rc %d refs %u\n", threads, rc, nodes8, stack, rr, refs);
        if (rc || rr || refs != nt) return 3;
        if (first.empty()) first = out; else if (memcmp(first.data(), out.data(), out.size() * 4) != 0) { printf("the tree depends on the thread count\n"); return 4; }
    }
    // a whole scene build (flatten, lights, build, collapse, any-hit probe) of a scene large enough for the parallel phase
    rtxh_scene* s = rtxh_scene_sponza_class(66000, 260);
    if (!s || rtxh_scene_save(s, "/tmp/tsan_scene.rtxscn") != 0) { printf("scene build failed: %s\n", rtxh_last_error()); return 5; }
    rtxh_scene_free(s);
    printf("done\n");
    return 0;
}
