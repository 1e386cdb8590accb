// Host-layer exercise for AddressSanitizer / UBSan on the CPU (GPU sanitizers are not available on the pool): scene
// generators, OBJ loader, BVH build / collapse / refit / validators, tiny-scene records, image writers.  The device API is
// stubbed (device_stubs.cpp); built and run by tests/test_host_layer.py::test_host_layer_under_asan_ubsan.
#include "rtx_host.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <random>
#include <string>
int main(int argc, char** argv) {
    // scenes
    rtxh_scene* c = rtxh_scene_cornell();
    rtxh_scene* s = rtxh_scene_sponza_class(262144, 260);
    rtxh_scene* b = rtxh_scene_bistro_class(300000, 3800);
    std::string gd = argc > 1 ? argv[1] : "tests/golden", f0 = gd + "/garage.obj", f1 = gd + "/monke.obj", md = gd + "/";
    const char* files[2] = {f0.c_str(), f1.c_str()};
    rtxh_scene* g = rtxh_scene_from_obj(files, 2, md.c_str());
    if (!c || !s || !b || !g) { printf("scene creation failed\n"); return 1; }
    std::vector<float> recs(64 * 20); std::vector<int32_t> ids(128); uint32_t n = 0, nocc = 0; float d, cm;
    int rc = rtxh_scene_small_records(c, recs.data(), ids.data(), 64, &n, &d, &cm);
    printf("small records rc=%d n=%u\n", rc, n);
    rc = rtxh_scene_small_occluders(c, &nocc);
    printf("occluders rc=%d nocc=%u\n", rc, nocc);
    if (n != 17 || nocc != 11) return 3;
    // BVH checks on soups
    std::mt19937 rng(7); std::uniform_real_distribution<float> U(-1, 1); std::normal_distribution<float> N(0, 0.03f);
    for (uint32_t nt : {0u, 1u, 2u, 3u, 7u, 65u, 1000u, 50000u}) {
        std::vector<float> w((size_t)nt * 9);
        for (uint32_t i = 0; i < nt; i++) { float cx = U(rng), cy = U(rng), cz = U(rng); for (int k = 0; k < 3; k++) { w[i * 9 + k * 3] = cx + N(rng); w[i * 9 + k * 3 + 1] = cy + N(rng); w[i * 9 + k * 3 + 2] = cz + N(rng); } }
        uint32_t nodes = 0, depth = 0, leaf = 0, n8 = 0, st = 0, hist[6];
        int r1 = rtxh_bvh_check(w.data(), nt, &nodes, &depth, &leaf), r2 = rtxh_bvh8_check(w.data(), nt, &n8, &st), r3 = rtxh_bvh8_stats(w.data(), nt, hist, &n8);
        std::vector<float> w2 = w; for (float& x : w2) x = x * 1.3f + 0.1f;
        int r4 = nt ? rtxh_bvh_refit_check(w.data(), w2.data(), nt) : 0;
        printf("nt=%u bvh2 rc=%d nodes=%u depth=%u leaf=%u | bvh8 rc=%d nodes=%u stack=%u | stats rc=%d | refit rc=%d\n", nt, r1, nodes, depth, leaf, r2, n8, st, r3, r4);
        if (r1 || r2 || r3 || r4) return 2;
    }
    // (round 5) the builder's other modes on a soup that exercises them: the GPU build's host twin (PLOC rounds + the SAH top over clusters + expansion), spatial splits with
    // re-insertion, leaves of one reference / of up to four, and the hard stand-in scenes' generators
    {
        const uint32_t nt = 40000u;
        std::vector<float> w((size_t)nt * 9);
        for (uint32_t i = 0; i < nt; i++) { float cx = U(rng), cy = U(rng), cz = U(rng), sz = (i % 97u == 0u) ? 0.6f : 1.0f; for (int k = 0; k < 3; k++) { w[i * 9 + k * 3] = cx + sz * N(rng) * (i % 97u == 0u ? 20.0f : 1.0f); w[i * 9 + k * 3 + 1] = cy + N(rng); w[i * 9 + k * 3 + 2] = cz + N(rng); } }
        struct Mode { const char* key; double on, off; };
        const Mode modes[] = {{"ploc", 16.0, 0.0}, {"split", 1e-5, 0.0}, {"leaf_stop", 4.0, 1.0}, {"sweep", 64.0, 0.0}, {"slot_assign", 1.0, 0.0}};
        for (const Mode& m : modes) {
            if (rtxh_bvh_option(m.key, m.on) != 0) { printf("unknown builder option %s\n", m.key); return 6; }
            if (std::string(m.key) == "ploc") rtxh_bvh_option("ploc_top", 2048.0);            // several PLOC rounds on 40 000 triangles
            uint32_t n8 = 0, st = 0, hist[6];
            const int r2 = rtxh_bvh8_check(w.data(), nt, &n8, &st), r3 = rtxh_bvh8_stats(w.data(), nt, hist, &n8);
            printf("builder %s=%g: bvh8 rc=%d nodes=%u stack=%u stats rc=%d\n", m.key, m.on, r2, n8, st, r3);
            rtxh_bvh_option(m.key, m.off); rtxh_bvh_option("ploc_top", 16384.0);
            if (r2 || r3) return 6;
        }
        rtxh_scene* sh = rtxh_scene_sponza_class_hard(60000, 260); rtxh_scene* bh = rtxh_scene_bistro_class_hard(120000, 3800);
        if (!sh || !bh) { printf("hard scene creation failed\n"); return 6; }
        printf("hard scenes: %u / %u triangles\n", rtxh_scene_num_triangles(sh), rtxh_scene_num_triangles(bh));
        rtxh_scene_free(sh); rtxh_scene_free(bh);
    }
    // image writers
    const std::string out = argc > 2 ? argv[2] : "/tmp";
    std::vector<uint8_t> img(37 * 21 * 4, 128); std::vector<float> acc(37 * 21 * 4, 1.5f);
    printf("png %d ppm %d exr %d\n", rtxh_write_png((out + "/a.png").c_str(), img.data(), 37, 21), rtxh_write_ppm((out + "/a.ppm").c_str(), img.data(), 37, 21), rtxh_write_exr((out + "/a.exr").c_str(), acc.data(), 37, 21));
    float lut[16]; rtxh_generate_ess_lut(0.5f, lut); printf("lut0 %f\n", lut[0]);
    // binary scene cache: save (host build), load, and every way a file can be damaged (must be refused without touching freed / foreign memory)
    for (rtxh_scene* sc : {c, g, b}) {
        const std::string path = out + "/cache.rtxscn";
        if (rtxh_scene_save(sc, path.c_str()) != 0) { printf("cache save failed: %s\n", rtxh_last_error()); return 4; }
        rtxh_scene* back = rtxh_scene_load(path.c_str());
        if (!back || rtxh_scene_num_triangles(back) != rtxh_scene_num_triangles(sc) || rtxh_scene_num_materials(back) != rtxh_scene_num_materials(sc)) { printf("cache load failed\n"); return 4; }
        rtxh_scene_free(back);
        FILE* f = fopen(path.c_str(), "rb"); fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> blob((size_t)len); if (fread(blob.data(), 1, blob.size(), f) != blob.size()) return 4; fclose(f);
        auto refused = [&](const std::vector<uint8_t>& d) { const std::string bp = out + "/bad.rtxscn"; FILE* w = fopen(bp.c_str(), "wb"); if (!d.empty()) fwrite(d.data(), 1, d.size(), w); fclose(w);
                                                             rtxh_scene* x = rtxh_scene_load(bp.c_str()); if (x) { rtxh_scene_free(x); return false; } return true; };
        std::vector<uint8_t> t(blob.begin(), blob.begin() + (long)(blob.size() / 3)), fl = blob, ver = blob, tiny(blob.begin(), blob.begin() + 40);
        fl[fl.size() - 100] ^= 0x40; ver[8] ^= 3;
        if (!refused(t) || !refused(fl) || !refused(ver) || !refused(tiny) || !refused({})) { printf("a damaged cache file was accepted\n"); return 4; }
    }
    printf("scene cache ok\n");
    // MTL extension records of an OBJ scene
    rtxh_material_ext mx; uint32_t nx = 0; while (rtxh_scene_material_ext(g, nx, &mx) == 0) nx++;
    printf("material ext records %u textures %u\n", nx, rtxh_scene_num_textures(g));
    if (nx != rtxh_scene_num_materials(g)) return 5;
    rtxh_scene_free(c); rtxh_scene_free(s); rtxh_scene_free(b); rtxh_scene_free(g);
    printf("done\n");
    return 0;
}
