#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <string>
#include <vector>
#include "../royaltracer-dx_amd/csrc/rtx_scene_host.hpp"
using namespace rtx;
static inline uint32_t f2u_(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
// soup_lab — is the closest / any hit the same for the product's tree as for brute force over all triangles?  No GPU: the product's builder (SceneHost::build) and the
// host replay of the device traversal (csrc/rtx_scene_host.cpp: replay_trace) against a loop over every triangle with the replay's own triangle test, on triangle soups and
// rays chosen to break it: aimed at vertices / edges / interiors, LYING IN a triangle's plane, axis-parallel, random.  usage: soup_lab <needles|slivers|coplanar|mixed|random> [tris] [rays] [seed]
// Exit code 1 when the product's definition (second row) shows a mismatch.
int main(int argc, char** argv) {
    int rc = 0;
    const std::string kind = argc > 1 ? argv[1] : "needles";
    const int n = argc > 2 ? atoi(argv[2]) : 6000; const long m = argc > 3 ? atol(argv[3]) : 1000000; const int seed = argc > 4 ? atoi(argv[4]) : 1;
    std::mt19937_64 rng(seed); std::uniform_real_distribution<double> U(-1, 1); std::normal_distribution<double> N(0, 1);
    std::vector<float> tri((size_t)n * 9);
    for (int i = 0; i < n; i++) {
        double c[3] = {U(rng), U(rng), U(rng)}; double v[3][3];
        if (kind == "needles") { double a[3] = {U(rng), U(rng), U(rng)}, d[3] = {N(rng), N(rng), N(rng)}; double l = sqrt(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]); for (int k = 0; k < 3; k++) { d[k] /= l; v[0][k] = a[k]; v[1][k] = a[k] + 1.5 * d[k]; v[2][k] = a[k] + 1.5 * d[k] + 0.002 * N(rng); } }
        else if (kind == "coplanar") { for (int j = 0; j < 3; j++) { v[j][0] = c[0] + 0.05 * N(rng); v[j][1] = c[1] + 0.05 * N(rng); v[j][2] = 0.25; } }
        else if (kind == "mixed") { if (i & 1) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) v[j][k] = 50 * c[k] + 3.0 * N(rng); else for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) v[j][k] = 0.01 * c[k] + 1e-4 * N(rng); }
        else if (kind == "slivers") { double d[3] = {N(rng), N(rng), N(rng)}; double l = sqrt(d[0]*d[0]+d[1]*d[1]+d[2]*d[2]); double L = pow(10.0, -2 + 1.5 * (U(rng) + 1) / 2 * 1.3), w = L * pow(10.0, -1 - 4 * (U(rng) + 1) / 2); for (int k = 0; k < 3; k++) { d[k] /= l; v[0][k] = c[k]; v[1][k] = c[k] + L * d[k]; v[2][k] = c[k] + 0.5 * L * d[k] + w * N(rng); } }
        else { for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) v[j][k] = c[k] + 0.03 * N(rng); }
        for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) tri[(size_t)i * 9 + j * 3 + k] = (float)v[j][k];
    }
    SceneHost H; std::vector<float> mat(32, 0.0f); mat[0] = mat[1] = mat[2] = 0.7f; mat[3] = 1; mat[12] = 1; H.set_materials(mat.data(), 1);
    std::vector<float> verts((size_t)n * 3 * 7, 0.0f); std::vector<uint32_t> idx((size_t)n * 3), mid((size_t)n * 3, 0u);
    for (size_t i = 0; i < (size_t)n * 3; i++) { verts[i * 7] = tri[i * 3]; verts[i * 7 + 1] = tri[i * 3 + 1]; verts[i * 7 + 2] = tri[i * 3 + 2]; idx[i] = (uint32_t)i; }
    uint32_t mo, io; H.add_mesh(verts.data(), n * 3, idx.data(), n * 3, mid.data(), &mo);
    const float I[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1}; H.add_instance(mo, I, &io);
    BuiltScene B; if (!H.build(B)) { printf("build failed: %s\n", H.err.c_str()); return 1; }
    double lo[3] = {1e30,1e30,1e30}, hi[3] = {-1e30,-1e30,-1e30}; for (size_t i = 0; i < tri.size(); i++) { lo[i % 3] = std::min(lo[i % 3], (double)tri[i]); hi[i % 3] = std::max(hi[i % 3], (double)tri[i]); }
    const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    printf("%s: %d tris, %zu nodes8, ext %.2f, rays %ld\n", kind.c_str(), n, B.nodes8.size(), ext, m);
    const std::vector<TriGPU> tris_floor = B.tris8;
    for (int with_floor = 0; with_floor < 2; with_floor++) {          // 0: plain Moeller-Trumbore (det != 0), the definition until round 4; 1: the product's (TriGPU::e1.w = tri_det_floor)
        B.tris8 = tris_floor; if (!with_floor) for (TriGPU& T : B.tris8) T.e1.w = 0.0f;
        long mism = 0, hits = 0, bogus = 0, mism_any = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : mism, hits, bogus, mism_any)
        for (long r = 0; r < m; r++) {
            std::mt19937_64 q(seed * 1000003ull + (uint64_t)r); std::uniform_real_distribution<double> V(0, 1); std::normal_distribution<double> G(0, 1);
            double o[3], d[3]; for (int a = 0; a < 3; a++) o[a] = lo[a] - 0.1 * ext + V(q) * (hi[a] - lo[a] + 0.2 * ext);
            const int mode = (int)(r % 6);
            if (mode <= 2) {            // aimed at a triangle: interior / vertex / edge
                const size_t ti = (size_t)(V(q) * n) % n; double w0 = V(q), w1 = V(q), w2 = V(q); if (mode == 1) { w0 = 1; w1 = w2 = 0; } if (mode == 2) w2 = 0; const double ws = w0 + w1 + w2;
                for (int a = 0; a < 3; a++) d[a] = (w0 * tri[ti * 9 + a] + w1 * tri[ti * 9 + 3 + a] + w2 * tri[ti * 9 + 6 + a]) / ws - o[a];
            } else if (mode == 3) {     // inside a triangle's plane (origin on the plane, direction along it)
                const size_t ti = (size_t)(V(q) * n) % n; const float* T = &tri[ti * 9]; double a1 = 3 * G(q), a2 = 3 * G(q), b1 = G(q), b2 = G(q);
                for (int a = 0; a < 3; a++) { o[a] = T[a] + a1 * (T[3 + a] - T[a]) + a2 * (T[6 + a] - T[a]); d[a] = b1 * (T[3 + a] - T[a]) + b2 * (T[6 + a] - T[a]); }
            } else if (mode == 4) { const int ax = (int)(V(q) * 3) % 3; d[0] = d[1] = d[2] = 0; d[ax] = V(q) < 0.5 ? -1 : 1; }
            else for (int a = 0; a < 3; a++) d[a] = G(q);
            const double l = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]); if (!(l > 0)) continue;
            const float of[3] = {(float)o[0], (float)o[1], (float)o[2]}, df[3] = {(float)(d[0] / l), (float)(d[1] / l), (float)(d[2] / l)};
            const ReplayHit Hh = replay_trace(B, of, df, 1e-5f, 1e30f, false);
            float bt = 1e30f; uint32_t bp = 0xffffffffu; size_t bs = 0;
            for (size_t i = 0; i < B.tris8.size(); i++) { float t; if (replay_tri_test(of, df, B.tris8[i], 1e-5f, 1e30f, t)) { const uint32_t g = f2u_(B.tris8[i].v0.w); if (t < bt || (t == bt && g < bp)) { bt = t; bp = g; bs = i; } } }
            if (bp != 0xffffffffu) { hits++; }
            if (bp != Hh.prim || (bp != 0xffffffffu && bt != Hh.t)) { mism++; }
            // any-hit with a random tmax
            const float tmax = (float)(V(q) * ext);
            const ReplayHit Ha = replay_trace(B, of, df, 1e-5f, tmax, true);
            bool any = false; for (size_t i = 0; i < B.tris8.size() && !any; i++) { float t; any = replay_tri_test(of, df, B.tris8[i], 1e-5f, tmax, t); }
            if (any != (Ha.prim != 0xffffffffu)) mism_any++;
        }
        printf("  %s: closest hits %ld, closest-hit mismatches (product tree, host replay of the device traversal, vs brute force) %ld, any-hit mismatches %ld\n", with_floor ? "determinant floor 2^-16 |e1| |e2| (product)" : "plain det != 0 (until round 4)        ", hits, mism, mism_any);
        if (with_floor && (mism || mism_any)) rc = 1;
    }
    return rc;
}
