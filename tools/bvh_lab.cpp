// bvh_lab.cpp — CPU laboratory for the BVH builder (tooling; no GPU, no HIP).  Builds a scene's wide BVH with the product's own
// builder code (csrc/rtx_scene_host.cpp compiled into this executable) and replays the device traversal of csrc/rtx_traverse.hpp
// (node8_hits / descend8 / traverse, non-speculative order) in scalar C++ on path-like rays (camera rays + cosine-sampled bounces),
// counting node steps and triangle tests per ray.  The point: judge a builder change by WORK PER RAY here, in seconds, before
// spending GPU minutes on frame times.
//   make -C tools bvh_lab && tools/bvh_lab sponza|bistro|garage [key=value ...]
// Keys are handed to rtx::bvh_build_options() (see csrc/rtx_scene_host.hpp) — e.g. sweep=4096 reinsert=2 split=1e-5
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../royaltracer-dx_amd/csrc/rtx_scene_host.hpp"
#include "../royaltracer-dx_amd/host/Scenes.h"

using namespace rtx;

// host/Scenes.cpp's UploadScene refers to the C-ABI; this tool never uploads anything
extern "C" {
int rtx_set_materials(rtx_ctx*, const void*, uint32_t) { return -1; }
int rtx_add_mesh(rtx_ctx*, const void*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, uint32_t*) { return -1; }
int rtx_add_instance(rtx_ctx*, uint32_t, const float*, uint32_t*) { return -1; }
int rtx_commit_scene(rtx_ctx*) { return -1; }
int rtx_set_camera(rtx_ctx*, const float*, const float*) { return -1; }
}

namespace {
struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 crs(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dt(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 nrm(V3 a) { float l = sqrtf(dt(a, a)); return l > 0 ? a * (1.0f / l) : a; }

inline float u2f_(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t f2u_(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

constexpr float kPlaneEps = 2.384185791015625e-07f, kSlabK = 1.00010002f;


// the product's triangle test (csrc/rtx_traverse.hpp: tri_test) in the product's own arithmetic — rtx_math.hpp's FUSED dot / cross and the determinant floor in e1.w.
// (Until round 5 this was an unfused restatement: the what-if rows then "found" 4 / 9 mismatching rays per 648 000 against the replay, which were nothing but
// hits on an edge — u or v exactly 0, u + v one ulp past 1 — decided differently by the two roundings.  With one arithmetic they are 0.)
inline bool tri_test(V3 o, V3 d, const TriGPU& Tg, float tmin, float tmax, float& t, float& u, float& v) {
    const f3 v0 = mk3(Tg.v0.x, Tg.v0.y, Tg.v0.z), e1 = mk3(Tg.e1.x, Tg.e1.y, Tg.e1.z), e2 = mk3(Tg.e2.x, Tg.e2.y, Tg.e2.z), dd = mk3(d.x, d.y, d.z);
    const f3 p = cross(dd, e2);
    const float det = dot(e1, p);
    if (!(fabsf(det) > Tg.e1.w)) return false;
    const float inv = 1.0f / det;
    const f3 s = mk3(o.x, o.y, o.z) - v0;
    u = dot(s, p) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 q = cross(s, e1);
    v = dot(dd, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot(e2, q) * inv;
    return t > tmin && t < tmax;
}

struct Hit { float t; uint32_t slot, prim; uint32_t steps, tris; };

int g_any_order = 0;      // any-hit visiting order: 0 = slot order, 1 = near first (octant order), 2 = far first
// the product's own host-side replay of the device traversal (csrc/rtx_scene_host.cpp: replay_trace)
template <bool ANY>
Hit traverse(const BuiltScene& B, V3 o, V3 d, float tmin, float tmax, float tknown = -1.0f) {
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    const ReplayHit R = replay_trace(B, oo, dd, tmin, tmax, ANY, ANY ? (uint32_t)g_any_order : 0u, tknown);
    Hit H{R.t, R.slot, ANY ? (R.prim == 0xffffffffu ? 0xffffffffu : 0u) : R.prim, R.steps, R.tris};
    return H;
}

// What-if traversal (closest hit): mode 1 = octant order, a child whose entry distance lies beyond the best hit WHEN IT IS POPPED is skipped without a node step
// (needs per-child distances on the stack); mode 2 = children visited by entry distance + the same skip.  Counts only.
struct WhatIf { const BuiltScene& B; V3 o, d, idir; uint32_t oct; float tmin; float bt; uint32_t bprim; uint32_t steps, tris; int mode; double* lev = nullptr; double* zero = nullptr;
    void node(uint32_t idx, int depth = 0) {
        const Node8GPU& N = B.nodes8[idx]; steps++; if (lev && depth < 16) lev[depth] += 1;
        const uint32_t w = N.e_imask;
        const float s[3] = {u2f_((w & 0xffu) << 23) * idir.x, u2f_((w & 0xff00u) << 15) * idir.y, u2f_((w & 0xff0000u) << 7) * idir.z};
        const float a3[3] = {(N.px - o.x) * idir.x, (N.py - o.y) * idir.y, (N.pz - o.z) * idir.z};
        float lo8[8]; uint32_t hits = 0;
        for (int k = 0; k < 8; k++) {
            float lo = tmin, hi = bt;
            for (int a = 0; a < 3; a++) {
                const uint32_t qlo = (N.q[2 * a + (k >> 2)] >> (8 * (k & 3))) & 0xffu, qhi = (N.q[2 * (3 + a) + (k >> 2)] >> (8 * (k & 3))) & 0xffu;
                const bool neg = (oct >> a) & 1u;
                const float an = fmaf(-fabsf(a3[a]), kPlaneEps, a3[a]), af = fmaf(fabsf(a3[a]), kPlaneEps, a3[a]);
                lo = fmaxf(lo, fmaf((float)(neg ? qhi : qlo), s[a], an)); hi = fminf(hi, fmaf((float)(neg ? qlo : qhi), s[a], af));
            }
            lo8[k] = lo;
            if (!(f2u_(fmaf(hi, kSlabK, -lo)) >> 31)) hits |= 1u << k;
        }
        const uint32_t imask = w >> 24;
        if (zero && !hits && depth < 16) zero[depth] += 1;
        // leaves first (as the device does: the triangles of a node are tested before its internal children are entered)
        uint32_t x = hits & ~imask;
        for (int k = 0; k < 8; k++) if (x & (1u << k)) {
            const uint32_t nib = (N.trivalid >> (4 * k)) & 0xfu;
            for (int b = 0; b < 4; b++) if (nib & (1u << b)) {
                const uint32_t slot = N.tri_base + (uint32_t)__builtin_popcount(N.trivalid & ((1u << (4 * k + b)) - 1u));
                float t, u, v; tris++;
                if (tri_test(o, d, B.tris8[slot], tmin, 1e30f, t, u, v)) { const uint32_t g = f2u_(B.tris8[slot].v0.w); if (t < bt || (t == bt && g < bprim)) { bt = t; bprim = g; } }
            }
        }
        int ord[8], m = 0;
        for (int j = 0; j < 8; j++) { const int k = j ^ (int)oct; if ((hits & imask) & (1u << k)) ord[m++] = k; }
        if (mode == 2 || mode == 4) std::stable_sort(ord, ord + m, [&](int a, int b) { return lo8[a] < lo8[b]; });
        for (int i = 0; i < m; i++) {
            const int k = ord[i];
            if ((mode == 1 || mode == 2) && lo8[k] > bt * kSlabK) continue;
            node(N.child_base + (uint32_t)__builtin_popcount(imask & ((1u << k) - 1u)), depth + 1);
        }
    }
};

// flat=1 — sizing of VERDICT r04 item 2: "replace the three dependent steps at the top of the tree by ONE flat, wave-uniform slab test of the <= 64 level-2 boxes".
// The 64 boxes are the child slots of the root's <= 8 children; a flat test of them stands for the node steps at depths 0 and 1 (which test exactly these boxes — the step
// at depth 2 tests the level-3 boxes and stays).  It runs once per ray, before any hit is known, so it cannot cull by the closest distance: counted here are the slots a ray's
// box test accepts over (tmin, inf) — internal ones (each costs a node step at depth 2 unless re-tested against the best hit when it is popped) and leaf slots.
inline void flat_level2(const BuiltScene& B, V3 o, V3 idir, uint32_t oct, float tmin, uint32_t& boxes, uint32_t& hit_internal, uint32_t& hit_leaf) {
    boxes = hit_internal = hit_leaf = 0;
    if (B.nodes8.empty()) return;
    const Node8GPU& R = B.nodes8[0];
    const uint32_t rmask = R.e_imask >> 24;
    for (uint32_t r = 0; r < (uint32_t)__builtin_popcount(rmask); r++) {
        const Node8GPU& N = B.nodes8[R.child_base + r];
        const uint32_t w = N.e_imask, imask = w >> 24;
        const float s[3] = {u2f_((w & 0xffu) << 23) * idir.x, u2f_((w & 0xff00u) << 15) * idir.y, u2f_((w & 0xff0000u) << 7) * idir.z};
        const float a3[3] = {(N.px - o.x) * idir.x, (N.py - o.y) * idir.y, (N.pz - o.z) * idir.z};
        for (int k = 0; k < 8; k++) {
            const bool internal = (imask >> k) & 1u; const uint32_t nib = (N.trivalid >> (4 * k)) & 0xfu;
            if (!internal && !nib) continue;
            boxes++;
            float lo = tmin, hi = 1e30f;
            for (int a = 0; a < 3; a++) {
                const uint32_t qlo = (N.q[2 * a + (k >> 2)] >> (8 * (k & 3))) & 0xffu, qhi = (N.q[2 * (3 + a) + (k >> 2)] >> (8 * (k & 3))) & 0xffu;
                const bool neg = (oct >> a) & 1u;
                const float an = fmaf(-fabsf(a3[a]), kPlaneEps, a3[a]), af = fmaf(fabsf(a3[a]), kPlaneEps, a3[a]);
                lo = fmaxf(lo, fmaf((float)(neg ? qhi : qlo), s[a], an)); hi = fminf(hi, fmaf((float)(neg ? qlo : qhi), s[a], af));
            }
            if (!(f2u_(fmaf(hi, kSlabK, -lo)) >> 31)) { if (internal) hit_internal++; else hit_leaf++; }
        }
    }
}

// ---- wave-schedule simulation (wavesim=1): the speculative voted schedule of csrc/rtx_traverse.hpp spec_step (PEND 2, vote "node step if ni >= 2 nl") replayed on the
// per-ray step sequences of the host replay, with the persistent refill (>= refill idle SLOTS), for ONE or TWO rays per lane.  Cost model: node iteration 205 + 12, triangle
// iteration 70 + 12 VALU instructions per wave (a wave instruction costs the same whatever its active lanes).  What it cannot see: culling changes from the order in which
// pending triangles shorten a ray (the sequences are the non-speculative order), memory stalls, occupancy.
struct SimRay { const uint8_t* seq; uint32_t n, idx, T, P; bool busy; };
struct SimOut { double cost = 0, node_it = 0, tri_it = 0, node_lanes = 0, tri_lanes = 0, rays = 0; };
static SimOut wave_sim(const std::vector<std::vector<uint8_t>>& seqs, size_t first, size_t count, int slots_per_lane, uint32_t refill_min, double vote = 2.0, int pend = 1) {
    SimOut S; const int NS = 64 * slots_per_lane;
    std::vector<SimRay> R(NS, SimRay{nullptr, 0, 0, 0, 0, false});
    size_t next = first; const size_t end = first + count;
    auto done = [](const SimRay& r) { return r.idx >= r.n && r.T == 0 && r.P == 0; };
    for (;;) {
        int idle = 0; for (auto& r : R) if (!r.busy) idle++;
        if (next < end && (idle >= (int)refill_min * slots_per_lane || idle == NS)) {
            for (auto& r : R) if (!r.busy && next < end) { const auto& q = seqs[next++]; r = SimRay{q.data(), (uint32_t)q.size(), 0, 0, 0, true}; if (done(r)) r.busy = false; S.rays++; }
            S.cost += 60;                                              // refill: atomic, loads, ray set-up
        }
        int ni = 0, nl = 0;
        for (int l = 0; l < 64; l++) {
            bool cn = false, ht = false;
            for (int k = 0; k < slots_per_lane; k++) { const SimRay& r = R[l * slots_per_lane + k]; if (!r.busy) continue; cn = cn || (r.P == 0 && r.idx < r.n); ht = ht || r.T > 0; }
            ni += cn; nl += ht;
        }
        if (!ni && !nl) { if (next >= end) break; continue; }
        if ((double)ni >= vote * (double)nl && ni) {
            S.cost += 217; S.node_it++; S.node_lanes += ni;
            for (int l = 0; l < 64; l++) for (int k = 0; k < slots_per_lane; k++) { SimRay& r = R[l * slots_per_lane + k]; if (r.busy && r.P == 0 && r.idx < r.n) { r.P = r.seq[r.idx++]; if (!r.T) { r.T = r.P; r.P = 0; } break; } }
        } else {
            S.cost += 82; S.tri_it++; S.tri_lanes += nl;
            for (int l = 0; l < 64; l++) for (int k = 0; k < slots_per_lane; k++) { SimRay& r = R[l * slots_per_lane + k]; if (r.busy && r.T > 0) { r.T--; if (!r.T) { r.T = r.P; r.P = 0; } break; } }
        }
        for (auto& r : R) if (r.busy && done(r)) r.busy = false;
    }
    return S;
}

inline uint32_t hash32(uint32_t a, uint32_t b) { uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15; return h; }
inline float rnd(uint32_t a, uint32_t b) { return (float)(hash32(a, b) >> 8) * (1.0f / 16777216.0f); }

// SAH cost of the wide tree in units of (node step, triangle test) weighted by the probability of a random ray hitting the box, relative to the root
void wide_sah(const BuiltScene& B, double& node_cost, double& tri_cost, double& fill, uint32_t hist[5]) {
    node_cost = tri_cost = 0; fill = 0; for (int i = 0; i < 5; i++) hist[i] = 0;
    if (B.nodes8.empty()) return;
    struct It { uint32_t n; double area; };
    std::vector<It> st; st.push_back({0, -1.0});
    double root_area = 0;
    size_t slots = 0;
    while (!st.empty()) {
        It it = st.back(); st.pop_back();
        const Node8GPU& N = B.nodes8[it.n];
        double step[3]; for (int a = 0; a < 3; a++) step[a] = std::ldexp(1.0, (int)((N.e_imask >> (8 * a)) & 0xff) - 127);
        double umn[3] = {1e300, 1e300, 1e300}, umx[3] = {-1e300, -1e300, -1e300};
        const uint32_t imask = N.e_imask >> 24; uint32_t rank = 0;
        struct C { double area; bool internal; uint32_t cnt; } ch[8]; int m = 0;
        for (int sl = 0; sl < 8; sl++) {
            const uint32_t nib = (N.trivalid >> (4 * sl)) & 0xf; const bool internal = (imask >> sl) & 1;
            if (!internal && !nib) continue;
            double e[3];
            for (int a = 0; a < 3; a++) {
                const double qlo = (N.q[2 * a + (sl >> 2)] >> (8 * (sl & 3))) & 0xff, qhi = (N.q[2 * (3 + a) + (sl >> 2)] >> (8 * (sl & 3))) & 0xff;
                e[a] = (qhi - qlo) * step[a]; umn[a] = std::min(umn[a], qlo * step[a]); umx[a] = std::max(umx[a], qhi * step[a]);
            }
            ch[m++] = {e[0] * e[1] + e[1] * e[2] + e[2] * e[0], internal, (uint32_t)__builtin_popcount(nib)};
            slots++;
        }
        double my_area = it.area;
        if (my_area < 0) { const double e0 = umx[0] - umn[0], e1 = umx[1] - umn[1], e2 = umx[2] - umn[2]; my_area = root_area = e0 * e1 + e1 * e2 + e2 * e0; }
        node_cost += my_area;
        for (int k = 0; k < m; k++) {
            if (ch[k].internal) st.push_back({N.child_base + rank++, ch[k].area});
            else { tri_cost += ch[k].area * ch[k].cnt; hist[ch[k].cnt]++; }
        }
    }
    node_cost /= root_area; tri_cost /= root_area; fill = (double)slots / (double)B.nodes8.size();
}
}  // namespace

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "sponza";
    int W = 480, Hh = 270, bounces = 4; bool check = false, lower = false, wavesim = false, flat = false; int whatif = 0;
    for (int i = 2; i < argc; i++) {
        std::string kv = argv[i]; const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq); const double v = atof(kv.c_str() + eq + 1);
        if (k == "w") W = (int)v; else if (k == "h") Hh = (int)v; else if (k == "bounces") bounces = (int)v; else if (k == "check") check = v != 0; else if (k == "lower") lower = v != 0; else if (k == "wavesim") wavesim = v != 0; else if (k == "whatif") whatif = (int)v; else if (k == "flat") { flat = v != 0; if (flat && !whatif) whatif = 3; } else if (k == "any_order") g_any_order = (int)v;
        else if (!bvh_build_option(k.c_str(), v)) { fprintf(stderr, "unknown key %s\n", k.c_str()); return 2; }
    }
    Scene s;
    if (which == "sponza") s = MakeSponzaClass();
    else if (which == "bistro") s = MakeBistroClass();
    else if (which == "sponza_hard") s = MakeSponzaClass(262144, 260, true);       // the size distribution of the real asset (host/Scenes.h)
    else if (which == "bistro_hard") s = MakeBistroClass(3800000, 3800, true);
    else if (which == "garage") s = LoadObjScene({"tests/golden/garage.obj", "tests/golden/monke.obj"}, "tests/golden/");
    else if (which == "cornell") s = MakeCornellBox();
    else { fprintf(stderr, "scene?\n"); return 2; }
    SceneHost Hs;
    Hs.set_materials(s.materials.data(), (uint32_t)s.materials.size());
    for (auto& m : s.models) { uint32_t id; if (!Hs.add_mesh(m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id)) { fprintf(stderr, "%s\n", Hs.err.c_str()); return 1; } }
    for (auto& in : s.instances) { uint32_t id; Hs.add_instance(in.model, in.transform.data(), &id); }
    BuiltScene B;
    const auto t0 = std::chrono::steady_clock::now();
    if (!Hs.build(B)) { fprintf(stderr, "build: %s\n", Hs.err.c_str()); return 1; }
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double nc, tc, fill; uint32_t hist[5];
    wide_sah(B, nc, tc, fill, hist);
    printf("any-hit order chosen by the commit-time probe: %u\n", B.any_order);
    { uint64_t h = 1469598103934665603ull; auto mix = [&](const void* d, size_t nb) { const uint8_t* q = (const uint8_t*)d; for (size_t i = 0; i < nb; i++) { h ^= q[i]; h *= 1099511628211ull; } };
      if (!B.nodes8.empty()) mix(B.nodes8.data(), B.nodes8.size() * sizeof(Node8GPU)); if (!B.tris8.empty()) mix(B.tris8.data(), B.tris8.size() * sizeof(TriGPU));
      printf("tree hash (nodes8 + tris8, FNV-1a) %016llx\n", (unsigned long long)h); }
    printf("%s: tris %zu refs %zu nodes2 %zu nodes8 %zu stack %u build %.2fs | wide SAH node %.2f tri %.2f | slots/node %.2f leaf hist 1:%u 2:%u 3:%u 4:%u\n", which.c_str(), B.shade.size(), B.tris8.size(),
           B.nodes.size(), B.nodes8.size(), B.stack8, build_s, nc, tc, fill, hist[1], hist[2], hist[3], hist[4]);
    // camera
    float view[16], proj[16];
    SceneViewProj(s, (float)W / Hh, view, proj);
    float vI[16], pI[16]; mat4_inverse(view, vI); mat4_inverse(proj, pI);
    auto mulp = [](const float* m, float x, float y, float z, float w, float* o) { for (int c = 0; c < 4; c++) o[c] = x * m[c] + y * m[4 + c] + z * m[8 + c] + w * m[12 + c]; };   // row-vector convention (DirectXMath)
    const int npx = W * Hh;
    std::vector<double> st_steps(bounces + 1, 0), st_tris(bounces + 1, 0), sh_steps(bounces + 1, 0), sh_tris(bounces + 1, 0); std::vector<size_t> cnt(bounces + 1, 0), shc(bounces + 1, 0), hitc(bounces + 1, 0);
    // a point on the first light for shadow rays
    V3 lightp{0, 0, 0}; bool have_light = !B.lights.empty();
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<std::vector<uint8_t>> seq_closest(wavesim ? (size_t)npx : 0);
    size_t mism = 0; double low_steps = 0, low_tris = 0; double lev_all[16] = {0}, zero_all[16] = {0}; double g_occ[4] = {0, 0, 0, 0}; double flat_all[3] = {0, 0, 0};
#pragma omp parallel
    {
        std::vector<double> a(bounces + 1, 0), b(bounces + 1, 0), sa(bounces + 1, 0), sb(bounces + 1, 0); std::vector<size_t> c(bounces + 1, 0), sc(bounces + 1, 0), hc(bounces + 1, 0); size_t mm = 0; double la = 0, lb = 0; double levl[16] = {0}, zerol[16] = {0}; double occ_n = 0, occ_steps = 0, vis_n = 0, vis_steps = 0; double flatl[3] = {0, 0, 0};
#pragma omp for schedule(dynamic, 64)
        for (int px = 0; px < npx; px++) {
            const int x = px % W, y = px / W;
            const float dx = ((x + 0.5f) / W) * 2.0f - 1.0f, dy = 1.0f - ((y + 0.5f) / Hh) * 2.0f;
            float org[4], tgt[4], dir4[4];
            mulp(vI, 0, 0, 0, 1, org); mulp(pI, dx, dy, 1, 1, tgt);
            V3 tg = nrm(V3{tgt[0], tgt[1], tgt[2]}); mulp(vI, tg.x, tg.y, tg.z, 0, dir4);
            V3 o{org[0], org[1], org[2]}, d = nrm(V3{dir4[0], dir4[1], dir4[2]});
            float tmin = 1e-4f;
            for (int bnc = 0; bnc <= bounces; bnc++) {
                std::vector<uint8_t> sq;
                const Hit H = wavesim ? [&] { const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}; const ReplayHit Rh = replay_trace(B, oo, dd, tmin, 1e30f, false, 0u, -1.0f, &sq); return Hit{Rh.t, Rh.slot, Rh.prim, Rh.steps, Rh.tris}; }() : traverse<false>(B, o, d, tmin, 1e30f);
                if (wavesim && bnc == 1) seq_closest[px] = sq;
                a[bnc] += H.steps; b[bnc] += H.tris; c[bnc]++;
                if (whatif) { const float dxs = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x, dys = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y, dzs = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
                    const V3 idir{1.0f / dxs, 1.0f / dys, 1.0f / dzs};
                    WhatIf Wf{B, o, d, idir, (idir.x < 0 ? 1u : 0u) | (idir.y < 0 ? 2u : 0u) | (idir.z < 0 ? 4u : 0u), tmin, 1e30f, 0xffffffffu, 0, 0, whatif == 3 ? 0 : whatif, levl, zerol};      // 3: the device's order and culling (depth statistics), 4: visit by entry distance, no skip
                    Wf.node(0); la += Wf.steps; lb += Wf.tris; if (Wf.bprim != H.prim) mm++;
                    if (flat) { uint32_t nb, hi_, hl_; flat_level2(B, o, idir, Wf.oct, tmin, nb, hi_, hl_); flatl[0] += nb; flatl[1] += hi_; flatl[2] += hl_; } }
                if (lower) { const Hit H2 = traverse<false>(B, o, d, tmin, 1e30f, H.prim == 0xffffffffu ? 1e30f : H.t); la += H2.steps; lb += H2.tris; }
                if (H.prim == 0xffffffffu) break;
                hc[bnc]++;
                if (check) {      // brute force over all references
                    float bt = 1e30f; uint32_t bp = 0xffffffffu;
                    for (size_t i = 0; i < B.tris8.size(); i++) { float t, u, v; if (tri_test(o, d, B.tris8[i], tmin, 1e30f, t, u, v)) { const uint32_t g = f2u_(B.tris8[i].v0.w); if (t < bt || (t == bt && g < bp)) { bt = t; bp = g; } } }
                    if (bp != H.prim || bt != H.t) mm++;
                }
                const TriGPU& Tg = B.tris8[H.slot];
                V3 n = nrm(crs(V3{Tg.e1.x, Tg.e1.y, Tg.e1.z}, V3{Tg.e2.x, Tg.e2.y, Tg.e2.z}));
                if (dt(n, d) > 0) n = n * -1.0f;
                const V3 pos = o + d * H.t;
                o = pos + n * 2e-5f;
                // shadow ray towards a point on a light (any-hit), like NEE
                if (have_light) {
                    const uint32_t li = hash32((uint32_t)px, 77u + bnc) % (uint32_t)B.lights.size(); const LightGPU& Lg = B.lights[li];
                    float r1 = rnd((uint32_t)px, 100u + bnc), r2 = rnd((uint32_t)px, 200u + bnc); if (r1 + r2 > 1) { r1 = 1 - r1; r2 = 1 - r2; }
                    const V3 lp{Lg.xv[0] + r1 * (Lg.yv[0] - Lg.xv[0]) + r2 * (Lg.zv[0] - Lg.xv[0]), Lg.xv[1] + r1 * (Lg.yv[1] - Lg.xv[1]) + r2 * (Lg.zv[1] - Lg.xv[1]), Lg.xv[2] + r1 * (Lg.yv[2] - Lg.xv[2]) + r2 * (Lg.zv[2] - Lg.xv[2])};
                    V3 sd = lp - o; const float dist = sqrtf(dt(sd, sd)); sd = sd * (1.0f / dist);
                    if (dt(sd, n) > 0) { const Hit S = traverse<true>(B, o, sd, 2e-5f, dist - 1e-4f); sa[bnc] += S.steps; sb[bnc] += S.tris; sc[bnc]++; if (S.prim == 0) { occ_n++; occ_steps += S.steps; } else { vis_n++; vis_steps += S.steps; } }
                }
                // cosine-sampled bounce
                const float u1 = rnd((uint32_t)px, 300u + bnc), u2 = rnd((uint32_t)px, 400u + bnc);
                const float r = sqrtf(u1), ph = 6.2831853f * u2;
                const V3 tx = nrm(fabsf(n.x) > 0.5f ? crs(n, V3{0, 1, 0}) : crs(n, V3{1, 0, 0})), ty = crs(n, tx);
                d = nrm(tx * (r * cosf(ph)) + ty * (r * sinf(ph)) + n * sqrtf(std::max(0.0f, 1.0f - u1)));
                tmin = 2e-5f;
            }
        }
#pragma omp critical
        { for (int i = 0; i <= bounces; i++) { st_steps[i] += a[i]; st_tris[i] += b[i]; cnt[i] += c[i]; sh_steps[i] += sa[i]; sh_tris[i] += sb[i]; shc[i] += sc[i]; hitc[i] += hc[i]; } mism += mm; low_steps += la; low_tris += lb; g_occ[0] += occ_n; g_occ[1] += occ_steps; g_occ[2] += vis_n; g_occ[3] += vis_steps; for (int i = 0; i < 16; i++) { lev_all[i] += levl[i]; zero_all[i] += zerol[i]; } for (int i = 0; i < 3; i++) flat_all[i] += flatl[i]; }
    }
    const double sim_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    double ts = 0, tt = 0, ss = 0, stt = 0; size_t tc_ = 0, sc_ = 0;
    for (int i = 0; i <= bounces; i++) {
        printf("  bounce %d: rays %zu hit %.3f steps/ray %.2f tris/ray %.2f | shadow rays %zu steps/ray %.2f tris/ray %.2f\n", i, cnt[i], cnt[i] ? (double)hitc[i] / cnt[i] : 0.0, cnt[i] ? st_steps[i] / cnt[i] : 0.0, cnt[i] ? st_tris[i] / cnt[i] : 0.0,
               shc[i], shc[i] ? sh_steps[i] / shc[i] : 0.0, shc[i] ? sh_tris[i] / shc[i] : 0.0);
        ts += st_steps[i]; tt += st_tris[i]; tc_ += cnt[i]; ss += sh_steps[i]; stt += sh_tris[i]; sc_ += shc[i];
    }
    // cost proxy in VALU lane-slots per ray: node step 205 at 47/64 lanes, triangle test 70 at 24/64 lanes (profiles/r02_traversal.md)
    const double cs = ts / tc_, ct = tt / tc_, hs = sc_ ? ss / sc_ : 0, ht = sc_ ? stt / sc_ : 0;
    printf("  ALL closest: steps/ray %.3f tris/ray %.3f cost %.0f | shadow: steps/ray %.3f tris/ray %.3f cost %.0f | sim %.1fs%s\n", cs, ct, cs * 205 * 64 / 47 + ct * 70 * 64 / 24, hs, ht, hs * 205 * 64 / 47 + ht * 70 * 64 / 24, sim_s,
           check ? (mism ? "  BRUTE-FORCE MISMATCH" : "  brute force: equal") : "");
    printf("  shadow rays: occluded %.3f (steps/ray %.2f), visible steps/ray %.2f\n", g_occ[0] / std::max(1.0, g_occ[0] + g_occ[2]), g_occ[1] / std::max(1.0, g_occ[0]), g_occ[3] / std::max(1.0, g_occ[2]));
    if (wavesim) {
        std::vector<std::vector<uint8_t>> qs; for (auto& q : seq_closest) if (!q.empty()) qs.push_back(q);
        // queue order = pixel order (neighbouring pixels' bounce-1 rays, like a sub-queue); one wave owns 2 048 consecutive rays
        for (int spl = 1; spl <= 3; spl++) for (double vote : {2.0, 1.0, 0.5, 0.25}) { const uint32_t rf = 12u;
            SimOut T; const size_t per = 2048;
            for (size_t f = 0; f + per <= qs.size(); f += per) { const SimOut S = wave_sim(qs, f, per, spl, rf, vote); T.cost += S.cost; T.node_it += S.node_it; T.tri_it += S.tri_it; T.node_lanes += S.node_lanes; T.tri_lanes += S.tri_lanes; T.rays += S.rays; }
            printf("  wave sim, bounce-1 closest-hit rays, %d ray(s) per lane, refill at %u idle, node step if ni >= %.2f nl: cost %.0f VALU wave-instructions per ray x64 = %.1f per ray; node iterations %.1f lanes, triangle iterations %.1f lanes, %.2f / %.2f iterations per ray\n",
                   spl, rf, vote, T.cost / T.rays * 64, T.cost / T.rays, T.node_lanes / T.node_it, T.tri_lanes / T.tri_it, T.node_it * 64 / T.rays, T.tri_it * 64 / T.rays);
        }
    }
    if (whatif) printf("  what-if %d (1: skip popped children beyond the best hit; 2: + visit by entry distance): steps/ray %.3f tris/ray %.3f mismatches %zu\n", whatif, low_steps / tc_, low_tris / tc_, mism);
    if (whatif) { printf("  steps per ray by depth (zero-hit share):"); for (int i = 0; i < 16 && lev_all[i] > 0; i++) printf(" %d: %.2f (%.0f%%)", i, lev_all[i] / tc_, 100.0 * zero_all[i] / lev_all[i]); printf("\n"); }
    if (flat) {
        const double d01 = (lev_all[0] + lev_all[1]) / tc_, d2 = lev_all[2] / tc_, fb = flat_all[0] / tc_, fi = flat_all[1] / tc_, fl = flat_all[2] / tc_;
        printf("  flat test of the level-2 boxes (children of the root's children): %.1f boxes per ray; accepted over (tmin, inf): %.2f internal + %.2f leaf slots per ray\n", fb, fi, fl);
        printf("    the traversal as it is: %.2f node steps per ray at depths 0 + 1 (what the flat test replaces), %.2f at depth 2 (culled by the closest hit so far)\n", d01, d2);
        // VALU wave-instructions per ray: a node step is 217 (205 + 12 of loop) at 47 of 64 lanes; the flat test ~13 per box (3 packed FMA + 6 min / max + min3 / max3 + compare + v_addc_co), wave-uniform
        const double cur = (d01 + d2) * 217.0 / 47.0;
        const double at64 = fb * 13.0 / 64.0 + fi * 217.0 / 47.0, at16 = fb * 13.0 / 16.0 + fi * 217.0 / 47.0, retest = fb * 13.0 / 64.0 + fi * 30.0 / 47.0 + d2 * 217.0 / 47.0;
        printf("    VALU wave-instructions per ray for depths 0-2: as it is %.1f | flat at 64 lanes (a pre-pass over the whole sub-queue), every accepted slot stepped %.1f | flat inside the persistent loop, where rays arrive %u at a time (refill threshold) %.1f | flat at 64 lanes + a 30-instruction re-test of every accepted slot against the best hit at pop (lower bound: as many depth-2 steps as today) %.1f\n",
               cur, at64, 16u, at16, retest);
    }
    if (lower) printf("  with the closest distance known in advance (bound for any visiting order): steps/ray %.3f tris/ray %.3f\n", low_steps / tc_, low_tris / tc_);
    if (check && mism) { printf("  mismatches: %zu\n", mism); return 1; }
    return 0;
}
