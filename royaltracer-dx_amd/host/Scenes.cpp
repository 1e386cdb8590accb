// Scenes.cpp — deterministic synthetic scenes in the reference's data model (see Scenes.h).
#include "Scenes.h"
#include <cmath>
#include <cstring>
#include <functional>
#include "ObjLoader.h"
#include "manipulator.h"

namespace {

struct P3 { float x, y, z; };
inline P3 operator+(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator-(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator*(P3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline P3 norm(P3 a) { float l = sqrtf(dot(a, a)); return l > 0 ? a * (1.0f / l) : a; }

// builds ONE model; triangles carry a LOCAL material index (0 = model default, k = k-th model material)
struct MeshBuilder {
    SceneModel m;
    UINT mat_base;        // global id of this model's default material
    float normal_w;       // Vertex.normal.w = base of this model in materialIDs[] (ObjLoader.h:466)
    bool count_only = false; size_t tri_count = 0;
    UINT vert(P3 p, P3 n) { m.vertices.emplace_back(XMFLOAT3(p.x, p.y, p.z), XMFLOAT4(n.x, n.y, n.z, normal_w)); return (UINT)m.vertices.size() - 1; }
    void tri(UINT a, UINT b, UINT c, UINT mat) {
        m.indices.push_back(a); m.indices.push_back(b); m.indices.push_back(c);
        for (int i = 0; i < 3; i++) m.materialIDs.push_back(mat_base + mat);
    }
    // flat-shaded triangle facing `toward` (winding chosen so that cross(e1,e2) . toward > 0)
    void flat_tri(P3 a, P3 b, P3 c, P3 toward, UINT mat) {
        tri_count++; if (count_only) return;
        if (dot(cross(b - a, c - a), toward) < 0) std::swap(b, c);
        P3 z{0, 0, 0};
        UINT i0 = vert(a, z), i1 = vert(b, z), i2 = vert(c, z);
        tri(i0, i1, i2, mat);
    }
    void quad(P3 a, P3 b, P3 c, P3 d, P3 toward, UINT mat) { flat_tri(a, b, c, toward, mat); flat_tri(a, c, d, toward, mat); }
    // nu x nv grid of quads over origin + s*du + t*dv, optional height function along `toward`
    void grid(P3 o, P3 du, P3 dv, int nu, int nv, P3 toward, const std::function<UINT(int, int)>& mat,
              const std::function<float(float, float)>& bump = nullptr) {
        if (count_only) { tri_count += (size_t)2 * nu * nv; return; }
        P3 nrm = norm(toward);
        auto at = [&](int i, int j) { float s = (float)i / nu, t = (float)j / nv; P3 p = o + du * s + dv * t; if (bump) p = p + nrm * bump(s, t); return p; };
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) quad(at(i, j), at(i + 1, j), at(i + 1, j + 1), at(i, j + 1), toward, mat(i, j));
    }
    void box(P3 lo, P3 hi, UINT mat, int sub = 1) {
        P3 dx{hi.x - lo.x, 0, 0}, dy{0, hi.y - lo.y, 0}, dz{0, 0, hi.z - lo.z};
        auto cm = [mat](int, int) { return mat; };
        grid(lo, dx, dz, sub, sub, {0, -1, 0}, cm); grid(lo + dy, dx, dz, sub, sub, {0, 1, 0}, cm);
        grid(lo, dx, dy, sub, sub, {0, 0, -1}, cm); grid(lo + dz, dx, dy, sub, sub, {0, 0, 1}, cm);
        grid(lo, dz, dy, sub, sub, {-1, 0, 0}, cm); grid(lo + dx, dz, dy, sub, sub, {1, 0, 0}, cm);
    }
    // smooth-shaded surface of revolution about +Y: profile radius r(t), height y(t), t in [0,1]
    void revolve(P3 base, int seg, int rings, const std::function<float(float)>& r, const std::function<float(float)>& y, UINT mat) {
        if (count_only) { tri_count += (size_t)2 * seg * rings; return; }
        std::vector<UINT> idx((size_t)(seg + 1) * (rings + 1));
        for (int j = 0; j <= rings; j++) for (int i = 0; i <= seg; i++) {
            float t = (float)j / rings, a = 6.2831853f * (float)i / seg;
            float rr = r(t), dr = (r(std::min(1.0f, t + 1e-3f)) - r(std::max(0.0f, t - 1e-3f))), dy = (y(std::min(1.0f, t + 1e-3f)) - y(std::max(0.0f, t - 1e-3f)));
            P3 p{base.x + rr * cosf(a), base.y + y(t), base.z + rr * sinf(a)};
            P3 n = norm(P3{dy * cosf(a), -dr, dy * sinf(a)});
            idx[(size_t)j * (seg + 1) + i] = vert(p, n);
        }
        for (int j = 0; j < rings; j++) for (int i = 0; i < seg; i++) {
            UINT a = idx[(size_t)j * (seg + 1) + i], b = idx[(size_t)j * (seg + 1) + i + 1], c = idx[(size_t)(j + 1) * (seg + 1) + i + 1], d = idx[(size_t)(j + 1) * (seg + 1) + i];
            tri(a, c, b, mat); tri(a, d, c, mat); tri_count += 2;
        }
    }
    void sphere(P3 c, float rad, int seg, int rings, UINT mat) {
        revolve({c.x, c.y - rad, c.z}, seg, rings, [rad](float t) { return rad * sinf(3.14159265f * t); }, [rad](float t) { return rad * (1.0f - cosf(3.14159265f * t)); }, mat);
    }
    // n small flat triangles ("leaves") of edge ~size, randomly placed inside the ellipsoid (c, radii r) and randomly oriented: foliage — incoherent, overlapping boxes
    void leaves(P3 c, P3 r, size_t n, float size, UINT mat0, UINT nmat, uint32_t seed);
    // prism with `seg` sides from a to b (a thin bar: 2 seg long thin triangles, no caps)
    void bar(P3 a, P3 b, float rad, int seg, UINT mat) {
        if (count_only) { tri_count += (size_t)2 * seg; return; }
        const P3 ax = norm(b - a), t0 = norm(fabsf(ax.y) < 0.9f ? cross(ax, P3{0, 1, 0}) : cross(ax, P3{1, 0, 0})), t1 = cross(ax, t0);
        for (int i = 0; i < seg; i++) {
            const float a0 = 6.2831853f * (float)i / seg, a1 = 6.2831853f * (float)(i + 1) / seg;
            const P3 d0 = t0 * (rad * cosf(a0)) + t1 * (rad * sinf(a0)), d1 = t0 * (rad * cosf(a1)) + t1 * (rad * sinf(a1));
            quad(a + d0, a + d1, b + d1, b + d0, (d0 + d1) * 0.5f, mat);
        }
    }
};

Material make_mat(float r, float g, float b, float ks, float pr, float pm, float ke_r = 0, float ke_g = 0, float ke_b = 0) {
    Material m(XMFLOAT4(r, g, b, 1.0f), XMFLOAT4(pr, pm, 0.0f, 0.0f));
    m.Ks = XMFLOAT3(ks, ks, ks); m.Ke = XMFLOAT3(ke_r, ke_g, ke_b);
    return m;
}
// LUTs depend on roughness only (ObjLoader.h:351-387 is called with Ks = 1): cache by roughness
void fill_luts(std::vector<Material>& mats) {
    std::vector<std::pair<float, Material>> cache;
    for (size_t i = 1; i < mats.size(); i++) {        // entry 0 is the loader's default material: LUT stays 0 (ObjLoader.h:415-417)
        bool hit = false;
        for (auto& c : cache) if (c.first == mats[i].Pr_Pm_Ps_Pc.x) { memcpy(mats[i].LUT, c.second.LUT, sizeof(mats[i].LUT)); hit = true; break; }
        if (!hit) { GenerateEssLUT(mats[i]); cache.emplace_back(mats[i].Pr_Pm_Ps_Pc.x, mats[i]); }
    }
}
Material default_material() { return Material(XMFLOAT4(1.0f, 1.0f, 1.0f, 1.0f), XMFLOAT4(1.0f, 0.0f, 0.0f, 0.0f)); }   // ObjLoader.h:415

// deterministic hash noise in [0,1)
inline float hash01(uint32_t a, uint32_t b, uint32_t seed) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
void MeshBuilder::leaves(P3 c, P3 r, size_t n, float size, UINT mat0, UINT nmat, uint32_t seed) {
    if (count_only) { tri_count += n; return; }
    for (size_t k = 0; k < n; k++) {
        const uint32_t a = (uint32_t)k, b = (uint32_t)(k >> 32) + 17u;
        // a point in the unit ball by rejection-free warping (cube root radius), scaled to the ellipsoid
        const float u = hash01(a, b, seed), v = hash01(a, b + 1, seed), w = hash01(a, b + 2, seed);
        const float z = 2.0f * u - 1.0f, ph = 6.2831853f * v, rr = cbrtf(w), sxy = sqrtf(std::max(0.0f, 1.0f - z * z));
        const P3 p{c.x + r.x * rr * sxy * cosf(ph), c.y + r.y * rr * z, c.z + r.z * rr * sxy * sinf(ph)};
        P3 e1{hash01(a, b + 3, seed) - 0.5f, hash01(a, b + 4, seed) - 0.5f, hash01(a, b + 5, seed) - 0.5f}, e2{hash01(a, b + 6, seed) - 0.5f, hash01(a, b + 7, seed) - 0.5f, hash01(a, b + 8, seed) - 0.5f};
        e1 = norm(e1) * (size * (0.6f + 0.8f * hash01(a, b + 9, seed))); e2 = norm(e2) * (size * (0.6f + 0.8f * hash01(a, b + 10, seed)));
        flat_tri(p, p + e1, p + e2, cross(e1, e2), mat0 + (UINT)(hash01(a, b + 11, seed) * (float)nmat) % nmat);
    }
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// Cornell Box: the classic data set divided by 555: 5 walls + short and tall block + light
// = 32 triangles, 2 emissive.  Materials as Material records: white .73, red (.65,.05,.05),
// green (.12,.45,.15), light Ke (17,12,4) Kd 0; all Ks = 0, Pr = 1, Pm = 0.
// ------------------------------------------------------------------------------------------------
Scene MakeCornellBox() {
    Scene s; s.name = "cornell";
    s.materials.push_back(default_material());
    s.materials.push_back(make_mat(0.73f, 0.73f, 0.73f, 0.0f, 1.0f, 0.0f));             // 1 white
    s.materials.push_back(make_mat(0.65f, 0.05f, 0.05f, 0.0f, 1.0f, 0.0f));             // 2 red
    s.materials.push_back(make_mat(0.12f, 0.45f, 0.15f, 0.0f, 1.0f, 0.0f));             // 3 green
    s.materials.push_back(make_mat(0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 17.0f, 12.0f, 4.0f));   // 4 light
    fill_luts(s.materials);
    MeshBuilder mb; mb.mat_base = 0; mb.normal_w = 0.0f;
    const float k = 1.0f / 555.0f;
    auto P = [k](float x, float y, float z) { return P3{x * k, y * k, z * k}; };
    const P3 room_c = P(278, 274, 280);
    auto wall = [&](P3 a, P3 b, P3 c, P3 d, UINT mat) { P3 ctr = (a + b + c + d) * 0.25f; mb.quad(a, b, c, d, room_c - ctr, mat); };
    wall(P(552.8f, 0, 0), P(0, 0, 0), P(0, 0, 559.2f), P(549.6f, 0, 559.2f), 1);                         // floor
    wall(P(556, 548.8f, 0), P(556, 548.8f, 559.2f), P(0, 548.8f, 559.2f), P(0, 548.8f, 0), 1);           // ceiling
    wall(P(549.6f, 0, 559.2f), P(0, 0, 559.2f), P(0, 548.8f, 559.2f), P(556, 548.8f, 559.2f), 1);        // back
    wall(P(0, 0, 559.2f), P(0, 0, 0), P(0, 548.8f, 0), P(0, 548.8f, 559.2f), 3);                         // right (green)
    wall(P(552.8f, 0, 0), P(549.6f, 0, 559.2f), P(556, 548.8f, 559.2f), P(556, 548.8f, 0), 2);           // left (red)
    mb.quad(P(343, 548.3f, 227), P(343, 548.3f, 332), P(213, 548.3f, 332), P(213, 548.3f, 227), {0, -1, 0}, 4);   // light, facing down
    auto block = [&](const float q[5][4][3]) {
        P3 c{0, 0, 0};
        for (int f = 0; f < 5; f++) for (int v = 0; v < 4; v++) c = c + P(q[f][v][0], q[f][v][1], q[f][v][2]);
        c = c * (1.0f / 20.0f);
        for (int f = 0; f < 5; f++) {
            P3 a = P(q[f][0][0], q[f][0][1], q[f][0][2]), b = P(q[f][1][0], q[f][1][1], q[f][1][2]), cc = P(q[f][2][0], q[f][2][1], q[f][2][2]), d = P(q[f][3][0], q[f][3][1], q[f][3][2]);
            mb.quad(a, b, cc, d, (a + b + cc + d) * 0.25f - c, 1);
        }
    };
    const float shortb[5][4][3] = {
        {{130, 165, 65}, {82, 165, 225}, {240, 165, 272}, {290, 165, 114}}, {{290, 0, 114}, {290, 165, 114}, {240, 165, 272}, {240, 0, 272}},
        {{130, 0, 65}, {130, 165, 65}, {290, 165, 114}, {290, 0, 114}},     {{82, 0, 225}, {82, 165, 225}, {130, 165, 65}, {130, 0, 65}},
        {{240, 0, 272}, {240, 165, 272}, {82, 165, 225}, {82, 0, 225}}};
    const float tallb[5][4][3] = {
        {{423, 330, 247}, {265, 330, 296}, {314, 330, 456}, {472, 330, 406}}, {{423, 0, 247}, {423, 330, 247}, {472, 330, 406}, {472, 0, 406}},
        {{472, 0, 406}, {472, 330, 406}, {314, 330, 456}, {314, 0, 456}},     {{314, 0, 456}, {314, 330, 456}, {265, 330, 296}, {265, 0, 296}},
        {{265, 0, 296}, {265, 330, 296}, {423, 330, 247}, {423, 0, 247}}};
    block(shortb); block(tallb);
    s.models.push_back(std::move(mb.m));
    s.instances.push_back({0, XMMatrixIdentity()});
    s.eye = XMFLOAT3(278 * k, 273 * k, -475 * k); s.center = XMFLOAT3(278 * k, 273 * k, 0); s.up = XMFLOAT3(0, 1, 0);
    return s;
}

// ------------------------------------------------------------------------------------------------
// Sponza-class atrium (C3/C4): colonnades, arches, drapes, tiled floor, gallery, one emissive sky quad.
// `detail` scales every tessellation; the floor tiling absorbs the remainder so the count lands on target.
// ------------------------------------------------------------------------------------------------
static void sponza_build(MeshBuilder& mb, float detail, int floor_n, uint32_t seed) {
    auto D = [detail](float base) { int v = (int)lroundf(base * detail); return v < 2 ? 2 : v; };
    const float L = 2.0f, Wd = 0.9f, Hh = 1.5f;                         // half length, half width, height
    // floor: checker of two stone materials with a shallow deterministic relief
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, 0, 2 * Wd}, floor_n * 2, floor_n, {0, 1, 0},
            [](int i, int j) { return (UINT)(1 + ((i / 4 + j / 4) & 1)); },
            [seed, floor_n](float s, float t) { return 0.004f * hash01((uint32_t)(s * floor_n * 2), (uint32_t)(t * floor_n), seed); });
    // walls + ceiling (ceiling has the sky opening: 4 strips around it)
    auto m3 = [](int, int) { return (UINT)3; };
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, Hh, 0}, D(48), D(20), {0, 0, 1}, m3);
    mb.grid({-L, 0, Wd}, {2 * L, 0, 0}, {0, Hh, 0}, D(48), D(20), {0, 0, -1}, m3);
    mb.grid({-L, 0, -Wd}, {0, 0, 2 * Wd}, {0, Hh, 0}, D(24), D(20), {1, 0, 0}, m3);
    mb.grid({L, 0, -Wd}, {0, 0, 2 * Wd}, {0, Hh, 0}, D(24), D(20), {-1, 0, 0}, m3);
    const float ox = 1.2f, oz = 0.35f;
    auto m4 = [](int, int) { return (UINT)4; };
    mb.grid({-L, Hh, -Wd}, {L - ox, 0, 0}, {0, 0, 2 * Wd}, D(10), D(16), {0, -1, 0}, m4);
    mb.grid({ox, Hh, -Wd}, {L - ox, 0, 0}, {0, 0, 2 * Wd}, D(10), D(16), {0, -1, 0}, m4);
    mb.grid({-ox, Hh, -Wd}, {2 * ox, 0, 0}, {0, 0, Wd - oz}, D(24), D(6), {0, -1, 0}, m4);
    mb.grid({-ox, Hh, oz}, {2 * ox, 0, 0}, {0, 0, Wd - oz}, D(24), D(6), {0, -1, 0}, m4);
    // sky opening: one emissive quad above it (material 12)
    mb.quad({-ox, Hh + 0.05f, -oz}, {ox, Hh + 0.05f, -oz}, {ox, Hh + 0.05f, oz}, {-ox, Hh + 0.05f, oz}, {0, -1, 0}, 12);
    mb.grid({-ox, Hh, -oz}, {2 * ox, 0, 0}, {0, 0.05f, 0}, D(12), 1, {0, 0, 1}, m4);
    mb.grid({-ox, Hh, oz}, {2 * ox, 0, 0}, {0, 0.05f, 0}, D(12), 1, {0, 0, -1}, m4);
    mb.grid({-ox, Hh, -oz}, {0, 0, 2 * oz}, {0, 0.05f, 0}, D(6), 1, {1, 0, 0}, m4);
    mb.grid({ox, Hh, -oz}, {0, 0, 2 * oz}, {0, 0.05f, 0}, D(6), 1, {-1, 0, 0}, m4);
    // two colonnades of 8 columns with bases, capitals and arches between neighbours
    const int ncol = 8;
    for (int side = 0; side < 2; side++) {
        const float z = side ? 0.5f : -0.5f;
        for (int c = 0; c < ncol; c++) {
            const float x = -L + 0.25f + (2 * L - 0.5f) * (float)c / (ncol - 1);
            mb.box({x - 0.07f, 0, z - 0.07f}, {x + 0.07f, 0.06f, z + 0.07f}, 5, D(2));
            mb.revolve({x, 0.06f, z}, D(24), D(32), [](float t) { return 0.05f - 0.008f * t + 0.004f * sinf(40.0f * t); }, [](float t) { return 0.74f * t; }, 6);
            mb.box({x - 0.075f, 0.80f, z - 0.075f}, {x + 0.075f, 0.86f, z + 0.075f}, 5, D(2));
            mb.sphere({x, 0.90f, z}, 0.035f, D(12), D(8), 7);
            if (c + 1 < ncol) {        // arch: half annulus in the XY plane, extruded in Z
                const float x1 = -L + 0.25f + (2 * L - 0.5f) * (float)(c + 1) / (ncol - 1);
                const float cx = 0.5f * (x + x1), r0 = 0.5f * (x1 - x) - 0.07f, r1 = r0 + 0.05f;
                const int na = D(20);
                for (int a = 0; a < na; a++) {
                    const float a0 = 3.14159265f * (float)a / na, a1 = 3.14159265f * (float)(a + 1) / na;
                    auto pt = [&](float ang, float r, float dz) { return P3{cx + r * cosf(ang), 0.86f + r * sinf(ang), z + dz}; };
                    mb.quad(pt(a0, r0, -0.05f), pt(a1, r0, -0.05f), pt(a1, r0, 0.05f), pt(a0, r0, 0.05f), P3{cx, 0.86f, z} - pt(0.5f * (a0 + a1), r0, 0), 8);
                    mb.quad(pt(a0, r1, -0.05f), pt(a1, r1, -0.05f), pt(a1, r1, 0.05f), pt(a0, r1, 0.05f), pt(0.5f * (a0 + a1), r1, 0) - P3{cx, 0.86f, z}, 8);
                    mb.quad(pt(a0, r0, -0.05f), pt(a1, r0, -0.05f), pt(a1, r1, -0.05f), pt(a0, r1, -0.05f), {0, 0, -1}, 8);
                    mb.quad(pt(a0, r0, 0.05f), pt(a1, r0, 0.05f), pt(a1, r1, 0.05f), pt(a0, r1, 0.05f), {0, 0, 1}, 8);
                }
            }
        }
        // gallery slab above the colonnade
        mb.box({-L, 1.05f, side ? 0.42f : -Wd}, {L, 1.10f, side ? Wd : -0.42f}, 4, D(6));
    }
    // drapes: hanging sine cloths in three colours
    for (int k = 0; k < 6; k++) {
        const float x = -1.5f + 0.6f * k; const UINT mat = 9 + (k % 3);
        mb.grid({x, 1.04f, -0.42f}, {0.4f, 0, 0}, {0, -0.55f, 0}, D(40), D(48), {0, 0, 1}, [mat](int, int) { return mat; },
                [k](float s, float t) { return 0.03f * sinf(25.0f * s + k) * (0.3f + t) + 0.01f * sinf(60.0f * t); });
    }
}

// ------------------------------------------------------------------------------------------------
// The HARD variant of the atrium (round 5).  The scene above is tessellated uniformly — every wall is a grid of centimetre quads — and on such input a
// median-quality tree is already good: sweep SAH, spatial splits and re-insertion change nothing (profiles/r04_bvh_lab.md), while on the reference's real model
// (garage.obj + monke.obj) the same knobs take 11 % each.  The real Sponza is of the second kind: walls, floors and the gallery are a few triangles METRES long, beside
// ornament tessellated to millimetres (lion heads, capitals, vases and plants), long thin trims (flutes, cornices, balusters) and curtains hanging in overlapping layers.
// This variant has the same shell, materials, light, camera and triangle budget, but that distribution: size ratio > 1000 : 1 between the largest and the smallest
// triangle, > 100 : 1 aspect ratios, overlapping cloth.  `leaf_n` (foliage triangles of the potted plants) absorbs the remainder of the budget.
// ------------------------------------------------------------------------------------------------
static void sponza_hard_build(MeshBuilder& mb, float detail, size_t leaf_n, uint32_t seed) {
    auto D = [detail](float base) { int v = (int)lroundf(base * detail); return v < 2 ? 2 : v; };
    const float L = 2.0f, Wd = 0.9f, Hh = 1.5f;
    // shell: a handful of triangles metres long
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, 0, 2 * Wd}, 4, 2, {0, 1, 0}, [](int i, int j) { return (UINT)(1 + ((i + j) & 1)); });
    auto m3 = [](int, int) { return (UINT)3; };
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, Hh, 0}, 2, 1, {0, 0, 1}, m3);
    mb.grid({-L, 0, Wd}, {2 * L, 0, 0}, {0, Hh, 0}, 2, 1, {0, 0, -1}, m3);
    mb.grid({-L, 0, -Wd}, {0, 0, 2 * Wd}, {0, Hh, 0}, 1, 1, {1, 0, 0}, m3);
    mb.grid({L, 0, -Wd}, {0, 0, 2 * Wd}, {0, Hh, 0}, 1, 1, {-1, 0, 0}, m3);
    const float ox = 1.2f, oz = 0.35f;
    auto m4 = [](int, int) { return (UINT)4; };
    mb.grid({-L, Hh, -Wd}, {L - ox, 0, 0}, {0, 0, 2 * Wd}, 1, 1, {0, -1, 0}, m4);
    mb.grid({ox, Hh, -Wd}, {L - ox, 0, 0}, {0, 0, 2 * Wd}, 1, 1, {0, -1, 0}, m4);
    mb.grid({-ox, Hh, -Wd}, {2 * ox, 0, 0}, {0, 0, Wd - oz}, 1, 1, {0, -1, 0}, m4);
    mb.grid({-ox, Hh, oz}, {2 * ox, 0, 0}, {0, 0, Wd - oz}, 1, 1, {0, -1, 0}, m4);
    mb.quad({-ox, Hh + 0.05f, -oz}, {ox, Hh + 0.05f, -oz}, {ox, Hh + 0.05f, oz}, {-ox, Hh + 0.05f, oz}, {0, -1, 0}, 12);          // the sky quad: the scene's one light
    mb.grid({-ox, Hh, -oz}, {2 * ox, 0, 0}, {0, 0.05f, 0}, 1, 1, {0, 0, 1}, m4);
    mb.grid({-ox, Hh, oz}, {2 * ox, 0, 0}, {0, 0.05f, 0}, 1, 1, {0, 0, -1}, m4);
    mb.grid({-ox, Hh, -oz}, {0, 0, 2 * oz}, {0, 0.05f, 0}, 1, 1, {1, 0, 0}, m4);
    mb.grid({ox, Hh, -oz}, {0, 0, 2 * oz}, {0, 0.05f, 0}, 1, 1, {-1, 0, 0}, m4);
    // cornices and ledges: boxes as long as the hall, 2-3 cm thick (12 triangles each, aspect > 100 : 1)
    for (int side = 0; side < 2; side++) {
        const float zw = side ? Wd : -Wd, zi = side ? -1.0f : 1.0f;
        for (int k = 0; k < 5; k++) {
            const float y = 0.30f + 0.27f * k, th = 0.012f + 0.004f * k;
            mb.box({-L, y, std::min(zw, zw + zi * (0.02f + 0.01f * k))}, {L, y + th, std::max(zw, zw + zi * (0.02f + 0.01f * k))}, 5, 1);
        }
    }
    // two colonnades: FLUTED shafts (many sides, one or two rings: triangles 0.4-0.7 m long and millimetres wide), dense capitals, coarse bases
    const int ncol = 8;
    for (int side = 0; side < 2; side++) {
        const float z = side ? 0.5f : -0.5f;
        for (int c = 0; c < ncol; c++) {
            const float x = -L + 0.25f + (2 * L - 0.5f) * (float)c / (ncol - 1);
            mb.box({x - 0.07f, 0, z - 0.07f}, {x + 0.07f, 0.06f, z + 0.07f}, 5, 1);
            mb.revolve({x, 0.06f, z}, D(64), 2, [](float t) { return 0.05f - 0.008f * t; }, [](float t) { return 0.74f * t; }, 6);
            for (int f = 0; f < 12; f++) {                                                    // applied flutes: thin bars standing proud of the shaft
                const float a = 6.2831853f * (float)f / 12.0f;
                mb.bar({x + 0.047f * cosf(a), 0.08f, z + 0.047f * sinf(a)}, {x + 0.043f * cosf(a), 0.78f, z + 0.043f * sinf(a)}, 0.003f, 4, 6);
            }
            mb.box({x - 0.075f, 0.80f, z - 0.075f}, {x + 0.075f, 0.86f, z + 0.075f}, 5, 1);
            mb.revolve({x, 0.86f, z}, D(40), D(24), [](float t) { return 0.05f + 0.03f * sinf(3.14159265f * t) + 0.004f * sinf(37.0f * t); }, [](float t) { return 0.05f * t; }, 7);   // carved capital
            mb.sphere({x, 0.935f, z}, 0.03f, D(24), D(16), 7);
            if (c + 1 < ncol) {
                const float x1 = -L + 0.25f + (2 * L - 0.5f) * (float)(c + 1) / (ncol - 1);
                const float cx = 0.5f * (x + x1), r0 = 0.5f * (x1 - x) - 0.07f, r1 = r0 + 0.05f;
                const int na = 12;
                for (int a = 0; a < na; a++) {
                    const float a0 = 3.14159265f * (float)a / na, a1 = 3.14159265f * (float)(a + 1) / na;
                    auto pt = [&](float ang, float r, float dz) { return P3{cx + r * cosf(ang), 0.86f + r * sinf(ang), z + dz}; };
                    mb.quad(pt(a0, r0, -0.05f), pt(a1, r0, -0.05f), pt(a1, r0, 0.05f), pt(a0, r0, 0.05f), P3{cx, 0.86f, z} - pt(0.5f * (a0 + a1), r0, 0), 8);
                    mb.quad(pt(a0, r1, -0.05f), pt(a1, r1, -0.05f), pt(a1, r1, 0.05f), pt(a0, r1, 0.05f), pt(0.5f * (a0 + a1), r1, 0) - P3{cx, 0.86f, z}, 8);
                    mb.quad(pt(a0, r0, -0.05f), pt(a1, r0, -0.05f), pt(a1, r1, -0.05f), pt(a0, r1, -0.05f), {0, 0, -1}, 8);
                    mb.quad(pt(a0, r0, 0.05f), pt(a1, r0, 0.05f), pt(a1, r1, 0.05f), pt(a0, r1, 0.05f), {0, 0, 1}, 8);
                }
            }
        }
        // gallery slab (two big boxes) and its balustrade: a rail as long as the hall on ~130 thin balusters
        mb.box({-L, 1.05f, side ? 0.42f : -Wd}, {L, 1.10f, side ? Wd : -0.42f}, 4, 1);
        const float zr = side ? 0.44f : -0.44f;
        mb.box({-L, 1.24f, zr - 0.01f}, {L, 1.26f, zr + 0.01f}, 5, 1);
        for (int k = 0; k < 130; k++) { const float x = -L + 0.02f + (2 * L - 0.04f) * (float)k / 129.0f; mb.bar({x, 1.10f, zr}, {x, 1.24f, zr}, 0.004f, 6, 5); }
    }
    // "lion heads": small patches of deeply carved relief on the side walls, tessellated to ~2 mm
    for (int side = 0; side < 2; side++) for (int k = 0; k < 6; k++) {
        const float x = -1.6f + 0.64f * k, zw = side ? Wd - 0.001f : -Wd + 0.001f, zi = side ? -1.0f : 1.0f;
        mb.grid({x - 0.09f, 0.55f, zw}, {0.18f, 0, 0}, {0, 0.18f, 0}, D(56), D(56), {0, 0, zi}, [](int, int) { return (UINT)8; },
                [k, side, seed](float u, float v) { const float r2 = (u - 0.5f) * (u - 0.5f) + (v - 0.5f) * (v - 0.5f);
                                                    return (0.035f * expf(-14.0f * r2) + 0.006f * sinf(31.0f * u + k) * sinf(27.0f * v + side)) * (r2 < 0.24f ? 1.0f : 0.0f) + 0.0015f * hash01((uint32_t)(u * 977), (uint32_t)(v * 991), seed + k); });
    }
    // drapes: three OVERLAPPING layers of fine cloth a few millimetres apart, six hangings
    for (int k = 0; k < 6; k++) for (int layer = 0; layer < 3; layer++) {
        const float x = -1.5f + 0.6f * k - 0.01f * layer; const UINT mat = 9 + ((k + layer) % 3);
        mb.grid({x, 1.04f, -0.42f + 0.004f * layer}, {0.4f + 0.02f * layer, 0, 0}, {0, -0.55f - 0.03f * layer, 0}, D(36), D(44), {0, 0, 1}, [mat](int, int) { return mat; },
                [k, layer](float u, float v) { return 0.03f * sinf(25.0f * u + k + 1.7f * layer) * (0.3f + v) + 0.01f * sinf(60.0f * v + layer); });
    }
    // vases with plants along the nave: dense pottery, and foliage — thousands of centimetre leaves in overlapping clouds (the remainder of the triangle budget)
    const int nvase = 10;
    for (int v = 0; v < nvase; v++) {
        const float x = -1.7f + 3.4f * (float)v / (nvase - 1), z = (v & 1) ? 0.28f : -0.28f;
        mb.revolve({x, 0, z}, D(40), D(36), [](float t) { return 0.035f + 0.03f * sinf(3.14159265f * (0.15f + 0.8f * t)) + 0.002f * sinf(50.0f * t); }, [](float t) { return 0.16f * t; }, 7);
        const size_t n0 = leaf_n * (size_t)v / nvase, n1 = leaf_n * (size_t)(v + 1) / nvase;
        mb.leaves({x, 0.32f, z}, {0.10f, 0.16f, 0.10f}, n1 - n0, 0.006f, 10, 2, seed + 31u * (uint32_t)v);
    }
}

Scene MakeSponzaClass(uint32_t target, uint32_t seed, bool hard) {
    if (hard) {
        Scene s; s.name = "sponza_class_hard";
        s.materials.push_back(default_material());
        const float c[12][3] = {{.62f, .58f, .50f}, {.42f, .40f, .36f}, {.70f, .62f, .50f}, {.55f, .52f, .48f}, {.50f, .46f, .40f}, {.66f, .62f, .55f},
                                {.58f, .50f, .38f}, {.60f, .56f, .50f}, {.62f, .10f, .10f}, {.10f, .28f, .55f}, {.12f, .45f, .18f}, {0, 0, 0}};
        for (int i = 0; i < 12; i++) s.materials.push_back(i == 11 ? make_mat(0, 0, 0, 0, 1, 0, 24.0f, 22.0f, 18.0f) : make_mat(c[i][0], c[i][1], c[i][2], 0.0f, 1.0f, 0.0f));
        fill_luts(s.materials);
        auto count = [&](float detail, size_t leaf_n) { MeshBuilder mb; mb.count_only = true; mb.mat_base = 0; mb.normal_w = 0; sponza_hard_build(mb, detail, leaf_n, seed); return mb.tri_count; };
        float lo = 0.05f, hi = 8.0f;                      // ornament, cloth and pottery take ~70 % of the budget, the foliage the rest
        for (int it = 0; it < 40; it++) { float mid = 0.5f * (lo + hi); if (count(mid, 0) < (size_t)(0.70 * target)) lo = mid; else hi = mid; }
        const size_t rest = count(lo, 0);
        MeshBuilder mb; mb.mat_base = 0; mb.normal_w = 0.0f;
        sponza_hard_build(mb, lo, target > rest ? target - rest : 0, seed);
        s.models.push_back(std::move(mb.m));
        s.instances.push_back({0, XMMatrixIdentity()});
        s.eye = XMFLOAT3(-1.8f, 0.45f, 0.0f); s.center = XMFLOAT3(0.5f, 0.55f, 0.0f); s.up = XMFLOAT3(0, 1, 0);
        return s;
    }
    Scene s; s.name = "sponza_class";
    s.materials.push_back(default_material());
    const float c[12][3] = {{.62f, .58f, .50f}, {.42f, .40f, .36f}, {.70f, .62f, .50f}, {.55f, .52f, .48f}, {.50f, .46f, .40f}, {.66f, .62f, .55f},
                            {.58f, .50f, .38f}, {.60f, .56f, .50f}, {.62f, .10f, .10f}, {.10f, .28f, .55f}, {.12f, .45f, .18f}, {0, 0, 0}};
    for (int i = 0; i < 12; i++) s.materials.push_back(i == 11 ? make_mat(0, 0, 0, 0, 1, 0, 24.0f, 22.0f, 18.0f) : make_mat(c[i][0], c[i][1], c[i][2], 0.0f, 1.0f, 0.0f));
    fill_luts(s.materials);
    // pick detail so that everything but the floor uses ~85 % of the budget, then size the floor for the rest
    auto count = [&](float detail, int floor_n) { MeshBuilder mb; mb.count_only = true; mb.mat_base = 0; mb.normal_w = 0; sponza_build(mb, detail, floor_n, seed); return mb.tri_count; };
    float lo = 0.05f, hi = 8.0f;
    for (int it = 0; it < 40; it++) { float mid = 0.5f * (lo + hi); if (count(mid, 1) < (size_t)(0.85 * target)) lo = mid; else hi = mid; }
    const float detail = lo;
    const size_t rest = count(detail, 1) - 4;
    int floor_n = (int)lroundf(sqrtf((float)(target > rest ? target - rest : 4) / 4.0f));
    if (floor_n < 1) floor_n = 1;
    MeshBuilder mb; mb.mat_base = 0; mb.normal_w = 0.0f;
    sponza_build(mb, detail, floor_n, seed);
    s.models.push_back(std::move(mb.m));
    s.instances.push_back({0, XMMatrixIdentity()});
    s.eye = XMFLOAT3(-1.8f, 0.45f, 0.0f); s.center = XMFLOAT3(0.5f, 0.55f, 0.0f); s.up = XMFLOAT3(0, 1, 0);
    return s;
}

// ------------------------------------------------------------------------------------------------
// Bistro-class street (C5): two rows of facades (window grids with frames and balconies), cobbled street,
// lamp posts with small emissive panels (~200 light triangles), ~40 materials of which ~30 % GGX.
// ------------------------------------------------------------------------------------------------
static void bistro_build(MeshBuilder& mb, float detail, int street_n, uint32_t seed) {
    auto D = [detail](float base) { int v = (int)lroundf(base * detail); return v < 1 ? 1 : v; };
    const float L = 3.0f, Wd = 0.6f;
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, 0, 2 * Wd}, street_n * 5, street_n, {0, 1, 0},
            [seed](int i, int j) { return (UINT)(1 + (uint32_t)(hash01(i / 3, j / 3, seed) * 3.0f)); },
            [seed, street_n](float s, float t) { return 0.006f * hash01((uint32_t)(s * street_n * 5), (uint32_t)(t * street_n), seed + 1); });
    const int nb = 12;
    for (int side = 0; side < 2; side++) {
        const float z0 = side ? Wd : -Wd, zn = side ? -1.0f : 1.0f;
        for (int b = 0; b < nb; b++) {
            const float x0 = -L + 2 * L * (float)b / nb, x1 = -L + 2 * L * (float)(b + 1) / nb;
            const float h = 0.9f + 0.6f * hash01(b, side, seed + 2);
            const UINT wall_mat = 4 + (UINT)(hash01(b, side, seed + 3) * 8.0f);            // 4..11 diffuse plaster
            mb.grid({x0, 0, z0}, {x1 - x0, 0, 0}, {0, h, 0}, D(24), D(36), {0, 0, zn}, [wall_mat](int, int) { return wall_mat; },
                    [b, side, seed](float s, float t) { return 0.004f * hash01((uint32_t)(s * 97) + b * 131, (uint32_t)(t * 89) + side * 17, seed + 4); });
            mb.grid({x0, h, z0}, {x1 - x0, 0, 0}, {0, 0, -zn * 0.5f}, D(8), D(6), {0, 1, 0}, [](int, int) { return (UINT)12; });
            const int floors = (int)(h / 0.3f), wins = 4;
            for (int f = 0; f < floors; f++) for (int w = 0; w < wins; w++) {
                const float wx = x0 + (x1 - x0) * ((float)w + 0.5f) / wins, wy = 0.12f + 0.3f * f;
                const float zz = z0 + zn * 0.012f;
                const UINT glass = 28 + (UINT)(hash01(b * 16 + w, f * 2 + side, seed + 5) * 4.0f);   // 28..31 smooth GGX "panes"
                const UINT frame = 20 + (UINT)(hash01(b, w, seed + 6) * 8.0f);                        // 20..27 GGX metals
                mb.grid({wx - 0.04f, wy, zz}, {0.08f, 0, 0}, {0, 0.14f, 0}, D(3), D(4), {0, 0, zn}, [glass](int, int) { return glass; });
                mb.box({wx - 0.05f, wy - 0.01f, zz - 0.004f}, {wx + 0.05f, wy, zz + 0.02f}, frame, D(2));
                mb.box({wx - 0.05f, wy + 0.14f, zz - 0.004f}, {wx + 0.05f, wy + 0.15f, zz + 0.02f}, frame, D(2));
                if (f > 0 && ((b + w + f) & 1)) {     // balcony rail: small spheres on a bar
                    mb.box({wx - 0.06f, wy - 0.02f, z0 + zn * 0.01f}, {wx + 0.06f, wy - 0.012f, z0 + zn * 0.07f}, 13, D(2));
                    for (int k = 0; k < 3; k++) mb.sphere({wx - 0.045f + 0.045f * k, wy + 0.02f, z0 + zn * 0.065f}, 0.008f, D(8), D(6), frame);
                }
            }
            // awning over the ground floor of every other building
            if (b & 1) mb.grid({x0 + 0.03f, 0.30f, z0}, {x1 - x0 - 0.06f, 0, 0}, {0, -0.06f, zn * 0.16f}, D(16), D(10), {0, 1, zn * 0.4f},
                               [b](int i, int) { return (UINT)(14 + ((i / 2 + b) & 1) * 2); }, [](float s, float) { return 0.004f * sinf(50.0f * s); });
        }
    }
    // street lamps: post + emissive box panels (12 emissive triangles each; 17 lamps ~ 204 light triangles)
    for (int l = 0; l < 17; l++) {
        const float x = -L + 0.2f + (2 * L - 0.4f) * (float)l / 16.0f, z = (l & 1) ? 0.45f : -0.45f;
        mb.revolve({x, 0, z}, D(10), D(16), [](float t) { return 0.012f - 0.004f * t; }, [](float t) { return 0.5f * t; }, 24);
        mb.box({x - 0.02f, 0.50f, z - 0.02f}, {x + 0.02f, 0.54f, z + 0.02f}, 39, 1);
        mb.sphere({x, 0.56f, z}, 0.012f, D(8), D(6), 24);
    }
    // tables / chairs / barrels along the pavement: spheres, cylinders, boxes
    for (int k = 0; k < 60; k++) {
        const float x = -L + 0.1f + (2 * L - 0.2f) * hash01(k, 1, seed + 7), z = (k & 1 ? 1.0f : -1.0f) * (0.30f + 0.1f * hash01(k, 2, seed + 8));
        const UINT mat = 32 + (UINT)(hash01(k, 3, seed + 9) * 7.0f);                        // 32..38 mixed
        mb.revolve({x, 0, z}, D(16), D(12), [](float t) { return 0.035f * (1.0f + 0.25f * sinf(3.14159265f * t)); }, [](float t) { return 0.09f * t; }, mat);
        mb.box({x - 0.04f, 0.09f, z - 0.04f}, {x + 0.04f, 0.10f, z + 0.04f}, mat, D(3));
    }
}

// The HARD variant of the street (see sponza_hard_build): facades and paving as a few large triangles, window frames / mullions / cables / balusters as long thin
// ones, carved cornices and cafe furniture tessellated to millimetres, awnings as fine cloth, and street trees whose foliage — centimetre leaves in overlapping
// clouds — takes the remainder of the 3.8 M-triangle budget (in the real Bistro the vegetation is the largest single part).  Same materials, lamps and camera.
static void bistro_hard_build(MeshBuilder& mb, float detail, size_t leaf_n, uint32_t seed) {
    auto D = [detail](float base) { int v = (int)lroundf(base * detail); return v < 1 ? 1 : v; };
    const float L = 3.0f, Wd = 0.6f;
    mb.grid({-L, 0, -Wd}, {2 * L, 0, 0}, {0, 0, 2 * Wd}, 3, 1, {0, 1, 0}, [seed](int i, int j) { return (UINT)(1 + (uint32_t)(hash01(i, j, seed) * 3.0f)); });      // paving: 6 triangles
    for (int k = 0; k < 8; k++) {              // a few patches of real cobbles (dense relief), flush above the paving
        const float x = -L + 0.3f + (2 * L - 0.9f) * hash01(k, 40, seed), z = -0.25f + 0.5f * hash01(k, 41, seed);
        mb.grid({x, 0.001f, z}, {0.3f, 0, 0}, {0, 0, 0.2f}, D(60), D(40), {0, 1, 0}, [](int i, int j) { return (UINT)(1 + ((i / 3 + j / 3) % 3)); },
                [k, seed](float u, float v) { return 0.004f * hash01((uint32_t)(u * 20) + 31u * k, (uint32_t)(v * 13), seed + 1) + 0.002f * sinf(120.0f * u) * sinf(80.0f * v); });
    }
    const int nb = 12;
    for (int side = 0; side < 2; side++) {
        const float z0 = side ? Wd : -Wd, zn = side ? -1.0f : 1.0f;
        for (int b = 0; b < nb; b++) {
            const float x0 = -L + 2 * L * (float)b / nb, x1 = -L + 2 * L * (float)(b + 1) / nb;
            const float h = 0.9f + 0.6f * hash01(b, side, seed + 2);
            const UINT wall_mat = 4 + (UINT)(hash01(b, side, seed + 3) * 8.0f);
            mb.grid({x0, 0, z0}, {x1 - x0, 0, 0}, {0, h, 0}, 1, 1, {0, 0, zn}, [wall_mat](int, int) { return wall_mat; });                       // the facade: two triangles
            mb.grid({x0, h, z0}, {x1 - x0, 0, 0}, {0, 0, -zn * 0.5f}, 1, 1, {0, 1, 0}, [](int, int) { return (UINT)12; });                      // the roof: two
            // carved cornice under the roof line: a strip of dense relief as long as the building
            mb.grid({x0, h - 0.05f, z0 + zn * 0.004f}, {x1 - x0, 0, 0}, {0, 0.05f, 0}, D(120), D(12), {0, 0, zn}, [](int, int) { return (UINT)13; },
                    [b, side](float u, float v) { return 0.008f * fabsf(sinf(60.0f * u + b)) * sinf(3.14159265f * v) + 0.002f * sinf(200.0f * u + side); });
            const int floors = (int)(h / 0.3f), wins = 4;
            for (int f = 0; f < floors; f++) for (int w = 0; w < wins; w++) {
                const float wx = x0 + (x1 - x0) * ((float)w + 0.5f) / wins, wy = 0.12f + 0.3f * f;
                const float zz = z0 + zn * 0.012f;
                const UINT glass = 28 + (UINT)(hash01(b * 16 + w, f * 2 + side, seed + 5) * 4.0f);
                const UINT frame = 20 + (UINT)(hash01(b, w, seed + 6) * 8.0f);
                mb.grid({wx - 0.04f, wy, zz}, {0.08f, 0, 0}, {0, 0.14f, 0}, 1, 1, {0, 0, zn}, [glass](int, int) { return glass; });             // a pane: two triangles
                mb.box({wx - 0.05f, wy - 0.01f, zz - 0.004f}, {wx + 0.05f, wy, zz + 0.02f}, frame, 1);
                mb.box({wx - 0.05f, wy + 0.14f, zz - 0.004f}, {wx + 0.05f, wy + 0.15f, zz + 0.02f}, frame, 1);
                mb.bar({wx, wy, zz + zn * 0.002f}, {wx, wy + 0.14f, zz + zn * 0.002f}, 0.002f, 4, frame);                                       // mullion and transom: thin bars
                mb.bar({wx - 0.04f, wy + 0.07f, zz + zn * 0.002f}, {wx + 0.04f, wy + 0.07f, zz + zn * 0.002f}, 0.002f, 4, frame);
                if (f > 0 && ((b + w + f) & 1)) {      // balcony: a slab, a rail and nine thin spindles
                    mb.box({wx - 0.06f, wy - 0.02f, std::min(z0 + zn * 0.01f, z0 + zn * 0.07f)}, {wx + 0.06f, wy - 0.012f, std::max(z0 + zn * 0.01f, z0 + zn * 0.07f)}, 13, 1);
                    mb.bar({wx - 0.06f, wy + 0.04f, z0 + zn * 0.068f}, {wx + 0.06f, wy + 0.04f, z0 + zn * 0.068f}, 0.003f, 5, frame);
                    for (int k = 0; k < 9; k++) mb.bar({wx - 0.056f + 0.014f * k, wy - 0.012f, z0 + zn * 0.068f}, {wx - 0.056f + 0.014f * k, wy + 0.04f, z0 + zn * 0.068f}, 0.0015f, 4, frame);
                }
            }
            if (b & 1) mb.grid({x0 + 0.03f, 0.30f, z0}, {x1 - x0 - 0.06f, 0, 0}, {0, -0.06f, zn * 0.16f}, D(60), D(36), {0, 1, zn * 0.4f},               // awning: fine cloth
                               [b](int i, int) { return (UINT)(14 + ((i / 6 + b) & 1) * 2); }, [](float u, float v) { return 0.004f * sinf(50.0f * u) + 0.002f * sinf(90.0f * v); });
        }
    }
    // cables across the street: 1.2 m long, 3 mm thick
    for (int k = 0; k < 40; k++) { const float x = -L + 0.1f + (2 * L - 0.2f) * (float)k / 39.0f, y = 0.62f + 0.2f * hash01(k, 50, seed); mb.bar({x, y, -Wd + 0.01f}, {x + 0.05f, y + 0.03f, Wd - 0.01f}, 0.0015f, 4, 24); }
    for (int l = 0; l < 17; l++) {             // street lamps: 12 emissive triangles each, as in the plain variant
        const float x = -L + 0.2f + (2 * L - 0.4f) * (float)l / 16.0f, z = (l & 1) ? 0.45f : -0.45f;
        mb.bar({x, 0, z}, {x, 0.5f, z}, 0.01f, 8, 24);
        mb.box({x - 0.02f, 0.50f, z - 0.02f}, {x + 0.02f, 0.54f, z + 0.02f}, 39, 1);
        mb.sphere({x, 0.56f, z}, 0.012f, D(8), D(6), 24);
    }
    for (int k = 0; k < 60; k++) {             // cafe furniture: turned table legs and barrels (dense), table tops (coarse), chair legs (thin)
        const float x = -L + 0.1f + (2 * L - 0.2f) * hash01(k, 1, seed + 7), z = (k & 1 ? 1.0f : -1.0f) * (0.30f + 0.1f * hash01(k, 2, seed + 8));
        const UINT mat = 32 + (UINT)(hash01(k, 3, seed + 9) * 7.0f);
        mb.revolve({x, 0, z}, D(28), D(40), [](float t) { return 0.012f + 0.02f * fabsf(sinf(9.0f * t)) * (1.0f - t) + 0.015f * t; }, [](float t) { return 0.09f * t; }, mat);
        mb.box({x - 0.04f, 0.09f, z - 0.04f}, {x + 0.04f, 0.10f, z + 0.04f}, mat, 1);
        for (int q = 0; q < 4; q++) mb.bar({x + 0.07f + 0.02f * (q & 1), 0, z + 0.02f * (q >> 1)}, {x + 0.07f + 0.02f * (q & 1), 0.05f, z + 0.02f * (q >> 1)}, 0.0015f, 4, mat);
    }
    // street trees: a trunk, a few branches, and the foliage that takes the rest of the budget
    const int ntree = 14;
    for (int t = 0; t < ntree; t++) {
        const float x = -L + 0.25f + (2 * L - 0.5f) * (float)t / (ntree - 1), z = (t & 1) ? -0.32f : 0.32f;
        mb.revolve({x, 0, z}, D(12), D(10), [](float u) { return 0.018f - 0.008f * u; }, [](float u) { return 0.45f * u; }, 34);
        for (int q = 0; q < 5; q++) { const float a = 1.2566f * q + 0.3f * t; mb.bar({x, 0.40f, z}, {x + 0.12f * cosf(a), 0.55f + 0.03f * q, z + 0.12f * sinf(a)}, 0.004f, 5, 34); }
        const size_t n0 = leaf_n * (size_t)t / ntree, n1 = leaf_n * (size_t)(t + 1) / ntree;
        mb.leaves({x, 0.62f, z}, {0.20f, 0.16f, 0.18f}, n1 - n0, 0.009f, 15, 3, seed + 131u * (uint32_t)t);
    }
}

Scene MakeBistroClass(uint32_t target, uint32_t seed, bool hard) {
    Scene s; s.name = hard ? "bistro_class_hard" : "bistro_class";
    s.materials.push_back(default_material());
    for (int i = 1; i <= 39; i++) {
        const float r = 0.25f + 0.6f * hash01(i, 1, seed), g = 0.25f + 0.6f * hash01(i, 2, seed), b = 0.25f + 0.6f * hash01(i, 3, seed);
        if (i == 39) s.materials.push_back(make_mat(0, 0, 0, 0, 1, 0, 40.0f, 34.0f, 22.0f));                 // lamps
        else if (i >= 28 && i <= 31) {     // panes: dielectric (F0 0.04, Ni 1.5, dissolve 0.1, smooth GGX interface): they TRANSMIT under RTX_FLAG_TRANSMISSION (strategy 3 is a
            Material m = make_mat(0.05f, 0.05f, 0.05f, 0.04f, 0.10f + 0.02f * (i - 28), 0.0f);   // stub in the reference, BRDF_v6.hlsl:44-47; see csrc/rtx_bsdf.hpp) and are dark, slightly glossy plates without it
            m.Kd.w = 0.1f; m.Ni = 1.5f;
            s.materials.push_back(m);
        }
        else if (i >= 20 && i <= 27) s.materials.push_back(make_mat(r, g, b, 0.6f + 0.3f * hash01(i, 4, seed), 0.1f + 0.5f * hash01(i, 5, seed), (float)(i & 1)));   // GGX, roughness U[0.1,0.6], metallic {0,1}
        else s.materials.push_back(make_mat(r, g, b, 0.04f, 1.0f, 0.0f));
    }
    fill_luts(s.materials);
    if (hard) {
        auto counth = [&](float detail, size_t leaf_n) { MeshBuilder mb; mb.count_only = true; mb.mat_base = 0; mb.normal_w = 0; bistro_hard_build(mb, detail, leaf_n, seed); return mb.tri_count; };
        float lo = 0.05f, hi = 16.0f;                     // carving, cloth and furniture take ~55 % of the budget, the trees the rest
        for (int it = 0; it < 40; it++) { float mid = 0.5f * (lo + hi); if (counth(mid, 0) < (size_t)(0.55 * target)) lo = mid; else hi = mid; }
        const size_t rest = counth(lo, 0);
        MeshBuilder mb; mb.mat_base = 0; mb.normal_w = 0.0f;
        bistro_hard_build(mb, lo, target > rest ? target - rest : 0, seed);
        s.models.push_back(std::move(mb.m));
        s.instances.push_back({0, XMMatrixIdentity()});
        s.eye = XMFLOAT3(-2.7f, 0.35f, 0.05f); s.center = XMFLOAT3(0.0f, 0.45f, 0.0f); s.up = XMFLOAT3(0, 1, 0);
        return s;
    }
    auto count = [&](float detail, int street_n) { MeshBuilder mb; mb.count_only = true; mb.mat_base = 0; mb.normal_w = 0; bistro_build(mb, detail, street_n, seed); return mb.tri_count; };
    float lo = 0.05f, hi = 16.0f;
    for (int it = 0; it < 40; it++) { float mid = 0.5f * (lo + hi); if (count(mid, 1) < (size_t)(0.9 * target)) lo = mid; else hi = mid; }
    const float detail = lo;
    const size_t rest = count(detail, 1) - 10;
    int street_n = (int)lroundf(sqrtf((float)(target > rest ? target - rest : 10) / 10.0f));
    if (street_n < 1) street_n = 1;
    MeshBuilder mb; mb.mat_base = 0; mb.normal_w = 0.0f;
    bistro_build(mb, detail, street_n, seed);
    s.models.push_back(std::move(mb.m));
    s.instances.push_back({0, XMMatrixIdentity()});
    s.eye = XMFLOAT3(-2.7f, 0.35f, 0.05f); s.center = XMFLOAT3(0.0f, 0.45f, 0.0f); s.up = XMFLOAT3(0, 1, 0);
    return s;
}

// the reference's startup scene (Renderer.cpp:363-407, 444-449, 46-48)
Scene LoadObjScene(const std::vector<std::string>& files, const std::string& mtl_dir) {
    Scene s; s.name = "obj";
    UINT materialIDOffset = 0, materialVertexOffset = 0;       // Renderer members of the same name
    size_t total_ids = 0;
    for (size_t i = 0; i < files.size(); i++) {
        SceneModel m;
        ObjLoader::loadObjFileEx(files[i], &m.vertices, &m.indices, &s.materials, &m.materialIDs, &materialIDOffset, &materialVertexOffset, &s.materialExt, &s.textures, mtl_dir);
        total_ids += m.materialIDs.size();
        materialVertexOffset = (UINT)total_ids;                // Renderer.cpp:1997
        s.models.push_back(std::move(m));
        XMMATRIX t = XMMatrixIdentity();
        if (i == 1) t = XMMatrixScaling(1, 1, 1) * XMMatrixRotationAxis({0.f, 1.f, 0.f}, 1.57f) * XMMatrixTranslation(0, 0, 0);   // Renderer.cpp:444-449
        s.instances.push_back({(UINT)i, t});
    }
    s.eye = XMFLOAT3(-1.5f, 1.5f, 3.5f); s.center = XMFLOAT3(0, 1, 0); s.up = XMFLOAT3(0, 1, 0);   // Renderer.cpp:46-48
    return s;
}

void SceneViewProj(const Scene& s, float aspect, float view[16], float proj[16]) {
    nv_helpers_dx12::Manipulator m;
    m.setLookat(s.eye, s.center, s.up);
    memcpy(view, m.getMatrix(), 64);
    XMMATRIX P = XMMatrixPerspectiveFovRH(s.fovY_deg * XM_PI / 180.0f, aspect, s.zn, s.zf);   // Renderer.cpp:1730-1731
    memcpy(proj, P.data(), 64);
}

int UploadScene(const Scene& s, rtx_ctx* ctx, float aspect) {
    int r;
    if ((r = rtx_set_materials(ctx, s.materials.data(), (uint32_t)s.materials.size()))) return r;
    for (const SceneModel& m : s.models) {
        uint32_t id;
        if ((r = rtx_add_mesh(ctx, m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id))) return r;
    }
    for (const SceneInstance& in : s.instances) { uint32_t id; if ((r = rtx_add_instance(ctx, in.model, in.transform.data(), &id))) return r; }
    if ((r = rtx_commit_scene(ctx))) return r;
    float view[16], proj[16];
    SceneViewProj(s, aspect, view, proj);
    return rtx_set_camera(ctx, view, proj);
}
