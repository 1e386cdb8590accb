// rtx_scene_host.cpp — host-side scene assembly: material packing, world-space flattening, shade records,
// emissive-triangle CDF and the binned-SAH BVH build.  No HIP calls in this file.
#include "rtx_scene_host.hpp"
#include "rtx_wide.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <system_error>
#include <thread>
#include <functional>
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace rtx {

// float -> binary16 (round to nearest even) -> float.  `-enable-16bit-types` makes HLSL `half` a true
// binary16 (DXRHelper.h:125), so half4(mat.Kd) etc. round (Sampler_v6.hlsl:71-83).
float half_round(float x) {
    uint32_t u = f2u(x), sign = u & 0x80000000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return x;
    if (a >= 0x477FF000u) return u2f(sign | 0x7F800000u);
    if (a < 0x33000001u) return u2f(sign);
    if (a < 0x38800000u) {
        float r = nearbyintf(u2f(a) * 16777216.0f);
        return u2f(sign | f2u(r * (1.0f / 16777216.0f)));
    }
    uint32_t rem = a & 0x1FFFu, base = a & ~0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
    return u2f(sign | base);
}

// General 4x4 inverse by cofactors in double, rounded to float once (XMMatrixInverse stand-in:
// Renderer.cpp:1735-1736, 2101-2118).
void mat4_inverse(const float* mf, float* out) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = (double)mf[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double id = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * id);
}

// objectToWorldNormal = transpose(inverse(upper 3x3, rest identity)): Renderer.cpp:2104-2116
void normal_matrix(const float* o2w, float* out) {
    float u[16], inv[16];
    memcpy(u, o2w, 64);
    u[3] = u[7] = u[11] = 0.0f; u[12] = u[13] = u[14] = 0.0f; u[15] = 1.0f;
    mat4_inverse(u, inv);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[c * 4 + r] = inv[r * 4 + c];
}

bool SceneHost::set_materials(const void* mats, uint32_t count) {
    if (!mats && count) { err = "materials pointer is null"; return false; }
    mats128.assign((const float*)mats, (const float*)mats + (size_t)count * 32);
    mats_dirty = true;
    return true;
}

bool SceneHost::add_mesh(const void* verts28, uint32_t nverts, const uint32_t* idx, uint32_t nidx, const uint32_t* mids, uint32_t* out) {
    if (!verts28 || !idx || !mids) { err = "add_mesh: null array"; return false; }
    if (nidx % 3) { err = "add_mesh: index count is not a multiple of 3"; return false; }
    const float* v = (const float*)verts28;
    for (uint32_t i = 0; i < nidx; i++) if (idx[i] >= nverts) { err = "add_mesh: index out of range"; return false; }
    // Vertex.normal.w is the model's base offset inside the global materialIDs[] (ObjLoader.h:466; Hit_v6.hlsl:17)
    for (uint32_t i = 0; i < nverts; i++)
        if ((uint32_t)v[(size_t)i * 7 + 6] != (uint32_t)matids.size()) { err = "add_mesh: Vertex.normal.w != materialIDs base offset of this mesh"; return false; }
    MeshHost m;
    m.verts.assign(v, v + (size_t)nverts * 7);
    m.idx.assign(idx, idx + nidx);
    m.matid_base = (uint32_t)matids.size();
    matids.insert(matids.end(), mids, mids + nidx);
    if (out) *out = (uint32_t)meshes.size();
    meshes.push_back(std::move(m));
    topo_dirty = true;
    return true;
}

bool SceneHost::add_instance(uint32_t mesh, const float* o2w, uint32_t* out) {
    if (mesh >= meshes.size()) { err = "add_instance: unknown mesh"; return false; }
    InstHost in; in.mesh = mesh; memcpy(in.o2w, o2w, 64); normal_matrix(o2w, in.nrm); in.tri_base = 0;
    mat4_inverse(o2w, in.o2w_inv); memcpy(in.prev_o2w, o2w, 64);          // Renderer.cpp:2098-2102
    if (out) *out = (uint32_t)insts.size();
    insts.push_back(in);
    topo_dirty = true;
    return true;
}

bool SceneHost::set_instance_transform(uint32_t inst, const float* o2w) {
    if (inst >= insts.size()) { err = "set_instance_transform: unknown instance"; return false; }
    memcpy(insts[inst].prev_o2w, insts[inst].o2w, 64);                        // prevObjectToWorld = last frame's objectToWorld
    memcpy(insts[inst].o2w, o2w, 64); normal_matrix(o2w, insts[inst].nrm); mat4_inverse(o2w, insts[inst].o2w_inv);
    return true;
}

static inline f3 vpos(const MeshHost& m, uint32_t vi) { const float* p = &m.verts[(size_t)vi * 7]; return mk3(p[0], p[1], p[2]); }
static inline f3 vnrm(const MeshHost& m, uint32_t vi) { const float* p = &m.verts[(size_t)vi * 7]; return mk3(p[3], p[4], p[5]); }

void SceneHost::build_lights(BuiltScene& B) const {
    const uint32_t nmat = (uint32_t)(mats128.size() / 32);
    // ---- emissive triangle list + CDF: Renderer.cpp:2123-2233, 2237-2243 ----
    struct Tmp { float w; uint32_t order; uint32_t inst; f3 p0, p1, p2; float em[3]; };
    std::vector<Tmp> tmp;
    for (size_t ii = 0; ii < insts.size(); ii++) {
        const MeshHost& m = meshes[insts[ii].mesh];
        for (uint32_t t = 0; t < m.idx.size() / 3; t++) {
            uint32_t m0 = matids[m.matid_base + t * 3], m1 = matids[m.matid_base + t * 3 + 1], m2 = matids[m.matid_base + t * 3 + 2];
            if (m0 != m1 || m0 != m2) continue;                      // :2153-2156
            if (m0 >= nmat) continue;
            const float* mat = &mats128[(size_t)m0 * 32];
            if (!(mat[8] + mat[9] + mat[10] > 0.0f)) continue;       // :2162
            Tmp L;
            L.p0 = vpos(m, m.idx[t * 3]); L.p1 = vpos(m, m.idx[t * 3 + 1]); L.p2 = vpos(m, m.idx[t * 3 + 2]);
            float area = 0.5f * length(cross(L.p1 - L.p0, L.p2 - L.p0));   // ComputeTriangleWeight :2217-2233
            float inten = (mat[8] + mat[9] + mat[10]) / 3.0f;
            L.w = area * inten; L.order = (uint32_t)tmp.size(); L.inst = (uint32_t)ii;
            L.em[0] = mat[8]; L.em[1] = mat[9]; L.em[2] = mat[10];
            tmp.push_back(L);
        }
    }
    // :2187-2190 sorts descending by weight with std::sort (unstable); ties are broken here by collection order
    std::sort(tmp.begin(), tmp.end(), [](const Tmp& a, const Tmp& b) { return a.w > b.w || (a.w == b.w && a.order < b.order); });
    float total = 0.0f;
    for (auto& L : tmp) total += L.w;
    B.total_weight = total;
    B.lights.resize(tmp.size()); B.lights80.assign(tmp.size() * 20, 0.0f);
    float cum = 0.0f;
    const uint32_t nl = (uint32_t)tmp.size();
    for (size_t i = 0; i < tmp.size(); i++) {
        Tmp& L = tmp[i];
        float wn = L.w / total; cum += wn;
        float cdf = (i + 1 == tmp.size()) ? 1.0f : cum;             // :2208-2210
        float* r = &B.lights80[i * 20];
        r[0] = L.p0.x; r[1] = L.p0.y; r[2] = L.p0.z; r[3] = cdf;
        r[4] = L.p1.x; r[5] = L.p1.y; r[6] = L.p1.z; memcpy(&r[7], &L.inst, 4);
        r[8] = L.p2.x; r[9] = L.p2.y; r[10] = L.p2.z; r[11] = wn;
        r[12] = L.em[0]; r[13] = L.em[1]; r[14] = L.em[2]; memcpy(&r[15], &nl, 4);
        r[16] = total;
    }
    refresh_lights(B);
}
// the world-space half of the light records (the sample-independent part of SampleLightNEE_GI, Sampler_v6.hlsl:540-545, 566-575) from the 80-byte records, which hold the
// object-space corners, the instance, the weight and the CDF: all a TRANSFORM-only commit has to redo (which triangles emit, their weights and their order do not depend on
// the instance matrices — scanning the 11 M material ids of the street scene for them again was 3 ms of every refit commit)
void SceneHost::refresh_lights(BuiltScene& B) const {
    B.lights.resize(B.lights80.size() / 20);
    for (size_t i = 0; i < B.lights.size(); i++) {
        const float* r = &B.lights80[i * 20];
        uint32_t inst; memcpy(&inst, &r[7], 4);
        const f3 p0 = mk3(r[0], r[1], r[2]), p1 = mk3(r[4], r[5], r[6]), p2 = mk3(r[8], r[9], r[10]);
        const float cdf = r[3], wn = r[11];
        LightGPU& G = B.lights[i];
        const float* M = insts[inst].o2w;
        f3 xv = xform_point(M, p0), yv = xform_point(M, p1), zv = xform_point(M, p2);
        f3 cl = cross(yv - xv, zv - xv);
        f3 nrm = normalize(cl);
        float area_l = fabsf(length(cl) * 0.5f);
        G.xv[0] = xv.x; G.xv[1] = xv.y; G.xv[2] = xv.z; G.cdf = cdf;
        G.yv[0] = yv.x; G.yv[1] = yv.y; G.yv[2] = yv.z; G.pdf_l = maxf_(kEps, wn / maxf_(area_l, kEps));
        G.zv[0] = zv.x; G.zv[1] = zv.y; G.zv[2] = zv.z; G.pad0 = 0.0f;
        G.em[0] = r[12]; G.em[1] = r[13]; G.em[2] = r[14]; G.pad1 = 0.0f;
        G.nl[0] = nrm.x; G.nl[1] = nrm.y; G.nl[2] = nrm.z; G.pad2 = 0.0f;
    }
}

// transform-only commit on the GPU-refit path: the kernels re-derive triangles and boxes; the host re-derives what is small
bool SceneHost::refresh_transforms(BuiltScene& B) {
    if (topo_dirty || B.insts.size() != insts.size()) { err = "refresh_transforms: topology changed"; return false; }
    const bool mats_changed = mats_dirty;
    if (mats_dirty) build_materials(B);          // rtx_set_materials since the last commit: new table (the light list below reads the new Ke)
    B.inst_moved.assign(insts.size(), 0u);
    for (size_t ii = 0; ii < insts.size(); ii++) {
        const InstHost& in = insts[ii];
        B.inst_moved[ii] = memcmp(B.insts[ii].o2w, in.o2w, 64) != 0 ? 1u : 0u;
        memcpy(B.insts[ii].o2w, in.o2w, 64); memcpy(B.insts[ii].nrm, in.nrm, 64); memcpy(B.insts[ii].o2w_inv, in.o2w_inv, 64); memcpy(B.insts[ii].prev_o2w, in.prev_o2w, 64);
    }
    if (mats_changed) build_lights(B); else refresh_lights(B);       // (new materials can change WHICH triangles emit: full scan)
    B.refit_count++;
    return true;
}

// materials: MaterialOptimized rounding (Common_v6.hlsl:62-74).  Shared by the full build and by a commit that only changed materials
// (rtx_set_materials after the scene is resident: the BVH stays, the material table and the light list are re-derived).
void SceneHost::build_materials(BuiltScene& B) {
    const uint32_t nmat = (uint32_t)(mats128.size() / 32);
    B.mats.resize(nmat);
    for (uint32_t i = 0; i < nmat; i++) {
        const float* m = &mats128[(size_t)i * 32];   // Kd[4] Ks[3] Ni Ke[3] pad Pr_Pm_Ps_Pc[4] LUT[16]
        MatGPU& g = B.mats[i];
        for (int k = 0; k < 3; k++) { g.Kd[k] = half_round(m[k]); g.Ks[k] = half_round(m[4 + k]); g.Ke[k] = half_round(m[8 + k]); }
        g.Pr = half_round(m[12]); g.Pm = half_round(m[13]);
        g.KeFull[0] = m[8]; g.KeFull[1] = m[9]; g.KeFull[2] = m[10]; g.KeFullLen = length(mk3(m[8], m[9], m[10]));
        g.Ke_len = length(mk3(g.Ke[0], g.Ke[1], g.Ke[2]));
        for (int k = 0; k < 3; k++) g.KdPi[k] = g.Kd[k] / kPI;
        g.pad = 0.0f;
        g.alpha = half_round(m[3]); g.Ni = m[7]; g.pad1 = g.pad2 = 0.0f;      // MaterialOptimized.Kd.w (fp16) and the full-precision Material.Ni (strategy-3 extension)
        memcpy(g.LUT, m + 16, 64);
    }
    mats_dirty = false;
}

bool SceneHost::prepare_device_build(BuiltScene& B) {
    const bool TT = getenv("RTX_BUILD_TIMES") != nullptr; auto T0 = std::chrono::steady_clock::now(); auto lap = [&](const char* w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[build] %-28s %.3f s\n", w, std::chrono::duration<double>(t - T0).count()); T0 = t; } };
    build_materials(B);
    uint32_t nt = 0;
    for (auto& in : insts) { in.tri_base = nt; nt += (uint32_t)(meshes[in.mesh].idx.size() / 3); }
    B.insts.resize(insts.size());
    for (size_t ii = 0; ii < insts.size(); ii++) { const InstHost& in = insts[ii]; memcpy(B.insts[ii].o2w, in.o2w, 64); memcpy(B.insts[ii].nrm, in.nrm, 64); memcpy(B.insts[ii].o2w_inv, in.o2w_inv, 64); memcpy(B.insts[ii].prev_o2w, in.prev_o2w, 64); }
    lap("materials + instances");
    build_lights(B);
    lap("lights");
    B.shade.clear(); B.shade.shrink_to_fit(); B.objtris.clear(); B.objtris.shrink_to_fit();
    B.bvh_pad = 2e-6f; B.nodes.clear(); B.nodes8.clear(); B.tri_slots8.clear(); B.tris8.clear(); B.tris.clear(); B.leaf_order.clear(); B.level_start8.clear();
    B.small_recs.clear(); B.small_tris.clear(); B.small_poly.clear(); B.small_nrec = 0; B.small_nocc = 0; B.built_tris = nt; B.refit_count = 0; B.any_order = 0;
    topo_dirty = false;
    return true;
}

bool SceneHost::build(BuiltScene& B, bool host_bvh) {
    // tooling: RTX_BUILD_TIMES=1 prints the phases of a commit to stderr (tools/bvh_lab, tools/build_time.py)
    const bool TT = getenv("RTX_BUILD_TIMES") != nullptr; auto T0 = std::chrono::steady_clock::now(); auto lap = [&](const char* w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[build] %-28s %.3f s\n", w, std::chrono::duration<double>(t - T0).count()); T0 = t; } };
    build_materials(B);
    // ---- flatten instances to world-space triangles; per-triangle shade records (Hit_v6.hlsl:12-61) ----
    uint32_t nt = 0;
    for (auto& in : insts) { in.tri_base = nt; nt += (uint32_t)(meshes[in.mesh].idx.size() / 3); }
    std::vector<float> wtri((size_t)nt * 9);
    B.shade.resize(nt);
    B.insts.resize(insts.size());
    B.objtris.resize((size_t)nt * 3);
    float scale = 1.0f;
    for (size_t ii = 0; ii < insts.size(); ii++) {
        const InstHost& in = insts[ii]; const MeshHost& m = meshes[in.mesh];
        memcpy(B.insts[ii].o2w, in.o2w, 64); memcpy(B.insts[ii].nrm, in.nrm, 64); memcpy(B.insts[ii].o2w_inv, in.o2w_inv, 64); memcpy(B.insts[ii].prev_o2w, in.prev_o2w, 64);
        // (round 5) the triangles of a large mesh on up to 16 threads — every triangle writes its own records, the coordinate scale is a maximum: 0.26 s of the 3.8 M-triangle
        // street's commit on one core, and what is left of the host's work when the tree is built on the GPU
        const uint32_t ntm = (uint32_t)(m.idx.size() / 3);
        const unsigned hwc = std::thread::hardware_concurrency();
        const uint32_t nth = ntm >= 65536u ? std::min<uint32_t>(16u, std::max(1u, hwc ? hwc : 4u)) : 1u;
        std::vector<float> tscale(nth, 1.0f);
        auto flatten_range = [&](uint32_t t_lo, uint32_t t_hi, float& scale) {
        for (uint32_t t = t_lo; t < t_hi; t++) {
            uint32_t g = in.tri_base + t;
            uint32_t i0 = m.idx[t * 3], i1 = m.idx[t * 3 + 1], i2 = m.idx[t * 3 + 2];
            const uint32_t vi[3] = {i0, i1, i2};
            for (int k = 0; k < 3; k++) {
                const f3 op = vpos(m, vi[k]);
                B.objtris[(size_t)g * 3 + k] = {op.x, op.y, op.z, 0.0f};
                f3 w = xform_point(in.o2w, op);
                wtri[(size_t)g * 9 + k * 3] = w.x; wtri[(size_t)g * 9 + k * 3 + 1] = w.y; wtri[(size_t)g * 9 + k * 3 + 2] = w.z;
                scale = std::max(scale, std::max(fabsf(w.x), std::max(fabsf(w.y), fabsf(w.z))));
            }
            TriShade& s = B.shade[g];
            uint32_t mi = m.matid_base + 3 * t;    // == 3*PrimitiveIndex() + uint(v0.normal.w), Hit_v6.hlsl:16-17
            s.mat = mi < matids.size() ? matids[mi] : kMissMat;
            s.inst = (uint32_t)ii;
            f3 p0 = vpos(m, i0);
            f3 cr = cross(vpos(m, i1) - p0, vpos(m, i2) - p0);      // :28-30
            s.area = fabsf(length(cr) * 0.5f);                      // :31
            f3 flat = normalize(cr);                                // :32
            s.flat[0] = flat.x; s.flat[1] = flat.y; s.flat[2] = flat.z;
            float* dst[3] = {s.n0, s.n1, s.n2};
            for (int k = 0; k < 3; k++) {                           // :40-46 (all(n != 0) is per component)
                f3 nk = vnrm(m, vi[k]);
                f3 use = (nk.x != 0.0f && nk.y != 0.0f && nk.z != 0.0f) ? nk : flat;
                dst[k][0] = use.x; dst[k][1] = use.y; dst[k][2] = use.z;
            }
            s.guard_tau = 0.0f;
        }
        };
        if (nth <= 1) flatten_range(0, ntm, tscale[0]);
        else {
            std::vector<std::thread> pool;
            for (uint32_t k = 0; k < nth; k++) {
                const uint32_t lo_t = (uint32_t)((uint64_t)ntm * k / nth), hi_t = (uint32_t)((uint64_t)ntm * (k + 1) / nth);
                try { pool.emplace_back([&, lo_t, hi_t, k] { flatten_range(lo_t, hi_t, tscale[k]); }); } catch (const std::system_error&) { flatten_range(lo_t, hi_t, tscale[k]); }
            }
            for (std::thread& th : pool) th.join();
        }
        for (float v : tscale) scale = std::max(scale, v);
    }
    lap("flatten + shade records");
    build_lights(B);
    lap("lights");
    if (!host_bvh) {                    // RTX_OPT_GPU_BUILD: the tree is the device's business (csrc/rtx_build.hip); nothing of it is mirrored on the host
        B.bvh_pad = 2e-6f * scale; B.nodes.clear(); B.nodes8.clear(); B.tri_slots8.clear(); B.tris8.clear(); B.tris.clear(); B.leaf_order.clear(); B.level_start8.clear();
        B.small_recs.clear(); B.small_tris.clear(); B.small_poly.clear(); B.small_nrec = 0; B.small_nocc = 0; B.built_tris = nt; B.refit_count = 0; B.any_order = 0;
        topo_dirty = false;
        return true;
    }
    // ---- BVH: full binned-SAH build, or a REFIT when only instance transforms changed since the last build
    //      (the reference refits its TLAS every frame: Renderer.cpp:594, TopLevelASGenerator.cpp:149-250) ----
    std::vector<uint32_t>& leaf_order = B.leaf_order;
    const bool refit = !topo_dirty && B.built_tris == nt && !B.leaf_order.empty() && !B.nodes.empty();
    const float bvh_pad = 2e-6f * scale; B.bvh_pad = bvh_pad;                  // absolute box padding (1e-5 measured 3 % slower; the relative margins kSlabLo / kSlabHi carry the triangle-test error)
    if (refit) refit_bvh(wtri, bvh_pad, B.nodes, leaf_order);
    else { build_bvh(wtri, bvh_pad, B.nodes, leaf_order, B.max_depth, bvh); B.built_tris = nt; }
    lap("build_bvh");
    B.refit_count = refit ? B.refit_count + 1 : 0;
    topo_dirty = false;
    B.tris.resize(leaf_order.size());
    for (size_t s = 0; s < leaf_order.size(); s++) {
        uint32_t g = leaf_order[s];
        const float* t = &wtri[(size_t)g * 9];
        f3 v0 = mk3(t[0], t[1], t[2]);
        f3 e1 = mk3(t[3], t[4], t[5]) - v0, e2 = mk3(t[6], t[7], t[8]) - v0;
        TriGPU& T = B.tris[s];
        T.v0 = {v0.x, v0.y, v0.z, u2f(g)};
        T.e1 = {e1.x, e1.y, e1.z, tri_det_floor(e1, e2)};       // the hit definition's determinant floor (rtx_math.hpp)
        T.e2 = {e2.x, e2.y, e2.z, 0.0f};
    }
    // device traversal form: derived data, redone after a refit too (O(nodes))
    if (!collapse_bvh8(B.nodes, B.nodes8, B.tri_slots8, B.stack8, &B.level_start8, bvh)) { err = "build: BVH collapse failed"; return false; }
    lap("collapse_bvh8");
    B.tris8.resize(B.tri_slots8.size());
    for (size_t i = 0; i < B.tri_slots8.size(); i++) B.tris8[i] = B.tris[B.tri_slots8[i]];
    // ---- tiny scenes: merge triangles into planar convex quads and build the conservative pre-test records ----
    B.small_recs.clear(); B.small_tris.clear(); B.small_poly.clear(); B.small_nrec = 0; B.small_nocc = 0;
    if (!leaf_order.empty() && leaf_order.size() <= kSmallSceneMaxTris) {
        const double delta = 2e-5 * (double)scale, tol = 1e-6 * (double)scale;
        B.small_delta = (float)delta; B.small_cm = 4e-6f * scale; B.small_hull_margin = 2e-6f * scale;     // how far inside every hull plane an NEE origin must lie (20 x the float error of a hit position)
        struct D3 { double x, y, z; };
        auto sub = [](D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; };
        auto crs = [](D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
        auto dt = [](D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
        auto nrm = [&](D3 a) { double l = sqrt(dt(a, a)); return l > 0 ? D3{a.x / l, a.y / l, a.z / l} : D3{0, 0, 0}; };
        const size_t n = leaf_order.size();
        std::vector<std::array<D3, 3>> V(n);
        for (size_t s = 0; s < n; s++) { const float* t = &wtri[(size_t)leaf_order[s] * 9]; for (int k = 0; k < 3; k++) V[s][k] = D3{t[k * 3], t[k * 3 + 1], t[k * 3 + 2]}; }
        auto same = [](D3 a, D3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
        struct Rec { double pl[4]; double e[4][4]; int s0, s1; double pv[4][3]; };     // pv: polygon vertices (a triangle repeats its last one)
        std::vector<Rec> recs; std::vector<uint8_t> used(n, 0);
        auto make_rec = [&](const std::vector<D3>& poly, D3 nu, int s0, int s1) {
            Rec R; R.s0 = s0; R.s1 = s1;
            for (int k = 0; k < 4; k++) { const D3& q = poly[std::min<size_t>((size_t)k, poly.size() - 1)]; R.pv[k][0] = q.x; R.pv[k][1] = q.y; R.pv[k][2] = q.z; }
            R.pl[0] = nu.x; R.pl[1] = nu.y; R.pl[2] = nu.z; R.pl[3] = dt(nu, poly[0]);
            for (int k = 0; k < 4; k++) { R.e[k][0] = R.e[k][1] = R.e[k][2] = 0.0; R.e[k][3] = 1e30; }     // always inside
            for (size_t k = 0; k < poly.size(); k++) {
                D3 A = poly[k], Bv = poly[(k + 1) % poly.size()];
                D3 m = nrm(crs(nu, sub(Bv, A)));                     // in-plane, pointing inside for a polygon wound CCW about nu
                R.e[k][0] = m.x; R.e[k][1] = m.y; R.e[k][2] = m.z; R.e[k][3] = -dt(m, A);
            }
            return R;
        };
        for (size_t i = 0; i < n; i++) {
            if (used[i]) continue;
            used[i] = 1;
            D3 ni = crs(sub(V[i][1], V[i][0]), sub(V[i][2], V[i][0]));
            const double nn = sqrt(dt(ni, ni));
            if (!(nn > 0.0)) {                                       // zero-area triangle: the exact test always rejects it
                Rec R; R.s0 = (int)i; R.s1 = -1; for (int k = 0; k < 4; k++) { R.pl[k] = 0; R.e[k][0] = R.e[k][1] = R.e[k][2] = 0; R.e[k][3] = -1e30; R.pv[k][0] = V[i][0].x; R.pv[k][1] = V[i][0].y; R.pv[k][2] = V[i][0].z; }
                recs.push_back(R); continue;
            }
            D3 nu = nrm(ni);
            int partner = -1; std::vector<D3> quad;
            for (size_t j = i + 1; j < n && partner < 0; j++) {
                if (used[j]) continue;
                for (int a = 0; a < 3 && partner < 0; a++) {         // apex of i = vertex a, shared edge (a+1, a+2)
                    D3 r = V[i][a], pp = V[i][(a + 1) % 3], q = V[i][(a + 2) % 3];
                    for (int bb = 0; bb < 3; bb++) {
                        D3 sA = V[j][bb], j1 = V[j][(bb + 1) % 3], j2 = V[j][(bb + 2) % 3];
                        if (!((same(j1, pp) && same(j2, q)) || (same(j1, q) && same(j2, pp)))) continue;
                        if (fabs(dt(nu, sub(sA, r))) > tol) continue;                       // coplanar
                        std::vector<D3> poly = {r, pp, sA, q};                                // around the quad, CCW about nu
                        bool convex = true;
                        for (int k = 0; k < 4 && convex; k++) {
                            D3 m = nrm(crs(nu, sub(poly[(k + 1) % 4], poly[k])));
                            for (int v = 0; v < 4; v++) if (dt(m, sub(poly[v], poly[k])) < -1e-7 * (double)scale) { convex = false; break; }
                        }
                        if (!convex) continue;
                        partner = (int)j; quad = poly; break;
                    }
                }
            }
            if (partner >= 0) { used[partner] = 1; recs.push_back(make_rec(quad, nu, (int)i, partner)); }
            else recs.push_back(make_rec({V[i][0], V[i][1], V[i][2]}, nu, (int)i, -1));
        }
        // Faces of the scene's convex hull last: a record whose plane has ALL scene vertices on one side (within tol) cannot lie
        // strictly between two points of the scene, so NEE shadow segments (surface point + bias -> light point, shortened at both
        // ends) only need the records before them.  In a closed room that is every wall: Cornell keeps 11 of its 17 records.
        {
            // EMISSIVE records always stay in the occluder list: an NEE segment ENDS on a light, a margin of 1e-4 short of it, and for a long
            // grazing segment the float Moeller-Trumbore t of the light's own triangle is off by more than that, so the brute-force
            // definition (and the oracle) reports the light as its own occluder.  Found by the analytic rectangle-light test, whose light is
            // a hull face; the Cornell light hangs below the ceiling and was in the list anyway.
            auto emissive = [&](int slot) {
                if (slot < 0) return false;
                const uint32_t g = f2u(B.tris[(size_t)slot].v0.w);
                const uint32_t m = g < B.shade.size() ? B.shade[g].mat : 0xFFFFFFFFu;
                return m < B.mats.size() && B.mats[m].Ke_len > 0.0f;
            };
            // The same holds for a hull face whose PLANE carries a light vertex (a lamp flush with a wall or ceiling): the segment's end point
            // lies in that plane, so the face's triangles can pass the float test too.  Such faces stay in the list as well.
            std::vector<D3> light_verts;
            for (size_t s = 0; s < n; s++) if (emissive((int)s)) for (int k = 0; k < 3; k++) light_verts.push_back(V[s][k]);
            const double near_plane = 5.0 * (double)kSBias + 2e-4 * (double)scale;      // the segment's end margin + the float test's error of t (Cornell's light hangs 9e-4 below its ceiling: not near)
            // The shortcut also needs FLAT shading everywhere: the segment starts at pos + bias * SHADING normal and is only cast when the
            // shading normal faces the light; with interpolated vertex normals neither keeps it on the inner side of the face it starts
            // on (brute force then reports that face as the occluder).  Any smooth-shaded triangle turns the shortcut off for the scene.
            // ... and it needs room for the per-ray guard (traverse_small): the origin sits s_bias inside its OWN face, which must stay
            // outside the guard's margin, or every ray would fall back anyway.
            bool all_flat = B.small_hull_margin < 0.9f * kSBias;
            for (const TriShade& ts : B.shade) for (int k = 0; k < 3; k++)
                if (ts.n0[k] != ts.flat[k] || ts.n1[k] != ts.flat[k] || ts.n2[k] != ts.flat[k]) all_flat = false;
            std::vector<Rec> occ, hull;
            for (const Rec& R : recs) {
                if (!all_flat) { occ.push_back(R); continue; }
                bool pos = false, neg = false;
                const bool degenerate = R.pl[0] == 0.0 && R.pl[1] == 0.0 && R.pl[2] == 0.0;
                bool light = emissive(R.s0) || emissive(R.s1);
                for (const D3& q : light_verts) if (fabs(R.pl[0] * q.x + R.pl[1] * q.y + R.pl[2] * q.z - R.pl[3]) <= near_plane) light = true;
                for (size_t s = 0; s < n && !degenerate; s++) for (int k = 0; k < 3; k++) {
                    const double dd = R.pl[0] * V[s][k].x + R.pl[1] * V[s][k].y + R.pl[2] * V[s][k].z - R.pl[3];
                    if (dd > tol) pos = true; else if (dd < -tol) neg = true;
                }
                (((pos && neg) || light) ? occ : hull).push_back(R);
            }
            B.small_nocc = (uint32_t)occ.size();
            // Per-ray guard of the shortcut.  An NEE segment may skip the hull faces only if its ORIGIN lies clearly inside every hull
            // plane: a shading point in a room corner can sit within rounding distance of the neighbouring wall's plane, whose triangles the
            // float test then accepts for a segment grazing that wall (expected about once per 1080p x 64 spp Cornell frame).
            //   origin = pos + s_bias n_T (n_T: the shading normal the kernels compute, flat shading here), so for triangle T and hull plane B
            //   dist(origin, B) = dist(pos, B) + s_bias (n_T . n_B), n_B the plane's inward normal;  required >= safety (20 x the float
            //   error of pos), i.e.  dist(pos, B) >= s_TB := safety - s_bias (n_T . n_B).  Planes with s_TB <= 0 never matter (T's own plane,
            //   the other half of a slightly twisted wall: the origin is s_bias inside them wherever it is on T).
            //   dist(pos, B) is the barycentric blend of T's vertex distances d_i(B) >= 0, hence >= min(b) max_i d_i(B):
            //   tau_T = max over the planes that matter of s_TB / max_i d_i(B), and "min barycentric >= tau_T" proves the origin safe.
            // Three instructions per hit (TriShade::guard_tau); a wave with a ray that fails runs its shadow rays against all records.
            const double safety = (double)B.small_hull_margin;
            for (size_t si = 0; si < n && !hull.empty(); si++) {
                const uint32_t g = f2u(B.tris[si].v0.w);
                if (g >= B.shade.size()) continue;
                const TriShade& ts = B.shade[g];
                const f3 nw = normalize(xform_dir(B.insts[ts.inst].nrm, mk3(ts.flat[0], ts.flat[1], ts.flat[2])));      // = Surf::normal of a flat-shaded hit (rtx_shade.hpp)
                double tau = 0.0;
                for (const Rec& Hf : hull) {
                    double dmax = 0.0, side = 0.0;
                    for (size_t s2 = 0; s2 < n; s2++) for (int k = 0; k < 3; k++) {                                     // the scene's side of the plane
                        const double dd = Hf.pl[0] * V[s2][k].x + Hf.pl[1] * V[s2][k].y + Hf.pl[2] * V[s2][k].z - Hf.pl[3];
                        if (fabs(dd) > fabs(side)) side = dd;
                    }
                    const double sgn = side >= 0.0 ? 1.0 : -1.0;
                    for (int k = 0; k < 3; k++) dmax = std::max(dmax, fabs(Hf.pl[0] * V[si][k].x + Hf.pl[1] * V[si][k].y + Hf.pl[2] * V[si][k].z - Hf.pl[3]));
                    const double ndot = sgn * (Hf.pl[0] * (double)nw.x + Hf.pl[1] * (double)nw.y + Hf.pl[2] * (double)nw.z);
                    const double need = safety - (double)kSBias * ndot + 1e-7 * (double)kSBias;                            // (+ rounding of n_T)
                    if (!(need > 0.0)) continue;
                    tau = dmax > 0.0 ? std::max(tau, need / dmax) : 2.0;
                }
                B.shade[g].guard_tau = (float)std::min(2.0, tau * 1.000001);
            }
            recs = occ; recs.insert(recs.end(), hull.begin(), hull.end());
        }
        B.small_nrec = (uint32_t)recs.size();
        B.small_tris.assign(((recs.size() + 1) & ~(size_t)1) * 2, TriGPU{{0, 0, 0, u2f(kMissPrim)}, {0, 0, 0, 0}, {0, 0, 0, 0}});   // the padding record of an odd count owns two zero-area triangles
        const TriGPU none{{0, 0, 0, u2f(kMissPrim)}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (size_t r = 0; r < recs.size(); r++) { B.small_tris[2 * r] = B.tris[recs[r].s0]; B.small_tris[2 * r + 1] = recs[r].s1 >= 0 ? B.tris[recs[r].s1] : none; }
        // polygon corners per record: the primary-ray kernel culls records against the pyramid of each 8x8 pixel block
        B.small_poly.assign(((recs.size() + 1) & ~(size_t)1) * 4, F4{0.0f, 0.0f, 0.0f, 0.0f});
        for (size_t r = 0; r < recs.size(); r++) for (int k = 0; k < 4; k++) B.small_poly[r * 4 + k] = {(float)recs[r].pv[k][0], (float)recs[r].pv[k][1], (float)recs[r].pv[k][2], 0.0f};
        for (size_t r = 0; r < recs.size(); r += 2) {
            SmallRecPair P;
            for (int e = 0; e < 2; e++) {
                const bool have = r + e < recs.size();
                for (int row = 0; row < 20; row++) {
                    double v;
                    if (!have) v = (row == 7 || row == 11 || row == 15 || row == 19) ? -1e30 : 0.0;       // padding: never inside
                    else v = row < 4 ? recs[r + e].pl[row] : recs[r + e].e[(row - 4) / 4][(row - 4) % 4] + ((row - 4) % 4 == 3 ? delta : 0.0);   // edge constants carry the distance tolerance
                    P.r[row][e] = (float)v;
                }
            }
            B.small_recs.push_back(P);
        }
    }
    lap("tris8 / small scene");
    B.any_order = probe_anyhit_order(B);
    lap("probe");
    return true;
}

// ------------------------------------------------------------------------------------------------
// binned SAH BVH2 (16 bins, leaves of <= 4 triangles unless a split is impossible, hard cap 8)
// ------------------------------------------------------------------------------------------------
BvhBuildOptions& bvh_build_options() {
    static BvhBuildOptions o;
    static bool env_read = false;
    if (!env_read) {                                       // tooling: RTX_BVH="reinsert=2,split=1e-5" (A/B runs of one binary)
        env_read = true;
        if (const char* e = getenv("RTX_BVH")) {
            std::string s = e; size_t at = 0;
            while (at < s.size()) {
                size_t end = s.find(',', at); if (end == std::string::npos) end = s.size();
                const std::string kv = s.substr(at, end - at); const size_t eq = kv.find('=');
                if (eq != std::string::npos && !bvh_build_option(o, kv.substr(0, eq).c_str(), atof(kv.c_str() + eq + 1))) fprintf(stderr, "[rtx] RTX_BVH: unknown key in '%s'\n", kv.c_str());
                at = end + 1;
            }
        }
    }
    return o;
}
bool bvh_build_option(const char* key, double v) { return bvh_build_option(bvh_build_options(), key, v); }
bool bvh_build_option(BvhBuildOptions& o, const char* key, double v) {
    const std::string k = key ? key : "";
    if (k == "bins") o.bins = (int)v;
    else if (k == "sweep") o.sweep_below = (uint32_t)v;
    else if (k == "tri_cost") o.tri_cost = v;
    else if (k == "threads") o.threads = (int)v;
    else if (k == "ploc") o.ploc_radius = (int)v;
    else if (k == "ploc_top") o.ploc_top = (uint32_t)v;
    else if (k == "leaf_stop") o.leaf_stop = (uint32_t)v;
    else if (k == "split") o.split_alpha = v;
    else if (k == "slot_assign") o.slot_assign = (int)v;
    else if (k == "split_budget") o.split_budget = v;
    else if (k == "reinsert") o.reinsert_passes = (int)v;
    else if (k == "reinsert_frac") o.reinsert_frac = v;
    else if (k == "reinsert_cap") o.reinsert_cap = (uint32_t)v;
    else return false;
    return true;
}

namespace {
struct Box { float mn[3], mx[3]; };
inline Box empty_box() { Box b; for (int a = 0; a < 3; a++) { b.mn[a] = INFINITY; b.mx[a] = -INFINITY; } return b; }
inline void grow(Box& b, const Box& o) { for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], o.mn[a]); b.mx[a] = std::max(b.mx[a], o.mx[a]); } }
inline float half_area(const Box& b) {
    float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx < 0) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
struct TmpNode { Box box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; };
}

// Refit: keep the topology (node links, leaf order), recompute every child box bottom-up.  Nodes are stored
// breadth-first, so a child always has a larger index than its parent: one reverse sweep suffices.
void refit_bvh(const std::vector<float>& wtri, float pad_abs, std::vector<NodeGPU>& nodes, const std::vector<uint32_t>& order) {
    auto child_box = [&](int32_t child, float* mn, float* mx) {
        for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; }
        if (child == kEmptyChild) return;
        if (child < 0) {                                  // leaf: bounds of its triangles
            const uint32_t v = ~(uint32_t)child, first = v >> 3, cnt = (v & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; k++) {
                const float* t = &wtri[(size_t)order[first + k] * 9];
                for (int vtx = 0; vtx < 3; vtx++) for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], t[vtx * 3 + a]); mx[a] = std::max(mx[a], t[vtx * 3 + a]); }
            }
            for (int a = 0; a < 3; a++) { mn[a] -= pad_abs; mx[a] += pad_abs; }
        } else {                                          // internal: union of its two (already refitted, already padded) child boxes
            const NodeGPU& N = nodes[child];
            const float amn[3] = {N.a.x, N.a.y, N.a.z}, amx[3] = {N.a.w, N.b.x, N.b.y}, bmn[3] = {N.b.z, N.b.w, N.c.x}, bmx[3] = {N.c.y, N.c.z, N.c.w};
            for (int a = 0; a < 3; a++) { mn[a] = std::min(amn[a], bmn[a]); mx[a] = std::max(amx[a], bmx[a]); }
        }
    };
    for (size_t i = nodes.size(); i-- > 0;) {
        NodeGPU& N = nodes[i];
        float mn[3], mx[3];
        child_box((int32_t)f2u(N.d.x), mn, mx);
        N.a = {mn[0], mn[1], mn[2], mx[0]}; N.b.x = mx[1]; N.b.y = mx[2];
        child_box((int32_t)f2u(N.d.y), mn, mx);
        N.b.z = mn[0]; N.b.w = mn[1]; N.c = {mn[2], mx[0], mx[1], mx[2]};
    }
}

// ---- the top-down builder works on REFERENCES (box, triangle): a spatial split (Stich, Friedrich, Dietrich, "Spatial Splits in Bounding Volume
//      Hierarchies", HPG 2009) may hand a triangle to both sides of a plane, each side keeping the box of ITS part.  The closest hit is defined as
//      the minimum over all triangles (ties: lowest id) and any hit as existence, so a triangle referenced from two leaves changes no result; what
//      has to hold is COVERAGE: every point of a triangle lies in the box of one of its references and in every box above it.  Parts are clipped
//      in double and their boxes rounded outward, so the pieces' boxes cover the triangle like the whole box did. ----
namespace {
struct Ref { Box box; uint32_t tri; };
inline float f_below(float x) { return std::nextafterf(x, -INFINITY); }
inline float f_above(float x) { return std::nextafterf(x, INFINITY); }
inline Box intersect(const Box& a, const Box& b) { Box r; for (int k = 0; k < 3; k++) { r.mn[k] = std::max(a.mn[k], b.mn[k]); r.mx[k] = std::min(a.mx[k], b.mx[k]); } return r; }
// parts of triangle t9 inside `in` on either side of the plane x[axis] = pos
inline void split_ref(const float* t9, const Box& in, int axis, float pos, Box& L, Box& R) {
    L = empty_box(); R = empty_box();
    auto add = [](Box& b, const double* p, bool exact) {
        for (int k = 0; k < 3; k++) {
            const float f = (float)p[k];
            const float lo = exact ? f : ((double)f > p[k] ? f_below(f) : f), hi = exact ? f : ((double)f < p[k] ? f_above(f) : f);
            b.mn[k] = std::min(b.mn[k], exact ? f : f_below(lo)); b.mx[k] = std::max(b.mx[k], exact ? f : f_above(hi));
        }
    };
    for (int e = 0; e < 3; e++) {
        const float* a = t9 + 3 * e; const float* b = t9 + 3 * ((e + 1) % 3);
        const double pa[3] = {a[0], a[1], a[2]};
        if (a[axis] <= pos) add(L, pa, true);
        if (a[axis] >= pos) add(R, pa, true);
        if ((a[axis] < pos && b[axis] > pos) || (a[axis] > pos && b[axis] < pos)) {
            const double t = std::min(1.0, std::max(0.0, ((double)pos - (double)a[axis]) / ((double)b[axis] - (double)a[axis])));
            double p[3]; for (int k = 0; k < 3; k++) p[k] = (double)a[k] + t * ((double)b[k] - (double)a[k]);
            p[axis] = pos;
            add(L, p, false); add(R, p, false);
        }
    }
    L.mx[axis] = std::min(L.mx[axis], pos); R.mn[axis] = std::max(R.mn[axis], pos);
    L = intersect(L, in); R = intersect(R, in);
    L.mx[axis] = std::max(L.mx[axis], L.mn[axis]); R.mx[axis] = std::max(R.mx[axis], R.mn[axis]);    // (a sliver part keeps a valid, zero-width box)
}
inline bool valid_box(const Box& b) { return b.mn[0] <= b.mx[0] && b.mn[1] <= b.mx[1] && b.mn[2] <= b.mx[2]; }

// Insertion-based optimisation of the binary tree (Bittner, Hapala, Havran, "Fast Insertion-Based Optimization of Bounding Volume Hierarchies", CGF 2013; the
// per-node search of Meister & Bittner, "Parallel Reinsertion for Bounding Volume Hierarchy Optimization", EG 2018): a subtree is cut out and put back where it
// enlarges the fewest / smallest boxes (branch-and-bound over the induced surface-area cost).  Topology only: leaves and their references stay as they are.
void reinsert_pass(std::vector<TmpNode>& tn, std::vector<int32_t>& parent, double frac) {
    const size_t n = tn.size();
    // the candidates: largest boxes first, ties in index order (what a stable sort by area gives) — but only the first `frac` of that order is wanted, so: select, then sort
    // the selection (a full stable_sort of 1.9 M nodes with the area recomputed in every comparison was 2/3 of the pass's time on the street scene)
    std::vector<std::pair<float, uint32_t>> keyed; keyed.reserve(n);
    for (size_t i = 1; i < n; i++) if (parent[i] > 0) keyed.push_back({half_area(tn[i].box), (uint32_t)i});      // not the root, not a child of the root (the root stays node 0)
    const auto before = [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); };
    const size_t keep = (size_t)((double)keyed.size() * frac);
    if (keep < keyed.size()) std::nth_element(keyed.begin(), keyed.begin() + keep, keyed.end(), before);
    std::sort(keyed.begin(), keyed.begin() + keep, before);
    std::vector<uint32_t> cand(keep);
    for (size_t i = 0; i < keep; i++) cand[i] = keyed[i].second;
    auto refit_up = [&](int32_t a) {
        for (; a >= 0; a = parent[a]) {
            Box b = tn[tn[a].left].box; grow(b, tn[tn[a].right].box);
            if (!memcmp(&b, &tn[a].box, sizeof(Box))) break;
            tn[a].box = b;
        }
    };
    struct It { float bound; float induced; int32_t node; };
    auto cmp = [](const It& a, const It& b) { return a.bound > b.bound; };
    std::vector<It> pq;
    for (uint32_t x : cand) {
        const int32_t p = parent[x];
        if (p <= 0) continue;                                   // (moves may have lifted x to the root's children)
        const int32_t g = parent[p], s = tn[p].left == (int32_t)x ? tn[p].right : tn[p].left;
        // cut x (and its parent node p) out
        (tn[g].left == p ? tn[g].left : tn[g].right) = s; parent[s] = g;
        refit_up(g);
        const Box xb = tn[x].box; const float xa = half_area(xb);
        float best = INFINITY; int32_t best_node = s;
        pq.clear();
        pq.push_back({0.0f, 0.0f, tn[0].left}); pq.push_back({0.0f, 0.0f, tn[0].right});
        {   // the root's own enlargement is paid by every position alike: leave it out
        }
        std::make_heap(pq.begin(), pq.end(), cmp);
        while (!pq.empty()) {
            std::pop_heap(pq.begin(), pq.end(), cmp); const It it = pq.back(); pq.pop_back();
            if (it.bound + xa >= best) break;
            Box u = tn[it.node].box; grow(u, xb);
            const float direct = half_area(u), total = it.induced + direct;
            if (total < best) { best = total; best_node = it.node; }
            if (!tn[it.node].count) {
                const float ind = it.induced + direct - half_area(tn[it.node].box);
                if (ind + xa < best) {
                    pq.push_back({ind, ind, tn[it.node].left}); std::push_heap(pq.begin(), pq.end(), cmp);
                    pq.push_back({ind, ind, tn[it.node].right}); std::push_heap(pq.begin(), pq.end(), cmp);
                }
            }
        }
        // put it back: p becomes the parent of (best_node, x) where best_node was
        const int32_t gb = parent[best_node];
        (tn[gb].left == best_node ? tn[gb].left : tn[gb].right) = p; parent[p] = gb;
        tn[p].left = best_node; tn[p].right = (int32_t)x; parent[best_node] = p; parent[x] = p;
        tn[p].box = tn[best_node].box; grow(tn[p].box, xb);
        refit_up(gb);
    }
}
}  // namespace

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG 2018) — the
//      BOTTOM-UP builder of the GPU build (csrc/rtx_build.hip: RTX_OPT_GPU_BUILD), restated here so that its trees can be judged by work per ray without a GPU
//      (tools/bvh_lab: ploc=<radius>) and so that the device code has a host twin to be compared with node for node.  Triangles are sorted along the Morton curve of
//      their box centres (63 bits, ties by triangle id); every cluster looks `radius` places to either side for the neighbour whose union with it has the smallest
//      surface area; mutual nearest neighbours merge; repeat until one cluster is left.  Everything is a pure function of the input order, so host and device agree. ----
namespace {
struct PlocNode { Box box; int32_t left, right; uint32_t tri; };
// clusters until at most `stop_at` are left; pool: [0, n) leaves in Morton order, internal nodes appended in creation order (iteration by iteration, left partners in cluster order)
void ploc_clusters(const std::vector<Ref>& refs, const Box& scene, int radius, uint32_t stop_at, std::vector<PlocNode>& pn, std::vector<int32_t>& cl) {
    const uint32_t n = (uint32_t)refs.size();
    // Morton keys of the box centres on a 2^21 grid over the scene's box (float arithmetic, the device's formula: rtx_wide.hpp ploc_morton)
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    float lo[3], inv[3];
    ploc_grid(scene.mn, scene.mx, lo, inv);
    auto wb = [](const Box& b) { WBox w; for (int a = 0; a < 3; a++) { w.mn[a] = b.mn[a]; w.mx[a] = b.mx[a]; } return w; };
    for (uint32_t i = 0; i < n; i++) keyed[i] = {ploc_morton(wb(refs[i].box), lo, inv), i};
    std::sort(keyed.begin(), keyed.end());
    pn.clear(); pn.reserve((size_t)2 * n);
    for (uint32_t i = 0; i < n; i++) { const Ref& r = refs[keyed[i].second]; pn.push_back(PlocNode{r.box, -1, -1, r.tri}); }
    cl.resize(n); for (uint32_t i = 0; i < n; i++) cl[i] = (int32_t)i;
    std::vector<int32_t> nn, nxt;
    while (cl.size() > std::max<size_t>(1, stop_at)) {
        const int m = (int)cl.size();
        nn.assign(m, -1);
        for (int i = 0; i < m; i++) nn[i] = ploc_nearest(i, m, radius, [&](int j) { return wb(pn[cl[j]].box); });      // nearest neighbour within the window (rtx_wide.hpp)
        nxt.clear();
        for (int i = 0; i < m; i++) {
            const int j = nn[i];
            if (j >= 0 && nn[j] == i) {                       // mutual: the lower position becomes the new node, the higher one disappears
                if (i < j) { PlocNode N; N.box = pn[cl[i]].box; grow(N.box, pn[cl[j]].box); N.left = cl[i]; N.right = cl[j]; N.tri = 0; pn.push_back(N); nxt.push_back((int32_t)pn.size() - 1); }
            } else nxt.push_back(cl[i]);
        }
        cl.swap(nxt);
    }
}
// hang the PLOC subtree `src` of the pool below node `dst` of the build's tree (leaves of ONE triangle each, depth-first left to right)
void ploc_expand(const std::vector<PlocNode>& pn, int32_t src0, int32_t dst0, uint32_t depth0, std::vector<TmpNode>& tn, std::vector<uint32_t>& order, uint32_t& max_depth) {
    struct It { int32_t src, dst; uint32_t depth; };
    std::vector<It> st; st.push_back({src0, dst0, depth0});
    while (!st.empty()) {
        const It it = st.back(); st.pop_back();
        max_depth = std::max(max_depth, it.depth);
        const PlocNode& N = pn[it.src];
        tn[it.dst].box = N.box;
        if (N.left < 0) { tn[it.dst].first = (uint32_t)order.size(); tn[it.dst].count = 1; tn[it.dst].left = tn[it.dst].right = -1; order.push_back(N.tri); continue; }
        const int32_t l = (int32_t)tn.size(); tn.emplace_back(); const int32_t r = (int32_t)tn.size(); tn.emplace_back();
        tn[it.dst].left = l; tn[it.dst].right = r; tn[it.dst].count = 0;
        st.push_back({N.right, r, it.depth + 1}); st.push_back({N.left, l, it.depth + 1});
    }
}
}  // namespace

// the top-down builder (+ the re-insertion passes) over a set of references: fills the temporary tree `tn` (root = node 0; leaves hold [first, first + count) of `order`)
static void build_tmp_tree(std::vector<Ref>& refs, const Box& scene, const BvhBuildOptions& opt, std::vector<TmpNode>& tn, std::vector<uint32_t>& order, uint32_t& max_depth,
                           const std::function<void(const char*)>& lap, const float* wtri /* 9 floats per triangle: spatial splits clip against them (nullptr: no splits) */) {
    const uint32_t nt = (uint32_t)refs.size();
    order.clear(); order.reserve(nt);
    tn.clear(); tn.reserve((size_t)2 * nt + 2);
    max_depth = 0;
    // spatial splits only for scenes that take the BVH path (the tiny-scene records are built from the leaf order as a permutation of the triangles)
    const bool spatial = opt.split_alpha > 0.0 && nt > kSmallSceneMaxTris && wtri != nullptr;
    const float spatial_min = (float)(opt.split_alpha * (double)half_area(scene));
    const size_t ref_budget = (size_t)((double)nt * (1.0 + opt.split_budget)) + 8;
    size_t refs_total = nt;                                                       // references handed out so far (leaves made + still on the stack)
    struct Job { int32_t node; uint32_t count, depth; };
    constexpr int NB = 16, NS = 16;
    auto cen = [](const Ref& r, int a) { return 0.5f * (r.box.mn[a] + r.box.mx[a]); };
    // PARALLEL top-down phase (round 4: 9.3 of the 10 s a commit of the 3.8 M-triangle street took were this function, on one core).  The serial loop runs until a node has at
    // most `cutoff` references, moves that node's references out as a TASK and goes on; the tasks then run the same loop on private stacks in a thread pool, and their
    // subtrees are spliced back in the order in which they were cut.  A subtree is a function of its references alone (no spatial-split budget is shared: with spatial
    // splits the build stays serial), `cutoff` depends on the triangle count only, and the nodes are renumbered into the serial loop's creation order afterwards — so the tree
    // is THE SAME tree, node for node, as the serial build's, whatever the number of threads.
    struct Task { int32_t node; uint32_t depth; std::vector<Ref> refs; std::vector<TmpNode> tn; std::vector<uint32_t> order; uint32_t max_depth = 0; };
    std::vector<Task> tasks;
    const uint32_t cutoff = (!spatial && nt >= 65536u && opt.threads != 1) ? std::max<uint32_t>(4096u, nt / 256u) : 0u;
    auto run = [&](std::vector<Ref>& refs, std::vector<TmpNode>& tn, std::vector<uint32_t>& order, std::vector<Job>& st, uint32_t& max_depth, size_t& refs_total, bool may_defer) {
    std::vector<uint32_t> sweep_ids; std::vector<float> sweep_ra; std::vector<Ref> tmp;
    while (!st.empty()) {
        const Job j = st.back(); st.pop_back();
        max_depth = std::max(max_depth, j.depth);
        Ref* R = refs.data() + (refs.size() - j.count);                              // this node's references: the top of the reference stack
        if (may_defer && j.count <= cutoff && j.count > 4u) {                         // cut this subtree out: a task of the pool
            tasks.emplace_back(); Task& T = tasks.back();
            T.node = j.node; T.depth = j.depth; T.refs.assign(R, R + j.count);
            refs.resize(refs.size() - j.count);
            continue;
        }
        Box nb = empty_box(), cb = empty_box();
        for (uint32_t i = 0; i < j.count; i++) {
            grow(nb, R[i].box);
            for (int a = 0; a < 3; a++) { const float c = cen(R[i], a); cb.mn[a] = std::min(cb.mn[a], c); cb.mx[a] = std::max(cb.mx[a], c); }
        }
        tn[j.node].box = nb;
        auto make_leaf = [&]() {
            tn[j.node].first = (uint32_t)order.size(); tn[j.node].count = j.count;
            for (uint32_t i = 0; i < j.count; i++) order.push_back(R[i].tri);
            refs.resize(refs.size() - j.count);
        };
        if (j.count <= opt.leaf_stop || j.depth >= 96u) {
            if (j.count <= 4) { make_leaf(); continue; }
        }
        // ---- best object split over 3 axes: full sweep over the sorted centroids for small nodes, bins above ----
        float best_cost = INFINITY, sweep_split = 0.0f; int best_axis = -1, best_bin = -1;
        Box best_lb = empty_box(), best_rb = empty_box();
        const bool sweep = j.count <= opt.sweep_below;
        if (sweep) {
            sweep_ids.resize(j.count); sweep_ra.resize(j.count);
            for (int a = 0; a < 3; a++) {
                if (!(cb.mx[a] - cb.mn[a] > 0.0f)) continue;
                for (uint32_t i = 0; i < j.count; i++) sweep_ids[i] = i;
                std::stable_sort(sweep_ids.begin(), sweep_ids.end(), [&](uint32_t x, uint32_t y) { return cen(R[x], a) < cen(R[y], a); });
                Box acc = empty_box();
                for (uint32_t i = j.count; i-- > 1;) { grow(acc, R[sweep_ids[i]].box); sweep_ra[i] = half_area(acc); }
                acc = empty_box();
                for (uint32_t i = 0; i + 1 < j.count; i++) {
                    grow(acc, R[sweep_ids[i]].box);
                    const float c0 = cen(R[sweep_ids[i]], a), c1 = cen(R[sweep_ids[i + 1]], a);
                    if (c0 == c1) continue;                                          // equal centroids stay together (the partition is by value)
                    const float cost = half_area(acc) * (float)(i + 1) + sweep_ra[i + 1] * (float)(j.count - i - 1);
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = (int)i; sweep_split = 0.5f * (c0 + c1); if (!(sweep_split > c0)) sweep_split = c1; }
                }
            }
        } else {
            for (int a = 0; a < 3; a++) {
                const float lo = cb.mn[a], ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0.0f)) continue;
                Box bb[NB]; uint32_t bc[NB];
                for (int b = 0; b < NB; b++) { bb[b] = empty_box(); bc[b] = 0; }
                const float k = (float)NB / ext;
                for (uint32_t i = 0; i < j.count; i++) {
                    int b = (int)((cen(R[i], a) - lo) * k); if (b >= NB) b = NB - 1; if (b < 0) b = 0;
                    grow(bb[b], R[i].box); bc[b]++;
                }
                float ra[NB]; uint32_t rc[NB]; Box rbx[NB]; Box acc = empty_box(); uint32_t c = 0;
                for (int b = NB - 1; b > 0; b--) { grow(acc, bb[b]); c += bc[b]; ra[b] = half_area(acc); rc[b] = c; rbx[b] = acc; }
                acc = empty_box(); c = 0;
                for (int b = 0; b < NB - 1; b++) {
                    grow(acc, bb[b]); c += bc[b];
                    if (!c || !rc[b + 1]) continue;
                    const float cost = half_area(acc) * (float)c + ra[b + 1] * (float)rc[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_lb = acc; best_rb = rbx[b + 1]; }
                }
            }
        }
        const float leaf_cost = half_area(nb) * (float)j.count;
        // ---- spatial split candidate: only where the object split leaves the two sides overlapping (Stich et al., section 4.5) ----
        float sp_cost = INFINITY, sp_pos = 0.0f; int sp_axis = -1;
        if (spatial && j.count > 2 && refs_total < ref_budget) {
            bool try_it = best_axis < 0;
            if (!try_it) {
                if (sweep) {                                                         // (the sweep kept no boxes: rebuild the two sides of its best split)
                    best_lb = empty_box(); best_rb = empty_box();
                    for (uint32_t i = 0; i < j.count; i++) grow(cen(R[i], best_axis) < sweep_split ? best_lb : best_rb, R[i].box);
                }
                const Box ov = intersect(best_lb, best_rb);
                try_it = valid_box(ov) && half_area(ov) > spatial_min;
            }
            if (try_it) {
                for (int a = 0; a < 3; a++) {
                    const float lo = nb.mn[a], ext = nb.mx[a] - nb.mn[a];
                    if (!(ext > 0.0f)) continue;
                    Box bb[NS]; uint32_t enter[NS], leave[NS];
                    for (int b = 0; b < NS; b++) { bb[b] = empty_box(); enter[b] = leave[b] = 0; }
                    const float k = (float)NS / ext;
                    auto plane = [&](int b) { return lo + ext * ((float)b / (float)NS); };
                    for (uint32_t i = 0; i < j.count; i++) {
                        int b0 = (int)((R[i].box.mn[a] - lo) * k), b1 = (int)((R[i].box.mx[a] - lo) * k);
                        b0 = std::min(NS - 1, std::max(0, b0)); b1 = std::min(NS - 1, std::max(b0, b1));
                        while (b0 < b1 && plane(b0 + 1) <= R[i].box.mn[a]) b0++;               // (float binning vs. the plane positions used for chopping)
                        while (b1 > b0 && plane(b1) >= R[i].box.mx[a]) b1--;
                        enter[b0]++; leave[b1]++;
                        Box cur = R[i].box;
                        for (int b = b0; b < b1; b++) {
                            Box l, r; split_ref(&wtri[(size_t)R[i].tri * 9], cur, a, plane(b + 1), l, r);
                            if (valid_box(l)) grow(bb[b], l);
                            cur = r;
                            if (!valid_box(cur)) break;
                        }
                        if (valid_box(cur)) grow(bb[b1], cur);
                    }
                    float ra[NS]; uint32_t rc[NS]; Box acc = empty_box(); uint32_t c = 0;
                    for (int b = NS - 1; b > 0; b--) { grow(acc, bb[b]); c += leave[b]; ra[b] = half_area(acc); rc[b] = c; }
                    acc = empty_box(); c = 0;
                    for (int b = 0; b < NS - 1; b++) {
                        grow(acc, bb[b]); c += enter[b];
                        if (!c || !rc[b + 1] || c >= j.count || rc[b + 1] >= j.count) continue;       // a split that sends every reference to one side makes no progress
                        const float cost = half_area(acc) * (float)c + ra[b + 1] * (float)rc[b + 1];
                        if (cost < sp_cost) { sp_cost = cost; sp_axis = a; sp_pos = plane(b + 1); }
                    }
                }
            }
        }
        uint32_t nl = 0, nr = 0;                                                     // sizes of the two sides, laid out as [.. | left | right] on the reference stack
        bool split = false;
        if (sp_axis >= 0 && sp_cost < best_cost && (j.count > 4 || sp_cost + half_area(nb) < leaf_cost)) {
            // ---- spatial split with reference unsplitting (section 4.4): a straddling reference goes to both sides, or whole to one if that is cheaper ----
            tmp.clear();
            Box lb = empty_box(), rb = empty_box();
            std::vector<Ref> left, right, both;
            for (uint32_t i = 0; i < j.count; i++) {
                if (R[i].box.mx[sp_axis] <= sp_pos) { left.push_back(R[i]); grow(lb, R[i].box); }
                else if (R[i].box.mn[sp_axis] >= sp_pos) { right.push_back(R[i]); grow(rb, R[i].box); }
                else both.push_back(R[i]);
            }
            uint32_t cl = (uint32_t)(left.size() + both.size()), cr = (uint32_t)(right.size() + both.size());
            for (const Ref& r : both) {
                Box l, rr; split_ref(&wtri[(size_t)r.tri * 9], r.box, sp_axis, sp_pos, l, rr);
                const bool lv = valid_box(l), rv = valid_box(rr);
                Box lbs = lb, rbs = rb, lbw = lb, rbw = rb;
                if (lv) grow(lbs, l); if (rv) grow(rbs, rr); grow(lbw, r.box); grow(rbw, r.box);
                const float c_split = half_area(lbs) * (float)cl + half_area(rbs) * (float)cr;
                const float c_left = half_area(lbw) * (float)cl + half_area(rb) * (float)(cr - 1);
                const float c_right = half_area(lb) * (float)(cl - 1) + half_area(rbw) * (float)cr;
                if (lv && rv && c_split <= c_left && c_split <= c_right && refs_total < ref_budget) {
                    left.push_back({l, r.tri}); right.push_back({rr, r.tri}); lb = lbs; rb = rbs; refs_total++;
                } else if ((c_left <= c_right && cr > 1) || !rv || cl <= 1) { left.push_back(r); lb = lbw; cr--; }
                else { right.push_back(r); rb = rbw; cl--; }
            }
            nl = (uint32_t)left.size(); nr = (uint32_t)right.size();
            if (nl && nr && nl < j.count + both.size() && nr < j.count + both.size() && !(nl >= j.count && nr >= j.count)) {
                refs.resize(refs.size() - j.count);
                refs.insert(refs.end(), left.begin(), left.end()); refs.insert(refs.end(), right.begin(), right.end());
                split = true;
            } else { refs_total -= (nl + nr > j.count) ? (nl + nr - j.count) : 0; nl = nr = 0; }
        }
        if (!split && best_axis >= 0 && (j.count > 4 || best_cost + half_area(nb) * 1.0f < leaf_cost)) {
            const float lo = cb.mn[best_axis], k = (float)NB / (cb.mx[best_axis] - cb.mn[best_axis]);
            Ref* mid = sweep ? std::stable_partition(R, R + j.count, [&](const Ref& r) { return cen(r, best_axis) < sweep_split; })
                             : std::stable_partition(R, R + j.count, [&](const Ref& r) { int b = (int)((cen(r, best_axis) - lo) * k); if (b >= NB) b = NB - 1; if (b < 0) b = 0; return b <= best_bin; });
            nl = (uint32_t)(mid - R); nr = j.count - nl;
            split = nl > 0 && nr > 0;
        }
        if (!split) {
            if (j.count <= 4) { make_leaf(); continue; }            // leaves hold at most 4 triangles (the wide node's 4-bit slots)
            nl = j.count / 2; nr = j.count - nl;                    // degenerate (all centroids equal) or forced: median split by index
        }
        const int32_t l = (int32_t)tn.size(); tn.emplace_back();
        const int32_t r = (int32_t)tn.size(); tn.emplace_back();
        tn[j.node].left = l; tn[j.node].right = r;
        st.push_back({l, nl, j.depth + 1});
        st.push_back({r, nr, j.depth + 1});                          // the right side lies on top of the reference stack: it is processed first
    }
    };
    {
        std::vector<Job> st; tn.emplace_back(); st.push_back({0, nt, 0u});
        run(refs, tn, order, st, max_depth, refs_total, cutoff != 0u);
    }
    lap("top-down, serial part");
    if (!tasks.empty()) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
                Task& T = tasks[k];
                std::vector<Job> st; size_t local_total = 0;
                T.tn.reserve(2 * T.refs.size() + 2); T.order.reserve(T.refs.size());
                T.tn.emplace_back(); st.push_back({0, (uint32_t)T.refs.size(), T.depth});
                run(T.refs, T.tn, T.order, st, T.max_depth, local_total, false);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthreads = std::min<size_t>(tasks.size(), opt.threads > 1 ? (unsigned)opt.threads : std::min<unsigned>(hw ? hw : 4u, 16u));
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) { try { pool.emplace_back(work); } catch (const std::system_error&) { break; } }      // (no more threads to be had: the ones that started and this one do the work)
        work();
        for (std::thread& t : pool) t.join();
        lap("top-down, tasks");
        for (Task& T : tasks) {                                      // splice: local node 0 is the node the task was cut at, local i > 0 becomes base + i - 1
            const int32_t base = (int32_t)tn.size(); const uint32_t obase = (uint32_t)order.size();
            auto gid = [&](int32_t i) { return i == 0 ? T.node : base + i - 1; };
            for (size_t i = 0; i < T.tn.size(); i++) {
                TmpNode n = T.tn[i];
                if (n.count) n.first += obase; else { n.left = gid(n.left); n.right = gid(n.right); }
                if (i == 0) tn[T.node] = n; else tn.push_back(n);
            }
            order.insert(order.end(), T.order.begin(), T.order.end());
            max_depth = std::max(max_depth, T.max_depth);
        }
        // the serial loop's node numbering: a node's two children are created when it is processed, and the right child is processed first
        std::vector<int32_t> new_id(tn.size(), -1), stack_; int32_t nid = 1; new_id[0] = 0; stack_.push_back(0);
        while (!stack_.empty()) {
            const int32_t x = stack_.back(); stack_.pop_back();
            if (tn[x].count) continue;
            new_id[tn[x].left] = nid++; new_id[tn[x].right] = nid++;
            stack_.push_back(tn[x].left); stack_.push_back(tn[x].right);
        }
        std::vector<TmpNode> ren(tn.size());
        for (size_t i = 0; i < tn.size(); i++) { TmpNode n = tn[i]; if (!n.count) { n.left = new_id[n.left]; n.right = new_id[n.right]; } ren[new_id[i]] = n; }
        tn.swap(ren);
        tasks.clear(); tasks.shrink_to_fit();
    }
    lap("splice");
    // ---- insertion-based optimisation of the topology ----
    if (opt.reinsert_passes > 0 && tn.size() > 7) {
        std::vector<int32_t> parent(tn.size(), -1);
        for (size_t i = 0; i < tn.size(); i++) if (!tn[i].count) { parent[tn[i].left] = (int32_t)i; parent[tn[i].right] = (int32_t)i; }
        // a pass tries the nodes with the largest boxes: all of them on small trees, the top `reinsert_cap` on large ones (on the 3.8 M-triangle street the largest 10 % of the
        // nodes carry 5.4 of the 7 % a full pass takes off the shadow rays' node steps, at a seventh of its time)
        const double frac = std::min(opt.reinsert_frac, (double)opt.reinsert_cap / (double)tn.size());
        for (int pass = 0; pass < opt.reinsert_passes; pass++) reinsert_pass(tn, parent, frac);
        std::vector<std::pair<int32_t, uint32_t>> dst; dst.push_back({0, 0u}); max_depth = 0;
        while (!dst.empty()) { const auto it = dst.back(); dst.pop_back(); max_depth = std::max(max_depth, it.second); if (!tn[it.first].count) { dst.push_back({tn[it.first].left, it.second + 1}); dst.push_back({tn[it.first].right, it.second + 1}); } }
    }
    lap("re-insertion");
}

// The TOP of a PLOC tree (host twin and GPU build alike): the top-down SAH builder and the re-insertion passes over the clusters PLOC stopped at, single clusters as leaves
// (a leaf the builder refuses to split becomes a chain).  out: root first; left / right >= 0: index into out, < 0: ~cluster index; boxes unpadded.
void build_cluster_top(const float* boxes6, uint32_t m, const BvhBuildOptions& opt_in, std::vector<ClusterTopNode>& out) {
    BvhBuildOptions opt = opt_in; opt.leaf_stop = 1; opt.split_alpha = 0.0; opt.ploc_radius = 0;
    std::vector<Ref> refs(m); Box scene = empty_box();
    for (uint32_t i = 0; i < m; i++) { for (int a = 0; a < 3; a++) { refs[i].box.mn[a] = boxes6[(size_t)i * 6 + a]; refs[i].box.mx[a] = boxes6[(size_t)i * 6 + 3 + a]; } refs[i].tri = i; grow(scene, refs[i].box); }
    const std::vector<Ref> cref = refs;                                         // (the builder consumes its reference stack)
    std::vector<TmpNode> tn; std::vector<uint32_t> order; uint32_t depth = 0;
    build_tmp_tree(refs, scene, opt, tn, order, depth, [](const char*) {}, nullptr);
    out.clear();
    if (m == 0) return;
    struct It { int32_t src, dst; };
    std::vector<It> st; out.emplace_back(); st.push_back({0, 0});
    auto setbox = [](ClusterTopNode& N, const Box& b) { for (int a = 0; a < 3; a++) { N.mn[a] = b.mn[a]; N.mx[a] = b.mx[a]; } };
    while (!st.empty()) {
        const It it = st.back(); st.pop_back();
        const TmpNode T = tn[it.src];
        if (!T.count) {
            setbox(out[it.dst], T.box);
            int32_t child[2];
            for (int w = 0; w < 2; w++) {
                const int32_t c = w ? T.right : T.left;
                if (tn[c].count == 1) child[w] = ~(int32_t)order[tn[c].first];
                else { child[w] = (int32_t)out.size(); out.emplace_back(); st.push_back({c, child[w]}); }
            }
            out[it.dst].left = child[0]; out[it.dst].right = child[1];
            continue;
        }
        // a leaf of k >= 2 clusters (or the root as a leaf): a chain  (c0, (c1, (c2, ...)))
        const uint32_t f = T.first, k = T.count;
        if (k == 1) { setbox(out[it.dst], T.box); out[it.dst].left = ~(int32_t)order[f]; out[it.dst].right = ~(int32_t)order[f]; continue; }      // (m == 1: the caller does not call)
        int32_t at = it.dst;
        for (uint32_t q = 0; q + 1 < k; q++) {
            Box rest = empty_box(); for (uint32_t z = q; z < k; z++) grow(rest, cref[order[f + z]].box);
            setbox(out[at], rest);
            out[at].left = ~(int32_t)order[f + q];
            if (q + 2 == k) out[at].right = ~(int32_t)order[f + q + 1];
            else { const int32_t nx = (int32_t)out.size(); out.emplace_back(); out[at].right = nx; at = nx; }
        }
    }
}

void build_bvh(const std::vector<float>& wtri, float pad_abs, std::vector<NodeGPU>& nodes, std::vector<uint32_t>& order, uint32_t& max_depth, const BvhBuildOptions& opt) {
    const uint32_t nt = (uint32_t)(wtri.size() / 9);
    const bool TT = getenv("RTX_BUILD_TIMES") != nullptr; auto T0 = std::chrono::steady_clock::now(); auto lap = [&](const char* w) { if (TT) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[build]   bvh2: %-20s %.3f s\n", w, std::chrono::duration<double>(t - T0).count()); T0 = t; } };
    std::vector<Ref> refs(nt);
    Box scene = empty_box();
    for (uint32_t i = 0; i < nt; i++) {
        const float* t = &wtri[(size_t)i * 9];
        for (int a = 0; a < 3; a++) {
            refs[i].box.mn[a] = std::min(t[a], std::min(t[3 + a], t[6 + a]));
            refs[i].box.mx[a] = std::max(t[a], std::max(t[3 + a], t[6 + a]));
        }
        refs[i].tri = i;
        grow(scene, refs[i].box);
    }
    order.clear(); order.reserve(nt);
    std::vector<TmpNode> tn; tn.reserve((size_t)2 * nt + 2);
    max_depth = 0;
    // PLOC (the GPU build's bottom-up half, restated on the host): bottom-up clusters, then the top of the tree — over <= ploc_top clusters, with single clusters as leaves —
    // by the top-down builder and the re-insertion passes (build_cluster_top), and the clusters' subtrees hung in below.  The top of a tree is where every ray passes
    // (5.7 of 12.4 node steps in the first three wide levels on the atrium): it gets the expensive builder, the bottom the parallel one.
    const bool ploc = opt.ploc_radius > 0 && nt > kSmallSceneMaxTris;
    if (ploc) {
        std::vector<PlocNode> pool; std::vector<int32_t> cl;
        ploc_clusters(refs, scene, opt.ploc_radius, std::max(1u, opt.ploc_top), pool, cl);
        lap("ploc clusters");
        if (cl.size() == 1) { tn.emplace_back(); ploc_expand(pool, cl[0], 0, 0u, tn, order, max_depth); }
        else {
            std::vector<float> boxes(cl.size() * 6);
            for (size_t i = 0; i < cl.size(); i++) for (int a = 0; a < 3; a++) { boxes[i * 6 + a] = pool[cl[i]].box.mn[a]; boxes[i * 6 + 3 + a] = pool[cl[i]].box.mx[a]; }
            std::vector<ClusterTopNode> top; build_cluster_top(boxes.data(), (uint32_t)cl.size(), opt, top);
            lap("ploc top");
            struct It { int32_t src, dst; uint32_t depth; };
            std::vector<It> st; tn.emplace_back(); st.push_back({0, 0, 0u});
            while (!st.empty()) {
                const It it = st.back(); st.pop_back();
                const ClusterTopNode N = top[it.src];
                for (int a = 0; a < 3; a++) { tn[it.dst].box.mn[a] = N.mn[a]; tn[it.dst].box.mx[a] = N.mx[a]; }
                const int32_t l = (int32_t)tn.size(); tn.emplace_back(); const int32_t r = (int32_t)tn.size(); tn.emplace_back();
                tn[it.dst].left = l; tn[it.dst].right = r; tn[it.dst].count = 0;
                // (right first on the stack so that the left subtree is expanded first: leaf order = depth-first left to right)
                if (N.right >= 0) st.push_back({N.right, r, it.depth + 1}); 
                if (N.left >= 0) st.push_back({N.left, l, it.depth + 1});
                if (N.left < 0) ploc_expand(pool, cl[~N.left], l, it.depth + 1, tn, order, max_depth);
                if (N.right < 0) ploc_expand(pool, cl[~N.right], r, it.depth + 1, tn, order, max_depth);
            }
        }
        lap("ploc expand");
    } else build_tmp_tree(refs, scene, opt, tn, order, max_depth, lap, wtri.data());
    // ---- leaf order: depth-first, left to right, so that every subtree owns ONE contiguous range of references (collapse_bvh8 merges small subtrees into a
    //      leaf slot by range; the build emits the right side first and the re-insertion moves subtrees) ----
    {
        std::vector<uint32_t> emitted; emitted.reserve(order.size());
        std::vector<int32_t> dfs; dfs.push_back(0);
        while (!dfs.empty() && !tn.empty()) {
            const int32_t i = dfs.back(); dfs.pop_back();
            if (tn[i].count) { const uint32_t f = tn[i].first; tn[i].first = (uint32_t)emitted.size(); for (uint32_t k = 0; k < tn[i].count; k++) emitted.push_back(order[f + k]); }
            else if (tn[i].left >= 0) { dfs.push_back(tn[i].right); dfs.push_back(tn[i].left); }
        }
        if (emitted.size() == order.size()) order.swap(emitted);
    }
    lap("leaf order");
    // ---- breadth-first relayout with children boxes stored in the parent ----
    nodes.clear();
    auto enc_leaf = [](const TmpNode& n) -> int32_t { return (int32_t)~((n.first << 3) | (n.count - 1)); };
    auto put_box = [&](NodeGPU& N, int which, const Box* b) {
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) { mn[a] = b ? b->mn[a] - pad_abs : INFINITY; mx[a] = b ? b->mx[a] + pad_abs : -INFINITY; }
        if (which == 0) { N.a = {mn[0], mn[1], mn[2], mx[0]}; N.b.x = mx[1]; N.b.y = mx[2]; }
        else { N.b.z = mn[0]; N.b.w = mn[1]; N.c = {mn[2], mx[0], mx[1], mx[2]}; }
    };
    if (nt == 0) {
        NodeGPU N{}; put_box(N, 0, nullptr); put_box(N, 1, nullptr);
        N.d = {u2f((uint32_t)kEmptyChild), u2f((uint32_t)kEmptyChild), 0.0f, 0.0f};
        nodes.push_back(N); return;
    }
    if (tn[0].count) {   // root is a leaf: wrap it
        NodeGPU N{}; put_box(N, 0, &tn[0].box); put_box(N, 1, nullptr);
        N.d = {u2f((uint32_t)enc_leaf(tn[0])), u2f((uint32_t)kEmptyChild), 0.0f, 0.0f};
        nodes.push_back(N); return;
    }
    std::vector<int32_t> bfs; bfs.push_back(0);            // internal nodes only
    std::vector<int32_t> gpu_index(tn.size(), -1);
    for (size_t h = 0; h < bfs.size(); h++) {
        const TmpNode& n = tn[bfs[h]];
        gpu_index[bfs[h]] = (int32_t)h;
        if (!tn[n.left].count) bfs.push_back(n.left);
        if (!tn[n.right].count) bfs.push_back(n.right);
    }
    // second pass needs the final indices of children: recompute in the same order
    nodes.resize(bfs.size());
    {
        std::vector<int32_t> idx_of(tn.size(), -1);
        for (size_t h = 0; h < bfs.size(); h++) idx_of[bfs[h]] = (int32_t)h;
        for (size_t h = 0; h < bfs.size(); h++) {
            const TmpNode& n = tn[bfs[h]];
            NodeGPU N{};
            put_box(N, 0, &tn[n.left].box); put_box(N, 1, &tn[n.right].box);
            int32_t c0 = tn[n.left].count ? enc_leaf(tn[n.left]) : idx_of[n.left];
            int32_t c1 = tn[n.right].count ? enc_leaf(tn[n.right]) : idx_of[n.right];
            N.d = {u2f((uint32_t)c0), u2f((uint32_t)c1), 0.0f, 0.0f};
            nodes[h] = N;
        }
    }
}

// Compressed 8-wide collapse.  Which binary subtrees become wide nodes or leaf slots is chosen by a surface-area-heuristic
// dynamic program (below).  The child boxes
// are the binary tree's padded boxes rounded OUTWARD onto the node's byte grid (checked in exact double arithmetic), so the
// wide tree is conservative whenever the binary one is.
bool collapse_bvh8(const std::vector<NodeGPU>& n2, std::vector<Node8GPU>& n8, std::vector<uint32_t>& tri_slots, uint32_t& max_stack,
                   std::vector<uint32_t>* level_start, const BvhBuildOptions& opt) {
    struct Ch { float mn[3], mx[3]; int32_t c; };
    auto get = [](const NodeGPU& N, int which) {
        Ch r;
        if (which == 0) { r.mn[0] = N.a.x; r.mn[1] = N.a.y; r.mn[2] = N.a.z; r.mx[0] = N.a.w; r.mx[1] = N.b.x; r.mx[2] = N.b.y; r.c = (int32_t)f2u(N.d.x); }
        else            { r.mn[0] = N.b.z; r.mn[1] = N.b.w; r.mn[2] = N.c.x; r.mx[0] = N.c.y; r.mx[1] = N.c.z; r.mx[2] = N.c.w; r.c = (int32_t)f2u(N.d.y); }
        return r;
    };
    auto area = [](const Ch& b) { const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; };
    n8.clear(); tri_slots.clear(); max_stack = 0;
    if (level_start) level_start->clear();
    if (n2.empty()) return true;
    // ---- which binary subtrees become wide nodes / leaf slots: surface-area-heuristic dynamic program (Ylitie et al. 2017,
    //      section 3.1).  cost[n][i] = cheapest way to represent binary subtree n with at most i child slots of its wide
    //      parent: as ONE slot (a leaf slot holding all its <= 4 triangles, or an internal slot = a wide node of its own with 8
    //      slots to distribute), or split between its two children.  Greedy "open the largest child" filled 4.1 of 8 slots. ----
    const size_t nn = n2.size();
    // a triangle test is 95 VALU instructions against ~200 of a node step, but triangle steps run with half the lanes of node steps
    // (profiles/r02_traversal.md), so per ray it costs more than the 0.45 the instruction counts say: measured k_trace_closest
    // 23.80 / 21.31 ms (C3 / C5) at 0.45, 23.39 / 20.93 at 0.7, 23.49 / 21.00 at 1.0, 23.53 / 20.95 at 1.5, 24.30 / 21.82 at 0.3
    const double kNodeCost = 1.0, kTriCost = opt.tri_cost;
    // (the program itself, the gathering of a wide node's children and the greedy slot assignment live in rtx_wide.hpp: the GPU build runs the same code)
    struct Sub { uint32_t first; };
    std::vector<Sub> sub(nn);
    std::vector<WideDp> dp(nn);
    auto is_leaf = [](int32_t c) { return c < 0; };
    auto leaf_cnt = [](int32_t c) { return ((~(uint32_t)c) & 7u) + 1u; };
    auto leaf_first = [](int32_t c) { return (~(uint32_t)c) >> 3; };
    auto wb = [](const Ch& c) { WBox b; for (int a = 0; a < 3; a++) { b.mn[a] = c.mn[a]; b.mx[a] = c.mx[a]; } return b; };
    for (size_t n = nn; n-- > 0;) {
        const Ch L = get(n2[n], 0), R = get(n2[n], 1);
        if (L.c == kEmptyChild || R.c == kEmptyChild) {      // only the root may have an unused child (scenes with < 2 leaves)
            if (n != 0) return false;
            sub[n] = Sub{0u}; memset(&dp[n], 0, sizeof(WideDp)); continue;
        }
        if ((L.c >= 0 && (size_t)L.c <= n) || (R.c >= 0 && (size_t)R.c <= n) || (L.c >= 0 && (size_t)L.c >= nn) || (R.c >= 0 && (size_t)R.c >= nn)) return false;
        const WideDpChild dl{area(L), is_leaf(L.c) ? leaf_cnt(L.c) : 0u, is_leaf(L.c) ? nullptr : &dp[(size_t)L.c]}, dr{area(R), is_leaf(R.c) ? leaf_cnt(R.c) : 0u, is_leaf(R.c) ? nullptr : &dp[(size_t)R.c]};
        wide_dp_combine(dl, dr, wbox_area(wbox_union(wb(L), wb(R))), kNodeCost, kTriCost, dp[n]);
        sub[n] = Sub{is_leaf(L.c) ? leaf_first(L.c) : sub[(size_t)L.c].first};
    }
    // children of the wide node made from binary node x, following the recorded decisions
    struct Acc {
        const std::vector<NodeGPU>& n2; const std::vector<Sub>& sub; const std::vector<WideDp>& dp; decltype(get)& get_;
        bool is_leaf(const Ch& c) const { return c.c < 0; }
        void children(const Ch& c, Ch& L, Ch& R) const { L = get_(n2[(size_t)c.c], 0); R = get_(n2[(size_t)c.c], 1); }
        uint8_t choice(const Ch& c, int i) const { return dp[(size_t)c.c].choice[i]; }
        Ch merged(const Ch& c) const { Ch r = c; r.c = (int32_t)~((sub[(size_t)c.c].first << 3) | (dp[(size_t)c.c].prims - 1u)); return r; }      // a leaf slot holding the subtree's <= 4 triangles (contiguous in leaf order)
    };
    const Acc acc{n2, sub, dp, get};
    std::vector<int32_t> src; src.push_back(0);            // binary node behind each wide node, breadth-first
    for (size_t h = 0; h < src.size(); h++) {
        Ch ch[8]; int m = 0;
        const NodeGPU& N = n2[(size_t)src[h]];
        {
            const Ch L = get(N, 0), R = get(N, 1);
            if (L.c == kEmptyChild || R.c == kEmptyChild) { if (L.c != kEmptyChild) ch[m++] = L; if (R.c != kEmptyChild) ch[m++] = R; }
            else { bool internal[8]; m = wide_children(acc, L, R, (int)dp[(size_t)src[h]].choice[8], ch, internal); }
            if (m > 8) return false;
        }
        Node8GPU W{};
        float bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
        // ---- slots: child with the largest projection on an octant's diagonal gets that octant's slot (greedy assignment) ----
        int slot_of[8]; bool slot_used[8] = {false, false, false, false, false, false, false, false};
        {
            WBox cb[8]; for (int k = 0; k < m; k++) cb[k] = wb(ch[k]);
            wide_assign_slots(cb, m, bmn, bmx, slot_of);           // (node bounds + the greedy assignment)
            if (opt.slot_assign == 0) { for (int k = 0; k < m; k++) slot_used[slot_of[k]] = true; }
            else {
                double cost[8][8];
                for (int k = 0; k < m; k++) for (int sl = 0; sl < 8; sl++) {
                    double c = 0.0;
                    for (int a = 0; a < 3; a++) {
                        const double rel = 0.5 * ((double)ch[k].mn[a] + (double)ch[k].mx[a]) - 0.5 * ((double)bmn[a] + (double)bmx[a]);
                        c += ((sl >> a) & 1) ? rel : -rel;
                    }
                    cost[k][sl] = c;
                }
                // the assignment that maximises the summed projections (Ylitie et al. solve it by auction; with eight slots a subset table is exact): best[k][S] = children
                // k.. placed into the free slots of S
                double best[9][256]; int8_t pick[9][256];
                for (int S = 0; S < 256; S++) best[m][S] = 0.0;
                for (int k = m - 1; k >= 0; k--) for (int S = 0; S < 256; S++) {
                    best[k][S] = -1e300; pick[k][S] = -1;
                    if (__builtin_popcount(S) != k) continue;                       // S = slots taken by children 0..k-1
                    for (int sl = 0; sl < 8; sl++) if (!(S & (1 << sl))) {
                        const double nxt = best[k + 1][S | (1 << sl)];
                        if (nxt <= -1e299 && k + 1 < m) continue;
                        const double c = cost[k][sl] + (k + 1 < m ? nxt : 0.0);
                        if (c > best[k][S]) { best[k][S] = c; pick[k][S] = (int8_t)sl; }
                    }
                }
                int S = 0;
                for (int k = 0; k < m; k++) { const int sl = pick[k][S]; slot_of[k] = sl; slot_used[sl] = true; S |= 1 << sl; }
            }
        }
        int child_at[8]; for (int sl = 0; sl < 8; sl++) child_at[sl] = -1;
        for (int k = 0; k < m; k++) child_at[slot_of[k]] = k;
        // ---- byte grid per axis: smallest power of two with 255 steps covering the node ----
        W.px = bmn[0]; W.py = bmn[1]; W.pz = bmn[2];
        int eb[3]; double step[3];
        for (int a = 0; a < 3; a++) {
            const double ext = (double)bmx[a] - (double)bmn[a];
            int e = -120;
            if (ext > 0.0) { e = std::max(-120, (int)std::ilogb(ext / 255.0)); while (std::ldexp(255.0, e) < ext) e++; }
            if (e > 120 || !std::isfinite(ext)) return false;
            eb[a] = e + 127; step[a] = std::ldexp(1.0, e);
        }
        uint32_t imask = 0, trivalid = 0;
        uint8_t qb[6][8];
        for (int sl = 0; sl < 8; sl++) {
            for (int r = 0; r < 6; r++) qb[r][sl] = 0;
            const int k = child_at[sl];
            if (k < 0) continue;
            const float p[3] = {W.px, W.py, W.pz};
            for (int a = 0; a < 3; a++) {
                double qlo = std::floor(((double)ch[k].mn[a] - (double)p[a]) / step[a]), qhi = std::ceil(((double)ch[k].mx[a] - (double)p[a]) / step[a]);
                qlo = std::min(255.0, std::max(0.0, qlo)); qhi = std::min(255.0, std::max(0.0, qhi));
                // exact check: the decoded planes bracket the source box
                if ((double)p[a] + qlo * step[a] > (double)ch[k].mn[a] || (double)p[a] + qhi * step[a] < (double)ch[k].mx[a]) return false;
                qb[a][sl] = (uint8_t)qlo; qb[3 + a][sl] = (uint8_t)qhi;
            }
            if (ch[k].c >= 0) imask |= 1u << sl;
        }
        W.child_base = (uint32_t)src.size();
        for (int sl = 0; sl < 8; sl++) if (imask & (1u << sl)) src.push_back(ch[child_at[sl]].c);
        W.tri_base = (uint32_t)tri_slots.size();
        for (int sl = 0; sl < 8; sl++) {
            const int k = child_at[sl];
            if (k < 0 || ch[k].c >= 0) continue;
            const uint32_t v = ~(uint32_t)ch[k].c, first = v >> 3, cnt = (v & 7u) + 1u;
            if (cnt > 4) return false;
            trivalid |= ((1u << cnt) - 1u) << (4 * sl);
            for (uint32_t t = 0; t < cnt; t++) tri_slots.push_back(first + t);
        }
        W.e_imask = (uint32_t)eb[0] | (uint32_t)eb[1] << 8 | (uint32_t)eb[2] << 16 | imask << 24;
        W.trivalid = trivalid; W.pad = 0;
        for (int r = 0; r < 6; r++) {
            W.q[2 * r]     = (uint32_t)qb[r][0] | (uint32_t)qb[r][1] << 8 | (uint32_t)qb[r][2] << 16 | (uint32_t)qb[r][3] << 24;
            W.q[2 * r + 1] = (uint32_t)qb[r][4] | (uint32_t)qb[r][5] << 8 | (uint32_t)qb[r][6] << 16 | (uint32_t)qb[r][7] << 24;
        }
        n8.push_back(W);
        if (n8.size() >= (1u << 28)) return false;
    }
    std::vector<uint32_t> need(n8.size(), 0);                // children have larger indices: one reverse sweep
    for (size_t i = n8.size(); i-- > 0;) {
        const uint32_t imask = n8[i].e_imask >> 24, nint = (uint32_t)__builtin_popcount(imask);
        uint32_t deep = 0;
        for (uint32_t r = 0; r < nint; r++) deep = std::max(deep, need[(size_t)n8[i].child_base + r]);
        need[i] = (nint > 1 ? 1u : 0u) + deep;
    }
    max_stack = need[0];
    if (level_start) {                                       // breadth-first order: a level is a contiguous index range
        std::vector<uint32_t> level(n8.size(), 0);
        for (size_t i = 0; i < n8.size(); i++) {
            const uint32_t nint = (uint32_t)__builtin_popcount(n8[i].e_imask >> 24);
            for (uint32_t r = 0; r < nint; r++) level[(size_t)n8[i].child_base + r] = level[i] + 1;
        }
        for (size_t i = 0; i < n8.size(); i++) {
            if (i && level[i] < level[i - 1]) return false;
            if (i == 0 || level[i] != level[i - 1]) level_start->push_back((uint32_t)i);
        }
        level_start->push_back((uint32_t)n8.size());
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Host-side REPLAY of the device traversal (csrc/rtx_traverse.hpp: node8_hits / descend8 / traverse, one ray at a time, non-speculative order) on the
// wide tree of a BuiltScene, counting node steps and triangle tests.  Two users: tools/bvh_lab.cpp (builder work judged by work per ray, no GPU) and the
// commit-time probe below.  Scalar float code with the kernels' formulas; not bit-pinned to them (the counts, not the hits, are what it is for).
// ------------------------------------------------------------------------------------------------
namespace {
struct RGrp { uint32_t base, bits; };
struct RTri { uint32_t base, bits, valid; };
constexpr float kRPlaneEps = 2.384185791015625e-07f, kRSlabK = 1.00010002f;
inline void replay_node(const Node8GPU& N, const float o[3], const float idir[3], uint32_t oct, bool ordered, uint32_t oct_order, float tmin, float tbest, RGrp& G, RTri& T) {
    const uint32_t w = N.e_imask;
    const float s[3] = {u2f((w & 0xffu) << 23) * idir[0], u2f((w & 0xff00u) << 15) * idir[1], u2f((w & 0xff0000u) << 7) * idir[2]};
    const float a3[3] = {(N.px - o[0]) * idir[0], (N.py - o[1]) * idir[1], (N.pz - o[2]) * idir[2]};
    uint32_t hits = 0;
    for (int k = 0; k < 8; k++) {
        float lo = tmin, hi = tbest;
        for (int a = 0; a < 3; a++) {
            const uint32_t qlo = (N.q[2 * a + (k >> 2)] >> (8 * (k & 3))) & 0xffu, qhi = (N.q[2 * (3 + a) + (k >> 2)] >> (8 * (k & 3))) & 0xffu;
            const bool neg = (oct >> a) & 1u;
            const float an = fmaf(-fabsf(a3[a]), kRPlaneEps, a3[a]), af = fmaf(fabsf(a3[a]), kRPlaneEps, a3[a]);
            lo = fmaxf(lo, fmaf((float)(neg ? qhi : qlo), s[a], an)); hi = fminf(hi, fmaf((float)(neg ? qlo : qhi), s[a], af));
        }
        if (!(f2u(fmaf(hi, kRSlabK, -lo)) >> 31)) hits |= 1u << k;
    }
    const uint32_t imask = w >> 24;
    uint32_t m = hits & imask;
    if (ordered) { uint32_t pm = 0; for (int j = 0; j < 8; j++) if (m & (1u << (j ^ oct_order))) pm |= 1u << j; m = pm; }
    G.base = N.child_base; G.bits = m | (imask << 8);
    uint32_t x = hits & ~imask, sp = 0;
    for (int k = 0; k < 8; k++) if (x & (1u << k)) sp |= 0xfu << (4 * k);
    T.base = N.tri_base; T.valid = N.trivalid; T.bits = sp & N.trivalid;
}
inline bool replay_tri(const float o[3], const float d[3], const TriGPU& Tg, float tmin, float tmax, float& t) {
    const f3 v0 = mk3(Tg.v0.x, Tg.v0.y, Tg.v0.z), e1 = mk3(Tg.e1.x, Tg.e1.y, Tg.e1.z), e2 = mk3(Tg.e2.x, Tg.e2.y, Tg.e2.z), dd = mk3(d[0], d[1], d[2]);
    const f3 pv = cross(dd, e2);
    const float det = dot(e1, pv);
    if (!(fabsf(det) > Tg.e1.w)) return false;
    const float inv = 1.0f / det;
    const f3 sv = mk3(o[0], o[1], o[2]) - v0;
    const float u = dot(sv, pv) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const f3 q = cross(sv, e1);
    const float v = dot(dd, q) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return false;
    t = dot(e2, q) * inv;
    return t > tmin && t < tmax;
}
}  // namespace

bool replay_tri_test(const float o[3], const float d[3], const TriGPU& Tg, float tmin, float tmax, float& t) { return replay_tri(o, d, Tg, tmin, tmax, t); }

ReplayHit replay_trace(const BuiltScene& B, const float o[3], const float d[3], float tmin, float tmax, bool any, uint32_t any_order, float t_known, std::vector<uint8_t>* seq) {
    float idir[3]; uint32_t oct = 0;
    for (int a = 0; a < 3; a++) { const float ds = fabsf(d[a]) < 1e-30f ? copysignf(1e-30f, d[a]) : d[a]; idir[a] = 1.0f / ds; if (idir[a] < 0.0f) oct |= 1u << a; }
    const bool ordered = !any || any_order != 0;
    const uint32_t oct_order = (any && any_order == 2) ? (oct ^ 7u) : oct;
    ReplayHit H{t_known > 0.0f ? t_known * 1.0000005f : tmax, 0xffffffffu, 0xffffffffu, 0u, 0u};
    if (B.nodes8.empty()) return H;
    RGrp stk[64]; int sp = 0;
    RGrp G{0u, (ordered ? (1u << oct_order) : 1u) | (1u << 8)};
    RTri T{0u, 0u, 0u};
    while (true) {
        if (G.bits & 0xffu) {
            const uint32_t k = (uint32_t)__builtin_ctz(G.bits), rest = G.bits & (G.bits - 1u);
            if ((rest & 0xffu) && sp < 64) stk[sp++] = RGrp{G.base, rest};
            const uint32_t slot = ordered ? (k ^ oct_order) : k;
            const uint32_t idx = G.base + (uint32_t)__builtin_popcount((G.bits >> 8) & ((1u << slot) - 1u));
            replay_node(B.nodes8[idx], o, idir, oct, ordered, oct_order, tmin, H.t, G, T);
            H.steps++;
            if (seq) seq->push_back((uint8_t)__builtin_popcount(T.bits));          // triangles this node step hands to the triangle steps
        }
        while (T.bits) {
            const uint32_t bit = (uint32_t)__builtin_ctz(T.bits);
            T.bits &= T.bits - 1u; H.tris++;
            const uint32_t slot = T.base + (uint32_t)__builtin_popcount(T.valid & ((1u << bit) - 1u));
            float t;
            if (replay_tri(o, d, B.tris8[slot], tmin, tmax, t)) {
                const uint32_t gid = f2u(B.tris8[slot].v0.w);
                if (any) { H.prim = gid; H.slot = slot; H.t = t; if (seq && !seq->empty()) seq->back() = (uint8_t)(seq->back() - __builtin_popcount(T.bits)); return H; }   // (the untested rest of the group is dropped)
                if (t < H.t || (t == H.t && gid < H.prim)) { H.t = t; H.prim = gid; H.slot = slot; }
            }
        }
        if (!(G.bits & 0xffu)) { if (sp == 0) break; G = stk[--sp]; }
    }
    return H;
}

// In which order should an any-hit ray visit the hit children of a node?  Any-hit is existence, so the order changes no result, only how soon an occluder is found:
// slot order (0), nearest octant first (1) or FARTHEST first (2: from the light's end — where a lamp's own housing, or the far faces of a closed emissive mesh, block
// the ray).  Which one wins is a property of the scene and its lights (Bistro-class street: far first -17 % node steps per occluded ray; the atrium under its sky
// quad: slot order), so it is probed once per commit: 2 048 NEE-like segments (a point on a random triangle to a CDF-sampled point on a light) replayed in the three
// orders; the cheapest by the traversal kernels' own cost model wins (node step 205 VALU at 47 of 64 lanes, triangle test 70 at 24), with 5 % hysteresis for order 0.
uint32_t probe_anyhit_order(const BuiltScene& B) {
    if (B.lights.empty() || B.tris8.empty() || B.nodes8.empty() || B.small_nrec) return 0u;
    auto h32 = [](uint32_t a, uint32_t b) { uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15; return h; };
    auto r01 = [&](uint32_t a, uint32_t b) { return (float)(h32(a, b) >> 8) * (1.0f / 16777216.0f); };
    double cost[3] = {0.0, 0.0, 0.0};
    for (uint32_t i = 0; i < 2048u; i++) {
        const TriGPU& Tg = B.tris8[h32(i, 1u) % (uint32_t)B.tris8.size()];
        float u = r01(i, 2u), v = r01(i, 3u); if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
        const f3 p = mk3(Tg.v0.x + u * Tg.e1.x + v * Tg.e2.x, Tg.v0.y + u * Tg.e1.y + v * Tg.e2.y, Tg.v0.z + u * Tg.e1.z + v * Tg.e2.z);
        f3 n = normalize(cross(mk3(Tg.e1.x, Tg.e1.y, Tg.e1.z), mk3(Tg.e2.x, Tg.e2.y, Tg.e2.z)));
        const float xi = r01(i, 4u);
        size_t li = 0; while (li + 1 < B.lights.size() && B.lights[li].cdf < xi) li++;
        const LightGPU& Lg = B.lights[li];
        float a = r01(i, 5u), b = r01(i, 6u); if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; }
        const f3 lp = mk3(Lg.xv[0] + a * (Lg.yv[0] - Lg.xv[0]) + b * (Lg.zv[0] - Lg.xv[0]), Lg.xv[1] + a * (Lg.yv[1] - Lg.xv[1]) + b * (Lg.zv[1] - Lg.xv[1]), Lg.xv[2] + a * (Lg.yv[2] - Lg.xv[2]) + b * (Lg.zv[2] - Lg.xv[2]));
        f3 dir = lp - p;
        if (dot(n, dir) < 0.0f) n = mk3(-n.x, -n.y, -n.z);                      // surfaces are lit from either side
        const f3 org = mk3(p.x + kSBias * n.x, p.y + kSBias * n.y, p.z + kSBias * n.z);
        dir = lp - org;
        const float dist = length(dir);
        if (!(dist > 10.0f * kSBias)) continue;
        const float od[3] = {org.x, org.y, org.z}, dd[3] = {dir.x / dist, dir.y / dist, dir.z / dist};
        for (uint32_t ord = 0; ord < 3u; ord++) {
            const ReplayHit H = replay_trace(B, od, dd, 0.5f * kSBias, dist - 5.0f * kSBias, true, ord);
            cost[ord] += (double)H.steps * (205.0 * 64.0 / 47.0) + (double)H.tris * (70.0 * 64.0 / 24.0);
        }
    }
    uint32_t best = 0;
    for (uint32_t ord = 1; ord < 3u; ord++) if (cost[ord] < 0.95 * cost[0] && cost[ord] < cost[best]) best = ord;      // (an ordered step carries ~3 % more instructions)
    return best;
}

// Coverage bookkeeping shared by the validators of the binary and of the wide tree.  A triangle referenced ONCE must lie inside every box above its
// reference (all three corners).  A triangle that spatial splits handed to several leaves is checked on 28 points (corners, edge thirds, an interior lattice):
// each must lie inside all boxes above ONE of the references — the property the traversal needs (a hit point is found through whichever reference's boxes
// contain it).  Points are evaluated in double; the tolerance covers that evaluation only (boxes of split parts are rounded outward by a float spacing).
CoverCheck::CoverCheck(const std::vector<float>& world_tris9) : w(world_tris9), refs((uint32_t)(world_tris9.size() / 9), 0u) {}
void CoverCheck::count(uint32_t g) { refs[g]++; }
int CoverCheck::add(uint32_t g, const double mn[3], const double mx[3]) {
    if (refs[g] == 1) {
        for (int vtx = 0; vtx < 3; vtx++) for (int a = 0; a < 3; a++) { const double c = w[(size_t)g * 9 + vtx * 3 + a]; if (c < mn[a] || c > mx[a]) return 16; }
        return 0;
    }
    boxes.push_back({g, {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]}});
    return 0;
}
int CoverCheck::finish() {
    for (uint32_t r : refs) if (!r) return 17;
    std::stable_sort(boxes.begin(), boxes.end(), [](const Part& a, const Part& b) { return a.tri < b.tri; });
    for (size_t i = 0; i < boxes.size();) {
        size_t j = i; while (j < boxes.size() && boxes[j].tri == boxes[i].tri) j++;
        const float* t = &w[(size_t)boxes[i].tri * 9];
        double scale = 1.0; for (int k = 0; k < 9; k++) scale = std::max(scale, std::fabs((double)t[k]));
        const double tol = 1e-12 * scale;
        for (int a = 0; a <= 6; a++) for (int b = 0; a + b <= 6; b++) {
            const double u = a / 6.0, v = b / 6.0, q = 1.0 - u - v;
            const double pt[3] = {q * t[0] + u * t[3] + v * t[6], q * t[1] + u * t[4] + v * t[7], q * t[2] + u * t[5] + v * t[8]};
            bool in = false;
            for (size_t k = i; k < j && !in; k++) { const double* bx = boxes[k].b; in = pt[0] >= bx[0] - tol && pt[1] >= bx[1] - tol && pt[2] >= bx[2] - tol && pt[0] <= bx[3] + tol && pt[1] <= bx[4] + tol && pt[2] <= bx[5] + tol; }
            if (!in) return 24;
        }
        i = j;
    }
    return 0;
}

// the compressed 8-wide collapse: same coverage properties, checked on the DECODED byte-grid boxes of the wide nodes
int validate_bvh8(const std::vector<float>& w, const std::vector<Node8GPU>& nodes, const std::vector<uint32_t>& order,
                  const std::vector<uint32_t>& tri_slots, uint32_t* max_stack_seen) {
    const uint32_t ntris = (uint32_t)(w.size() / 9), nrefs = (uint32_t)tri_slots.size();
    if (order.size() != nrefs || nrefs < ntris) return 20;
    struct It { uint32_t node; double mn[3], mx[3]; uint32_t pushes; };
    std::vector<uint8_t> used(nrefs, 0), visited(nodes.size(), 0);
    if (nodes.empty()) return ntris ? 10 : 0;
    CoverCheck cover(w);
    for (uint32_t s = 0; s < nrefs; s++) { if (tri_slots[s] >= nrefs || order[tri_slots[s]] >= ntris) return 14; cover.count(order[tri_slots[s]]); }
    std::vector<It> st;
    const double inf = INFINITY;
    st.push_back({0u, {-inf, -inf, -inf}, {inf, inf, inf}, 0u});
    uint32_t deepest = 0;
    while (!st.empty()) {
        const It it = st.back(); st.pop_back();
        if (it.node >= nodes.size()) return 13;
        if (visited[it.node]) return 11;
        visited[it.node] = 1;
        const Node8GPU& N = nodes[it.node];
        const double p[3] = {N.px, N.py, N.pz};
        double step[3];
        for (int a = 0; a < 3; a++) { const int eb = (int)((N.e_imask >> (8 * a)) & 0xffu); if (eb < 1 || eb > 254) return 21; step[a] = std::ldexp(1.0, eb - 127); }
        const uint32_t imask = N.e_imask >> 24;
        const uint32_t nint = (uint32_t)__builtin_popcount(imask);
        const uint32_t pushes = it.pushes + (nint > 1 ? 1u : 0u);
        deepest = std::max(deepest, pushes);
        uint32_t rank = 0, tri_at = N.tri_base;
        for (int sl = 0; sl < 8; sl++) {
            const uint32_t nib = (N.trivalid >> (4 * sl)) & 0xfu;
            const bool internal = (imask >> sl) & 1u;
            if (internal && nib) return 22;
            if (!internal && !nib) continue;
            double mn[3], mx[3];
            for (int a = 0; a < 3; a++) {
                const uint32_t qlo = (N.q[2 * a + (sl >> 2)] >> (8 * (sl & 3))) & 0xffu, qhi = (N.q[2 * (3 + a) + (sl >> 2)] >> (8 * (sl & 3))) & 0xffu;
                mn[a] = std::max(p[a] + qlo * step[a], it.mn[a]); mx[a] = std::min(p[a] + qhi * step[a], it.mx[a]);
            }
            if (internal) {
                const uint32_t c = N.child_base + rank++;
                if (c <= it.node) return 12;                                     // breadth-first: children after parents
                It nx; nx.node = c; nx.pushes = pushes;
                for (int a = 0; a < 3; a++) { nx.mn[a] = mn[a]; nx.mx[a] = mx[a]; }
                st.push_back(nx);
            } else {
                if (nib != 1 && nib != 3 && nib != 7 && nib != 15) return 23;
                const uint32_t cnt = (uint32_t)__builtin_popcount(nib);
                for (uint32_t k = 0; k < cnt; k++, tri_at++) {
                    if (tri_at >= nrefs) return 14;
                    if (used[tri_at]) return 15;                                 // every leaf entry belongs to one leaf slot
                    used[tri_at] = 1;
                    if (int r = cover.add(order[tri_slots[tri_at]], mn, mx)) return r;
                }
            }
        }
    }
    for (uint32_t i = 0; i < nrefs; i++) if (!used[i]) return 17;
    for (size_t i = 0; i < nodes.size(); i++) if (!visited[i]) return 18;
    if (int r = cover.finish()) return r;
    if (max_stack_seen) *max_stack_seen = deepest;
    return 0;
}

}  // namespace rtx
