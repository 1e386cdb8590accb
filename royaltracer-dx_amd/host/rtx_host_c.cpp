// rtx_host_c.cpp — C entry points of include/rtx_host.h over the C++ host layer.
#include "../../include/rtx_host.h"
#include <cstring>
#include <type_traits>
#include <exception>
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include <string>
#include "../csrc/rtx_scene_host.hpp"
#include "ObjLoader.h"
#include "Renderer.h"
#include "ImageIO.h"
#include "Scenes.h"
#include "manipulator.h"

struct rtxh_scene { Scene s; std::string cache_path; };     // cache_path: the scene came from rtxh_scene_load; rtxh_scene_upload hands the FILE to the context (no rebuild)
struct rtxh_renderer { Renderer* r; };
static thread_local std::string g_err;

template <class F> static rtxh_scene* guarded(F&& f) {
    try { rtxh_scene* h = new rtxh_scene(); h->s = f(); return h; }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
template <class F> static int guarded_rc(F&& f) {
    try { f(); return RTX_OK; } catch (const std::exception& e) { g_err = e.what(); return RTX_ERR_INVALID; }
}

extern "C" {

rtxh_scene* rtxh_scene_cornell(void) { return guarded([] { return MakeCornellBox(); }); }
rtxh_scene* rtxh_scene_sponza_class(uint32_t t, uint32_t seed) { return guarded([=] { return MakeSponzaClass(t, seed); }); }
rtxh_scene* rtxh_scene_bistro_class(uint32_t t, uint32_t seed) { return guarded([=] { return MakeBistroClass(t, seed); }); }
rtxh_scene* rtxh_scene_sponza_class_hard(uint32_t t, uint32_t seed) { return guarded([=] { return MakeSponzaClass(t, seed, true); }); }
rtxh_scene* rtxh_scene_bistro_class_hard(uint32_t t, uint32_t seed) { return guarded([=] { return MakeBistroClass(t, seed, true); }); }
rtxh_scene* rtxh_scene_from_obj(const char* const* files, uint32_t n, const char* mtl_dir) {
    std::vector<std::string> f; for (uint32_t i = 0; i < n; i++) f.emplace_back(files[i]);
    std::string dir = mtl_dir ? mtl_dir : "./";
    return guarded([&] { return LoadObjScene(f, dir); });
}
void rtxh_scene_free(rtxh_scene* s) { delete s; }
const char* rtxh_last_error(void) { return g_err.c_str(); }

uint32_t rtxh_scene_num_materials(const rtxh_scene* s) { return (uint32_t)s->s.materials.size(); }
const void* rtxh_scene_materials(const rtxh_scene* s) { return s->s.materials.data(); }
uint32_t rtxh_scene_num_textures(const rtxh_scene* s) { return (uint32_t)s->s.textures.size(); }
const char* rtxh_scene_texture(const rtxh_scene* s, uint32_t i) { return i < s->s.textures.size() ? s->s.textures[i].c_str() : nullptr; }
int rtxh_scene_material_ext(const rtxh_scene* s, uint32_t i, rtxh_material_ext* out) {
    if (!out || i >= s->s.materialExt.size()) return RTX_ERR_INVALID;
    const MaterialExt& x = s->s.materialExt[i];
    static_assert(RTXH_NUM_MAP_SLOTS == kNumMapSlots, "map slot count");
    out->Ni = x.Ni; out->Ns = x.Ns; out->Pcr = x.Pcr; out->aniso = x.aniso; out->anisor = x.anisor; out->illum = x.illum;
    for (int k = 0; k < 3; k++) { out->Ka[k] = x.Ka[k]; out->Tf[k] = x.Tf[k]; }
    for (int k = 0; k < kNumMapSlots; k++) out->map[k] = x.map[k];
    return RTX_OK;
}
uint32_t rtxh_scene_num_meshes(const rtxh_scene* s) { return (uint32_t)s->s.models.size(); }
int rtxh_scene_mesh(const rtxh_scene* s, uint32_t i, const void** v, uint32_t* nv, const uint32_t** idx, uint32_t* nidx, const uint32_t** mids) {
    if (i >= s->s.models.size()) return RTX_ERR_INVALID;
    const SceneModel& m = s->s.models[i];
    *v = m.vertices.data(); *nv = (uint32_t)m.vertices.size(); *idx = m.indices.data(); *nidx = (uint32_t)m.indices.size(); *mids = m.materialIDs.data();
    return RTX_OK;
}
uint32_t rtxh_scene_num_instances(const rtxh_scene* s) { return (uint32_t)s->s.instances.size(); }
int rtxh_scene_instance(const rtxh_scene* s, uint32_t i, uint32_t* mesh, float o2w[16]) {
    if (i >= s->s.instances.size()) return RTX_ERR_INVALID;
    *mesh = s->s.instances[i].model; memcpy(o2w, s->s.instances[i].transform.data(), 64);
    return RTX_OK;
}
uint64_t rtxh_scene_num_triangles(const rtxh_scene* s) { return s->s.triangles(); }
int rtxh_scene_camera(const rtxh_scene* s, float eye[3], float center[3], float up[3], float* fov, float* zn, float* zf) {
    eye[0] = s->s.eye.x; eye[1] = s->s.eye.y; eye[2] = s->s.eye.z;
    center[0] = s->s.center.x; center[1] = s->s.center.y; center[2] = s->s.center.z;
    up[0] = s->s.up.x; up[1] = s->s.up.y; up[2] = s->s.up.z;
    *fov = s->s.fovY_deg; *zn = s->s.zn; *zf = s->s.zf;
    return RTX_OK;
}
int rtxh_scene_set_camera(rtxh_scene* s, const float eye[3], const float center[3], const float up[3]) {
    s->s.eye = XMFLOAT3(eye[0], eye[1], eye[2]); s->s.center = XMFLOAT3(center[0], center[1], center[2]); s->s.up = XMFLOAT3(up[0], up[1], up[2]);
    return RTX_OK;
}
int rtxh_scene_view_proj(const rtxh_scene* s, float aspect, float view[16], float proj[16]) { SceneViewProj(s->s, aspect, view, proj); return RTX_OK; }
int rtxh_scene_upload(const rtxh_scene* s, rtx_ctx* c, float aspect) {
    if (s->cache_path.empty()) return UploadScene(s->s, c, aspect);
    int r = rtx_load_scene_cache(c, s->cache_path.c_str());               // prebuilt BVH / shading records / LUTs straight from the file
    if (r) return r;
    float view[16], proj[16];
    SceneViewProj(s->s, aspect, view, proj);
    return rtx_set_camera(c, view, proj);
}

// the host scene with everything rtx_commit_scene would derive from it, in one file (format: csrc/rtx_scene_cache.cpp).
// Saving builds on the host only (no GPU needed); a loaded scene reads like any other (materials, meshes, instances, camera) and uploads
// without a BVH build.
int rtxh_scene_save(const rtxh_scene* s, const char* path) {
    rtx::SceneHost H; rtx::BuiltScene B;
    const Scene& sc = s->s;
    bool ok = H.set_materials(sc.materials.data(), (uint32_t)sc.materials.size());
    for (const SceneModel& m : sc.models) { uint32_t id; ok = ok && H.add_mesh(m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id); }
    for (const SceneInstance& in : sc.instances) { uint32_t id; ok = ok && H.add_instance(in.model, in.transform.data(), &id); }
    if (!ok || !H.build(B)) { g_err = H.err; return RTX_ERR_INVALID; }
    const float cam[12] = {sc.eye.x, sc.eye.y, sc.eye.z, sc.center.x, sc.center.y, sc.center.z, sc.up.x, sc.up.y, sc.up.z, sc.fovY_deg, sc.zn, sc.zf};
    std::string err;
    rtx::CacheAux aux;                                                // MaterialExt + texture names ride along (an OBJ / MTL scene answers the accessors after a load, too)
    static_assert(std::is_trivially_copyable<MaterialExt>::value, "MaterialExt is stored as raw records");
    aux.rec_bytes = (uint32_t)sizeof(MaterialExt);
    aux.records.resize(sc.materialExt.size() * sizeof(MaterialExt));
    if (!sc.materialExt.empty()) memcpy(aux.records.data(), sc.materialExt.data(), aux.records.size());
    for (const std::string& t : sc.textures) { aux.text.insert(aux.text.end(), t.begin(), t.end()); aux.text.push_back('\0'); }
    if (!rtx::save_scene_cache(H, B, path, err, cam, &aux)) { g_err = err; return RTX_ERR_INVALID; }
    return RTX_OK;
}
rtxh_scene* rtxh_scene_load(const char* path) {
    rtx::SceneHost H; rtx::BuiltScene B; float cam[12]; std::string err; rtx::CacheAux aux;
    if (!rtx::load_scene_cache(path, H, B, err, cam, &aux)) { g_err = err; return nullptr; }
    const size_t next = aux.rec_bytes == sizeof(MaterialExt) ? aux.records.size() / sizeof(MaterialExt) : 0;
    if ((!aux.records.empty() && (aux.rec_bytes != sizeof(MaterialExt) || next != H.mats128.size() / 32)) || (!aux.text.empty() && aux.text.back() != '\0')) {
        g_err = "scene cache: material extension records do not match this library"; return nullptr;
    }
    rtxh_scene* h = new rtxh_scene();
    Scene& sc = h->s;
    sc.materialExt.resize(next);
    if (next) memcpy((void*)sc.materialExt.data(), aux.records.data(), next * sizeof(MaterialExt));
    for (size_t at = 0; at < aux.text.size();) { const std::string t(&aux.text[at]); at += t.size() + 1; sc.textures.push_back(t); }
    for (MaterialExt& e : sc.materialExt) for (int& m : e.map) if (m < -1 || m >= (int)sc.textures.size()) m = -1;      // ids stay inside the name list
    sc.name = path; h->cache_path = path;
    const size_t nmat = H.mats128.size() / 32;
    sc.materials.assign(nmat, Material(XMFLOAT4(1, 1, 1, 1), XMFLOAT4(0, 0, 0, 0)));
    static_assert(sizeof(Material) == 128 && sizeof(Vertex) == 28, "reference record sizes");
    if (nmat) memcpy((void*)sc.materials.data(), H.mats128.data(), nmat * 128);
    sc.models.resize(H.meshes.size());
    for (size_t i = 0; i < H.meshes.size(); i++) {
        const rtx::MeshHost& m = H.meshes[i]; SceneModel& o = sc.models[i];
        o.vertices.assign(m.verts.size() / 7, Vertex(XMFLOAT3(0, 0, 0), XMFLOAT4(0, 0, 0, 0)));
        if (!m.verts.empty()) memcpy((void*)o.vertices.data(), m.verts.data(), m.verts.size() * 4);
        o.indices = m.idx;
        o.materialIDs.assign(H.matids.begin() + m.matid_base, H.matids.begin() + m.matid_base + (long)m.idx.size());
    }
    for (const rtx::InstHost& in : H.insts) { SceneInstance si; si.model = in.mesh; memcpy((void*)si.transform.data(), in.o2w, 64); sc.instances.push_back(si); }
    sc.eye = XMFLOAT3(cam[0], cam[1], cam[2]); sc.center = XMFLOAT3(cam[3], cam[4], cam[5]); sc.up = XMFLOAT3(cam[6], cam[7], cam[8]);
    sc.fovY_deg = cam[9]; sc.zn = cam[10]; sc.zf = cam[11];
    return h;
}

void rtxh_lookat(const float e[3], const float c[3], const float u[3], float view[16]) {
    nv_helpers_dx12::Manipulator m;
    m.setLookat(XMFLOAT3(e), XMFLOAT3(c), XMFLOAT3(u));
    memcpy(view, m.getMatrix(), 64);
}
void rtxh_perspective_fov_rh(float fovy, float aspect, float zn, float zf, float proj[16]) {
    XMMATRIX P = XMMatrixPerspectiveFovRH(fovy, aspect, zn, zf); memcpy(proj, P.data(), 64);
}
void rtxh_generate_ess_lut(float roughness, float lut[16]) {
    Material m(XMFLOAT4(1, 1, 1, 1), XMFLOAT4(roughness, 0, 0, 0)); GenerateEssLUT(m); memcpy(lut, m.LUT, 64);
}
void rtxh_mat4_inverse(const float m[16], float out[16]) { rtx::mat4_inverse(m, out); }
float rtxh_half_round(float x) { return rtx::half_round(x); }

// every triangle in exactly one leaf, every child box contains its subtree
static int bvh_validate(const std::vector<float>& w, const std::vector<rtx::NodeGPU>& nodes, const std::vector<uint32_t>& order, uint32_t* max_leaf_out) {
    const uint32_t ntris = (uint32_t)(w.size() / 9);
    if (order.size() < ntris) return 1;
    rtx::CoverCheck cover(w);
    for (uint32_t g : order) { if (g >= ntris) return 2; cover.count(g); }
    struct It { int32_t child; float mn[3], mx[3]; };
    std::vector<uint8_t> used(order.size(), 0);
    uint32_t max_leaf = 0;
    std::vector<It> st;
    auto push_children = [&](const rtx::NodeGPU& N, const float* pmn, const float* pmx) {
        It a, b;
        a.child = (int32_t)rtx::f2u(N.d.x); b.child = (int32_t)rtx::f2u(N.d.y);
        float amn[3] = {N.a.x, N.a.y, N.a.z}, amx[3] = {N.a.w, N.b.x, N.b.y}, bmn[3] = {N.b.z, N.b.w, N.c.x}, bmx[3] = {N.c.y, N.c.z, N.c.w};
        for (int k = 0; k < 3; k++) { a.mn[k] = std::max(amn[k], pmn[k]); a.mx[k] = std::min(amx[k], pmx[k]); b.mn[k] = std::max(bmn[k], pmn[k]); b.mx[k] = std::min(bmx[k], pmx[k]); }
        if (a.child != rtx::kEmptyChild) st.push_back(a);
        if (b.child != rtx::kEmptyChild) st.push_back(b);
    };
    const float inf = INFINITY; float rmn[3] = {-inf, -inf, -inf}, rmx[3] = {inf, inf, inf};
    push_children(nodes[0], rmn, rmx);
    while (!st.empty()) {
        It it = st.back(); st.pop_back();
        if (it.child >= 0) { if ((size_t)it.child >= nodes.size()) return 3; push_children(nodes[it.child], it.mn, it.mx); continue; }
        uint32_t v = ~(uint32_t)it.child, first = v >> 3, cnt = (v & 7u) + 1u;
        max_leaf = std::max(max_leaf, cnt);
        for (uint32_t k = 0; k < cnt; k++) {
            if (first + k >= order.size()) return 4;
            if (used[first + k]) return 5;                         // a leaf entry belongs to one leaf
            used[first + k] = 1;
            const double mn[3] = {it.mn[0], it.mn[1], it.mn[2]}, mx[3] = {it.mx[0], it.mx[1], it.mx[2]};
            if (cover.add(order[first + k], mn, mx)) return 6;
        }
    }
    for (uint8_t u : used) if (!u) return 7;
    if (int r = cover.finish()) return r == 17 ? 7 : 8;
    if (max_leaf_out) *max_leaf_out = max_leaf;
    return 0;
}

// build + collapse; returns 0 when the wide tree covers every triangle exactly once inside its decoded boxes and the reported
// stack bound is what the deepest root-to-leaf path can push
int rtxh_bvh8_check(const float* wt, uint32_t ntris, uint32_t* nodes8_out, uint32_t* stack_out) {
    std::vector<float> w(wt, wt + (size_t)ntris * 9);
    std::vector<rtx::NodeGPU> nodes; std::vector<uint32_t> order; uint32_t depth = 0;
    rtx::build_bvh(w, 0.0f, nodes, order, depth);
    std::vector<rtx::Node8GPU> n8; std::vector<uint32_t> slots, levels; uint32_t stack = 0;
    if (!rtx::collapse_bvh8(nodes, n8, slots, stack, &levels)) return 30;
    if (nodes8_out) *nodes8_out = (uint32_t)n8.size();
    if (stack_out) *stack_out = stack;
    uint32_t seen = 0;
    if (int r = rtx::validate_bvh8(w, n8, order, slots, &seen)) return r;
    if (seen != stack) return 19;
    // the level table the GPU refit sweeps bottom-up: contiguous ranges, and every internal child lives exactly one level below
    if (levels.size() < 2 || levels.front() != 0 || levels.back() != n8.size()) return 31;
    for (size_t l = 0; l + 1 < levels.size(); l++) {
        if (levels[l] >= levels[l + 1]) return 32;
        for (uint32_t i = levels[l]; i < levels[l + 1]; i++) {
            const uint32_t nint = (uint32_t)__builtin_popcount(n8[i].e_imask >> 24);
            for (uint32_t r = 0; r < nint; r++) {
                const uint32_t c = n8[i].child_base + r;
                if (l + 2 >= levels.size() || c < levels[l + 1] || c >= levels[l + 2]) return 33;
            }
        }
    }
    return 0;
}

// the device traversal replayed on the host (csrc/rtx_scene_host.cpp: replay_trace) over the wide tree the CURRENT builder options give for these triangles, leaf
// boxes padded as rtx_commit_scene pads them: out4 = (t, node steps, triangle tests, global id bits or 0xffffffff) per ray (o.xyz, tmin, d.xyz, tmax)
int rtxh_bvh_replay(const float* wt, uint32_t ntris, const float* rays8, uint32_t nrays, int any, uint32_t any_order, float* out4, uint32_t* refs_out) {
    std::vector<float> w(wt, wt + (size_t)ntris * 9);
    float scale = 1.0f; for (float x : w) scale = std::max(scale, std::fabs(x));
    rtx::BuiltScene B; uint32_t depth = 0;
    rtx::build_bvh(w, 2e-6f * scale, B.nodes, B.leaf_order, depth);
    if (!rtx::collapse_bvh8(B.nodes, B.nodes8, B.tri_slots8, B.stack8)) return 30;
    B.tris8.resize(B.tri_slots8.size());
    for (size_t i = 0; i < B.tri_slots8.size(); i++) {
        const uint32_t g = B.leaf_order[B.tri_slots8[i]]; const float* t = &w[(size_t)g * 9];
        const rtx::f3 e1 = rtx::mk3(t[3] - t[0], t[4] - t[1], t[5] - t[2]), e2 = rtx::mk3(t[6] - t[0], t[7] - t[1], t[8] - t[2]);
        B.tris8[i].v0 = {t[0], t[1], t[2], rtx::u2f(g)}; B.tris8[i].e1 = {e1.x, e1.y, e1.z, rtx::tri_det_floor(e1, e2)}; B.tris8[i].e2 = {e2.x, e2.y, e2.z, 0.0f};
    }
    if (refs_out) *refs_out = (uint32_t)B.tris8.size();
    for (uint32_t i = 0; i < nrays; i++) {
        const float* r = rays8 + (size_t)i * 8;
        const rtx::ReplayHit H = rtx::replay_trace(B, r, r + 4, r[3], r[7], any != 0, any_order);
        out4[(size_t)i * 4] = H.t; out4[(size_t)i * 4 + 1] = (float)H.steps; out4[(size_t)i * 4 + 2] = (float)H.tris; out4[(size_t)i * 4 + 3] = rtx::u2f(H.prim);
    }
    return 0;
}

int rtxh_bvh_option(const char* key, double value) { return rtx::bvh_build_option(key, value) ? RTX_OK : RTX_ERR_INVALID; }

// shape of the wide tree: hist[0..4] = leaf slots holding 0 (unused slot) / 1 / 2 / 3 / 4 triangles, hist[5] = internal child slots
int rtxh_bvh8_stats(const float* wt, uint32_t ntris, uint32_t hist[6], uint32_t* nodes8_out) {
    std::vector<float> w(wt, wt + (size_t)ntris * 9);
    std::vector<rtx::NodeGPU> nodes; std::vector<uint32_t> order; uint32_t depth = 0;
    rtx::build_bvh(w, 0.0f, nodes, order, depth);
    std::vector<rtx::Node8GPU> n8; std::vector<uint32_t> slots; uint32_t stack = 0;
    if (!rtx::collapse_bvh8(nodes, n8, slots, stack)) return 30;
    for (int i = 0; i < 6; i++) hist[i] = 0;
    for (const rtx::Node8GPU& N : n8) {
        const uint32_t imask = N.e_imask >> 24;
        for (int sl = 0; sl < 8; sl++) {
            if ((imask >> sl) & 1u) { hist[5]++; continue; }
            hist[__builtin_popcount((N.trivalid >> (4 * sl)) & 0xfu)]++;
        }
    }
    if (nodes8_out) *nodes8_out = (uint32_t)n8.size();
    return 0;
}

int rtxh_bvh_check(const float* wt, uint32_t ntris, uint32_t* nodes_out, uint32_t* depth_out, uint32_t* max_leaf_out) {
    std::vector<float> w(wt, wt + (size_t)ntris * 9);
    std::vector<rtx::NodeGPU> nodes; std::vector<uint32_t> order; uint32_t depth = 0;
    rtx::build_bvh(w, 0.0f, nodes, order, depth);
    if (nodes_out) *nodes_out = (uint32_t)nodes.size();
    if (depth_out) *depth_out = depth;
    return bvh_validate(w, nodes, order, max_leaf_out);
}

// build on `before`, refit (topology kept) to `after`: the refitted boxes must contain the moved triangles
int rtxh_bvh_refit_check(const float* before, const float* after, uint32_t ntris) {
    std::vector<float> a(before, before + (size_t)ntris * 9), b(after, after + (size_t)ntris * 9);
    std::vector<rtx::NodeGPU> nodes; std::vector<uint32_t> order; uint32_t depth = 0;
    rtx::build_bvh(a, 0.0f, nodes, order, depth);
    rtx::refit_bvh(b, 0.0f, nodes, order);
    if (int r = bvh_validate(b, nodes, order, nullptr)) return r;
    std::vector<rtx::Node8GPU> n8; std::vector<uint32_t> slots; uint32_t stack = 0;
    if (!rtx::collapse_bvh8(nodes, n8, slots, stack)) return 30;
    return rtx::validate_bvh8(b, n8, order, slots, nullptr);
}

// the tiny-scene pre-test records as rtx_commit_scene builds them (for host-side conservativeness tests)
static bool build_for_inspection(const rtxh_scene* sc, rtx::BuiltScene& B);
// records [0, *nocc_out) can lie between two scene points; the rest are faces of the scene's convex hull (skipped by NEE segments)
int rtxh_scene_small_occluders(const rtxh_scene* sc, uint32_t* nocc_out) {
    rtx::BuiltScene B;
    if (!sc || !nocc_out || !build_for_inspection(sc, B)) return RTX_ERR_INVALID;
    *nocc_out = B.small_nocc;
    return RTX_OK;
}
// what rtx_commit_scene's probe would choose for this scene (csrc/rtx_scene_host.cpp: probe_anyhit_order), and the replayed cost of the probe's segments per order
int rtxh_scene_anyhit_order(const rtxh_scene* sc, uint32_t* order_out) {
    rtx::BuiltScene B;
    if (!sc || !order_out || !build_for_inspection(sc, B)) return RTX_ERR_INVALID;
    *order_out = B.any_order;
    return RTX_OK;
}
static bool build_for_inspection(const rtxh_scene* sc, rtx::BuiltScene& B) {
    rtx::SceneHost H;
    const Scene& s = sc->s;
    if (!H.set_materials(s.materials.data(), (uint32_t)s.materials.size())) return false;
    for (const SceneModel& m : s.models) { uint32_t id; if (!H.add_mesh(m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id)) return false; }
    for (const SceneInstance& in : s.instances) { uint32_t id; if (!H.add_instance(in.model, in.transform.data(), &id)) return false; }
    return H.build(B);
}
int rtxh_scene_small_records(const rtxh_scene* sc, float* recs20, int32_t* tri_ids2, uint32_t max_recs, uint32_t* nrec_out, float* delta_out, float* cm_out) {
    rtx::SceneHost H; rtx::BuiltScene B;
    const Scene& s = sc->s;
    if (!H.set_materials(s.materials.data(), (uint32_t)s.materials.size())) return RTX_ERR_INVALID;
    for (const SceneModel& m : s.models) { uint32_t id; if (!H.add_mesh(m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id)) return RTX_ERR_INVALID; }
    for (const SceneInstance& in : s.instances) { uint32_t id; if (!H.add_instance(in.model, in.transform.data(), &id)) return RTX_ERR_INVALID; }
    if (!H.build(B)) return RTX_ERR_INVALID;
    if (nrec_out) *nrec_out = B.small_nrec;
    if (delta_out) *delta_out = 0.0f;                     // the tolerance is already folded into the records' edge constants (B.small_delta)
    if (cm_out) *cm_out = B.small_cm;
    for (uint32_t r = 0; r < B.small_nrec && r < max_recs; r++) {
        for (int row = 0; row < 20; row++) recs20[(size_t)r * 20 + row] = B.small_recs[r / 2].r[row][r & 1];
        for (int h = 0; h < 2; h++) { uint32_t g = rtx::f2u(B.small_tris[2 * r + h].v0.w); tri_ids2[2 * r + h] = g == rtx::kMissPrim ? -1 : (int32_t)g; }
    }
    return RTX_OK;
}

rtxh_renderer* rtxh_renderer_create(uint32_t w, uint32_t h, const char* name, int device) {
    rtxh_renderer* r = new rtxh_renderer(); r->r = new Renderer(w, h, name ? name : "rtx"); r->r->SetDevice(device); return r;
}
int rtxh_renderer_set_scene(rtxh_renderer* r, const rtxh_scene* s) { r->r->SetScene(s->s); return RTX_OK; }
rtx_params* rtxh_renderer_params(rtxh_renderer* r) { return &r->r->Params(); }
int rtxh_renderer_set_mode(rtxh_renderer* r, int mode) {
    if (mode != 0 && mode != 1) { g_err = "renderer mode must be 0 (path tracer) or 1 (ReSTIR frame)"; return RTX_ERR_INVALID; }
    r->r->SetMode(mode ? Renderer::Mode::ReSTIR : Renderer::Mode::PathTracer); return RTX_OK;
}
rtx_params* rtxh_renderer_restir_params(rtxh_renderer* r) { return &r->r->RestirParams(); }
rtx_ctx* rtxh_renderer_context(rtxh_renderer* r) { return r->r->Context(); }
int rtxh_renderer_on_init(rtxh_renderer* r) { return guarded_rc([&] { r->r->OnInit(); }); }
int rtxh_renderer_on_update(rtxh_renderer* r) { return guarded_rc([&] { r->r->OnUpdate(); }); }
int rtxh_renderer_set_instance_transform(rtxh_renderer* r, uint32_t instance, const float o2w[16]) { return guarded_rc([&] { XMMATRIX m; memcpy(m.data(), o2w, 64); r->r->SetInstanceTransform(instance, m); }); }
int rtxh_renderer_on_render(rtxh_renderer* r) { return guarded_rc([&] { r->r->OnRender(); }); }
int rtxh_renderer_read_accum(rtxh_renderer* r, float* out, size_t bytes) {
    return guarded_rc([&] { auto v = r->r->ReadAccumulation(); if (bytes < v.size() * 4) throw std::runtime_error("buffer too small"); memcpy(out, v.data(), v.size() * 4); });
}
int rtxh_renderer_read_output(rtxh_renderer* r, uint8_t* out, size_t bytes) {
    return guarded_rc([&] { auto v = r->r->ReadOutput(); if (bytes < v.size()) throw std::runtime_error("buffer too small"); memcpy(out, v.data(), v.size()); });
}
int rtxh_renderer_on_key_up(rtxh_renderer* h, uint8_t key) { return guarded_rc([&] { h->r->OnKeyUp(key); }); }
uint32_t rtxh_renderer_display_layer(const rtxh_renderer* h) { return h->r->CurrentDisplayLayer(); }
void rtxh_renderer_destroy(rtxh_renderer* r) { if (r) { delete r->r; delete r; } }

// image writers of the headless display path (host/ImageIO.h)
int rtxh_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) { return path && WritePNG(path, rgba8, width, height) ? RTX_OK : RTX_ERR_INVALID; }
int rtxh_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) { return path && WritePPM(path, rgba8, width, height) ? RTX_OK : RTX_ERR_INVALID; }
int rtxh_write_exr(const char* path, const float* rgba32f, uint32_t width, uint32_t height) { return path && WriteEXR(path, rgba32f, width, height) ? RTX_OK : RTX_ERR_INVALID; }

}  // extern "C"
