#include "ObjLoader.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>
#include "../csrc/rtx_math.hpp"

using rtx::f3; using rtx::mk3;

namespace {
constexpr float PI_HOST = 3.14159265359f;     // ObjLoader.h:22

// ---- host-side GGX helpers of the LUT generator, ObjLoader.h:140-289 ----
inline float D_GGX_h(float NdotH, float roughness) {
    float alpha = roughness * roughness, alpha2 = alpha * alpha, NdotH2 = NdotH * NdotH;
    float denom = std::max(NdotH2 * (alpha2 - 1.0f) + 1.0f, 1e-7f);
    return alpha2 / (PI_HOST * denom * denom);
}
inline float G1_h(float NdotV, float alpha) {
    float a2 = alpha * alpha;
    float d = sqrtf(a2 + (1.0f - a2) * NdotV * NdotV) + NdotV;
    return 2.0f * NdotV / std::max(d, 1e-7f);
}
inline float G2_h(float NdotV, float NdotL, float alpha) {
    float a2 = alpha * alpha;
    float dA = NdotV * sqrtf(a2 + (1.0f - a2) * NdotL * NdotL);
    float dB = NdotL * sqrtf(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * NdotL * NdotV / (dA + dB);
}
inline void coord_h(f3 N, f3& T1, f3& T2) {
    if (fabsf(N.z) < 0.999f) T1 = rtx::normalize(rtx::cross(mk3(0, 0, 1), N));
    else T1 = rtx::normalize(rtx::cross(mk3(1, 0, 0), N));
    T2 = rtx::cross(N, T1);
}
// SampleGGX, ObjLoader.h:176-252 (the older, non-warped VNDF variant the host uses)
f3 SampleGGX_h(float rough, f3 outgoing, f3 normal, float e0, float e1) {
    float alpha = rough * rough;
    f3 N = rtx::normalize(normal), V = rtx::normalize(outgoing), T1, T2;
    coord_h(N, T1, T2);
    f3 Vh = rtx::normalize(mk3(rtx::dot(T1, V), rtx::dot(T2, V), rtx::dot(N, V)));
    f3 Vs = rtx::normalize(mk3(alpha * Vh.x, alpha * Vh.y, Vh.z));
    float lensq = Vs.x * Vs.x + Vs.y * Vs.y;
    f3 T1h, T2h;
    if (lensq > 0.0f) { float inv = 1.0f / sqrtf(lensq); T1h = rtx::normalize(mk3(-Vs.y * inv, Vs.x * inv, 0.0f)); T2h = rtx::cross(Vs, T1h); }
    else { T1h = mk3(1, 0, 0); T2h = mk3(0, 1, 0); }
    float r = sqrtf(e0), phi = 2.0f * 3.141592654f * e1;
    float x = r * cosf(phi), y = r * sinf(phi);
    float z = sqrtf(std::max(0.0f, 1.0f - x * x - y * y));
    f3 Nhs = rtx::normalize(mk3(x * T1h.x + y * T2h.x + z * Vs.x, x * T1h.y + y * T2h.y + z * Vs.y, x * T1h.z + y * T2h.z + z * Vs.z));
    f3 Nh = rtx::normalize(mk3(alpha * Nhs.x, alpha * Nhs.y, std::max(0.0f, Nhs.z)));
    f3 H = rtx::normalize(mk3(Nh.x * T1.x + Nh.y * T2.x + Nh.z * N.x, Nh.x * T1.y + Nh.y * T2.y + Nh.z * N.y, Nh.x * T1.z + Nh.y * T2.z + Nh.z * N.z));
    f3 I = -V; float k = 2.0f * rtx::dot(H, I);
    return rtx::normalize(mk3(I.x - k * H.x, I.y - k * H.y, I.z - k * H.z));
}
// deterministic stand-in for std::mt19937(std::random_device) + uniform_real_distribution (ObjLoader.h:298-300)
struct Lcg { uint64_t s; float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f); } };
}  // namespace

float ComputeEss(const XMFLOAT3& Nn, const XMFLOAT3& Vv, float roughness, XMFLOAT3 /*Ks*/, int numSamples, Material& /*mat*/, uint32_t seed) {
    float Ess = 0.0f;
    Lcg rng{(uint64_t)seed * 0x9E3779B97F4A7C15ull + 1};
    f3 N = mk3(Nn.x, Nn.y, Nn.z), V = mk3(Vv.x, Vv.y, Vv.z);
    for (int i = 0; i < numSamples; ++i) {
        float u1 = rng.next(), u2 = rng.next();
        f3 L = SampleGGX_h(roughness, V, N, u1, u2);
        if (rtx::dot(N, L) <= 0.0f) continue;                                   // :311
        f3 Nu = rtx::normalize(N), Lu = rtx::normalize(L), Vu = rtx::normalize(V);
        float NdotL = fabsf(rtx::dot(Nu, Lu));
        // EvaluateBRDF_GGX with F = 1 and D cancelled (ObjLoader.h:256-268)
        float NdotV = std::max(rtx::dot(Nu, Vu), 0.0f), NdotLc = std::max(rtx::dot(Nu, Lu), 0.0f);
        float G = G2_h(NdotV, NdotLc, roughness * roughness);
        float brdf = G / std::max(4.0f * NdotV * NdotLc, 1e-7f);
        // BRDF_PDF_GGX (ObjLoader.h:271-289): G1 / (4 NdotV)
        float NdotV2 = std::max(rtx::dot(Nu, Vu), 0.0f);
        float pdf = std::max(G1_h(NdotV2, roughness * roughness) / std::max(NdotV2 * 4.0f, 1e-7f), 1e-7f);
        float lum = (brdf + brdf + brdf) / 3.0f;
        if (lum > 0.0f) Ess += (NdotL * lum) / pdf;                             // :324-326
    }
    (void)D_GGX_h;
    return numSamples > 0 ? Ess / numSamples : 0.0f;
}

void GenerateEssLUT(Material& mat) {
    constexpr float EPS = 0.04f;                                                // ObjLoader.h:352
    for (int t = 0; t < LUT_SIZE_THETA; ++t) {
        float cosT = EPS + (float)t / (LUT_SIZE_THETA - 1) * (1.0f - EPS);      // :360
        float sinT = sqrtf(std::max(EPS, 1.0f - cosT * cosT));                  // :363
        XMFLOAT3 N = {0.0f, 0.0f, 1.0f}, V = {sinT, 0.0f, cosT};
        mat.LUT[t] = ComputeEss(N, V, mat.Pr_Pm_Ps_Pc.x, XMFLOAT3(1.0f, 1.0f, 1.0f), NUM_SAMPLES_MC, mat, 0x9E3779B9u ^ (uint32_t)t);
    }
}

// ------------------------------------------------------------------------------------------------
// minimal OBJ / MTL reader
// ------------------------------------------------------------------------------------------------
namespace {
inline const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') p++; return p; }
inline bool parse_float(const char*& p, float& out) { p = skip_ws(p); char* e; double d = strtod(p, &e); if (e == p) return false; out = (float)d; p = e; return true; }
// tinyobj::material_t's defaults (InitMaterial, tiny_obj_loader.h): everything zero except dissolve = 1, ior = 1, shininess = 1
struct MtlRec { std::string name; float Kd[3] = {0, 0, 0}, Ks[3] = {0, 0, 0}, Ke[3] = {0, 0, 0}; float d = 1.0f, Pr = 0, Pm = 0, Ps = 0, Pc = 0, Ni = 1.0f; bool has_d = false;
                float Ka[3] = {0, 0, 0}, Tf[3] = {0, 0, 0}; float Ns = 1.0f, Pcr = 0, aniso = 0, anisor = 0; int illum = 0;
                std::string tex[kNumMapSlots]; };
// `map_* [options] file name until the end of the line` (ParseTextureNameAndOption, tiny_obj_loader.h): options first, each with its own
// argument count, then the rest of the line is the name (it may contain blanks)
std::string parse_texname(const char* p) {
    auto word = [&](const char* k) { size_t n = strlen(k); if (!strncmp(p, k, n) && (p[n] == ' ' || p[n] == '\t')) { p += n + 1; return true; } return false; };
    auto skip_tok = [&]() { p = skip_ws(p); while (*p && *p != ' ' && *p != '\t') p++; };
    // tinyobj's parseReal advances past one blank-separated word whether or not it is a number, and parseReal3 always takes three: `-o 0.5 a b.png`
    // eats `a` and `b.png` as the missing y and z (tiny_obj_loader.h: parseReal / parseReal3) — restated, since the reference's loader sees exactly that
    auto skip_reals = [&](int n) { for (int i = 0; i < n; i++) skip_tok(); };
    std::string name;
    while (true) {
        p = skip_ws(p);
        if (!*p) break;
        if (word("-blendu") || word("-blendv") || word("-clamp") || word("-imfchan") || word("-colorspace") || word("-type") || word("-texres")) skip_tok();
        else if (word("-boost") || word("-bm")) skip_reals(1);
        else if (word("-mm")) skip_reals(2);
        else if (word("-o") || word("-s") || word("-t")) skip_reals(3);
        else { name = p; break; }
    }
    while (!name.empty() && (name.back() == ' ' || name.back() == '\t')) name.pop_back();
    return name;
}
struct ObjIdx { int v, vn; };
struct ObjParsed { std::vector<float> v, vn; std::vector<ObjIdx> idx; std::vector<int> face_mat; std::vector<MtlRec> mats; };


void load_mtl(const std::string& path, std::vector<MtlRec>& mats, std::unordered_map<std::string, int>& map) {
    std::ifstream f(path);
    if (!f) return;                         // tinyobj only warns when the .mtl is missing
    std::string line; MtlRec cur; bool have = false;
    bool has_kd = false;                    // tiny_obj_loader.h:2083-2085,2173: set by any `Kd` line and (sic) never cleared at `newmtl`
    auto flush = [&]() { if (have) { map[cur.name] = (int)mats.size(); mats.push_back(cur); } };
    auto set_tex = [&](int slot, const char* q) { std::string n = parse_texname(q); if (!n.empty()) cur.tex[slot] = n; };   // a statement without a name (its options ate it) leaves the slot as it was, as tinyobj does
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* p = skip_ws(line.c_str());
        if (!*p || *p == '#') continue;
        auto key = [&](const char* k) { size_t n = strlen(k); if (!strncmp(p, k, n) && (p[n] == ' ' || p[n] == '\t')) { p += n; return true; } return false; };
        if (key("newmtl")) { flush(); cur = MtlRec(); have = true; cur.name = skip_ws(p); while (!cur.name.empty() && (cur.name.back() == ' ' || cur.name.back() == '\t')) cur.name.pop_back(); }
        else if (key("Kd")) { for (int i = 0; i < 3; i++) parse_float(p, cur.Kd[i]); has_kd = true; }
        else if (key("map_Kd")) { if (!has_kd) cur.Kd[0] = cur.Kd[1] = cur.Kd[2] = 0.6f; set_tex(MAP_KD, p); }   // a diffuse texture without a Kd before it: tiny_obj_loader.h:2328-2341
        else if (key("Ks")) { for (int i = 0; i < 3; i++) parse_float(p, cur.Ks[i]); }
        else if (key("Ke")) { for (int i = 0; i < 3; i++) parse_float(p, cur.Ke[i]); }
        else if (key("d")) { parse_float(p, cur.d); cur.has_d = true; }
        else if (key("Tr")) { float t; if (parse_float(p, t) && !cur.has_d) cur.d = 1.0f - t; }
        else if (key("Ni")) parse_float(p, cur.Ni);
        else if (key("Pr")) parse_float(p, cur.Pr);
        else if (key("Pm")) parse_float(p, cur.Pm);
        else if (key("Ps")) parse_float(p, cur.Ps);
        else if (key("Pc")) parse_float(p, cur.Pc);
        // the rest of tinyobj's material_t: parsed and carried beside the 128-byte record (Vertex.h:21 "ADD MAP IDs LATER"), not consumed by any shader
        else if (key("Pcr")) parse_float(p, cur.Pcr);
        else if (key("aniso")) parse_float(p, cur.aniso);
        else if (key("anisor")) parse_float(p, cur.anisor);
        else if (key("Ns")) parse_float(p, cur.Ns);
        else if (key("Ka")) { for (int i = 0; i < 3; i++) parse_float(p, cur.Ka[i]); }
        else if (key("Kt") || key("Tf")) { for (int i = 0; i < 3; i++) parse_float(p, cur.Tf[i]); }
        else if (key("illum")) { float v = 0; if (parse_float(p, v)) cur.illum = (int)v; }
        else if (key("map_Ka")) set_tex(MAP_KA, p);
        else if (key("map_Ks")) set_tex(MAP_KS, p);
        else if (key("map_Ke")) set_tex(MAP_KE, p);
        else if (key("map_Ns")) set_tex(MAP_NS, p);
        else if (key("map_bump") || key("map_Bump") || key("bump")) set_tex(MAP_BUMP, p);
        else if (key("map_d")) set_tex(MAP_D, p);
        else if (key("disp")) set_tex(MAP_DISP, p);
        else if (key("refl")) set_tex(MAP_REFL, p);
        else if (key("map_Pr")) set_tex(MAP_PR, p);
        else if (key("map_Pm")) set_tex(MAP_PM, p);
        else if (key("map_Ps")) set_tex(MAP_PS, p);
        else if (key("norm")) set_tex(MAP_NORM, p);
    }
    flush();
}

// Polygons with more than four vertices: the ear clipping of tinyobjloader v2.0.0's default build (tiny_obj_loader.h:1741-1963, no
// mapbox earcut), restated with its exact decisions, because the ORDER and the NUMBER of the triangles it emits are what the reference's
// loader sees (pinned by tests/golden/objfuzz):
//  * the polygon is projected on two coordinate axes picked from the FIRST corner (three consecutive vertices) whose cross product has a
//    component above FLT_EPSILON: (y,z) if |c.x| is the strict maximum, (x,y) if |c.z| is, else (x,z); (y,z) if there is no such corner;
//  * candidate ear = three consecutive vertices from `guess`; it is skipped when cross(e0, e1) * (v0.x v1.y - v0.y v1.x) / 2 < 0 (sic: the
//    second factor is not the polygon's area) or when another remaining vertex lies inside it (the pnpoly crossing test);
//  * a clipped ear removes its middle vertex; the search gives up after as many fruitless tries as there are vertices left, and whatever
//    remains is emitted only if it is exactly one triangle.  All arithmetic is binary32, in the reference's operation order.
template <class Emit>
void ear_clip(const float* V, const std::vector<ObjIdx>& face, Emit&& tri) {
    const size_t n0 = face.size();
    int ax0 = 1, ax1 = 2;
    for (size_t k = 0; k < n0; k++) {
        const float* a = V + (size_t)face[k % n0].v * 3; const float* b = V + (size_t)face[(k + 1) % n0].v * 3; const float* c = V + (size_t)face[(k + 2) % n0].v * 3;
        const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2], e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
        const float cx = fabsf(e0y * e1z - e0z * e1y), cy = fabsf(e0z * e1x - e0x * e1z), cz = fabsf(e0x * e1y - e0y * e1x);
        const float eps = 1.1920928955078125e-7f;                    // std::numeric_limits<float>::epsilon()
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) { ax0 = 0; if (cz > cx && cz > cy) ax1 = 1; }
            break;
        }
    }
    std::vector<int> rem(n0);                                        // positions in `face`
    for (size_t k = 0; k < n0; k++) rem[k] = (int)k;
    size_t guess = 0, budget = n0, prev = n0;
    while (rem.size() > 3 && budget > 0) {
        const size_t m = rem.size();
        if (guess >= m) guess -= m;
        if (prev != m) { prev = m; budget = m; } else budget--;
        int ind[3]; float vx[3], vy[3];
        for (int k = 0; k < 3; k++) { ind[k] = rem[(guess + k) % m]; const float* q = V + (size_t)face[ind[k]].v * 3; vx[k] = q[ax0]; vy[k] = q[ax1]; }
        const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        const float cross = e0x * e1y - e0y * e1x;
        const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
        if (cross * area < 0.0f) { guess++; continue; }
        bool overlap = false;
        for (size_t other = 3; other < m && !overlap; other++) {
            const float* q = V + (size_t)face[rem[(guess + other) % m]].v * 3;
            const float tx = q[ax0], ty = q[ax1];
            int cflag = 0;
            for (int i = 0, j = 2; i < 3; j = i++)
                if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) cflag = !cflag;
            overlap = cflag != 0;
        }
        if (overlap) { guess++; continue; }
        tri(ind[0], ind[1], ind[2]);
        rem.erase(rem.begin() + (long)((guess + 1) % m));
    }
    if (rem.size() == 3) tri(rem[0], rem[1], rem[2]);
}

bool parse_obj(const std::string& file, const std::string& mtl_dir, ObjParsed& out, std::string& err) {
    std::ifstream f(file);
    if (!f) { err = "cannot open " + file; return false; }
    std::unordered_map<std::string, int> mtl_map;
    int cur_mat = -1;
    std::string line;
    std::vector<ObjIdx> face;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* p = skip_ws(line.c_str());
        if (!*p || *p == '#') continue;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) { p += 2; float a[3] = {0, 0, 0}; for (int i = 0; i < 3; i++) parse_float(p, a[i]); out.v.insert(out.v.end(), a, a + 3); }
        else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) { p += 3; float a[3] = {0, 0, 0}; for (int i = 0; i < 3; i++) parse_float(p, a[i]); out.vn.insert(out.vn.end(), a, a + 3); }
        else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2; face.clear();
            const int nv = (int)(out.v.size() / 3), nn = (int)(out.vn.size() / 3);
            while (true) {
                p = skip_ws(p);
                if (!*p) break;
                char* e; long vi = strtol(p, &e, 10);
                if (e == p) break;
                p = e; long ni = 0; bool has_n = false;
                if (*p == '/') { p++; if (*p != '/') { (void)strtol(p, &e, 10); p = e; } if (*p == '/') { p++; ni = strtol(p, &e, 10); has_n = e != p; p = e; } }
                ObjIdx ix; ix.v = vi > 0 ? (int)vi - 1 : nv + (int)vi; ix.vn = has_n ? (ni > 0 ? (int)ni - 1 : nn + (int)ni) : -1;
                face.push_back(ix);
            }
            const int n = (int)face.size();
            if (n < 3) continue;
            for (auto& ix : face) if (ix.v < 0 || ix.v >= nv) { err = "face with invalid vertex index"; return false; }
            auto tri = [&](int a, int b, int c) { out.idx.push_back(face[a]); out.idx.push_back(face[b]); out.idx.push_back(face[c]); out.face_mat.push_back(cur_mat); };
            if (n == 3) tri(0, 1, 2);
            else if (n == 4) {                                   // tiny_obj_loader.h:1563-1608
                const float* V = out.v.data();
                auto d2 = [&](int a, int b) { float s = 0; for (int k = 0; k < 3; k++) { float e = V[face[b].v * 3 + k] - V[face[a].v * 3 + k]; s += e * e; } return s; };
                if (d2(0, 2) < d2(1, 3)) { tri(0, 1, 2); tri(0, 2, 3); } else { tri(0, 1, 3); tri(1, 2, 3); }
            } else ear_clip(out.v.data(), face, tri);            // tiny_obj_loader.h:1741-1963
        }
        else if (!strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
            std::string name = skip_ws(p + 6);
            while (!name.empty() && (name.back() == ' ' || name.back() == '\t')) name.pop_back();
            auto it = mtl_map.find(name);
            cur_mat = it != mtl_map.end() ? it->second : -1;
        }
        else if (!strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
            std::istringstream ss(skip_ws(p + 6)); std::string name;
            while (ss >> name) {
                std::string dir = mtl_dir;
                if (!dir.empty() && dir.back() != '/') dir += '/';
                size_t before = out.mats.size();
                load_mtl(dir + name, out.mats, mtl_map);
                if (out.mats.size() > before) break;             // tinyobj stops at the first library that loads
            }
        }
    }
    return true;
}
}  // namespace

void ObjLoader::loadObjFile(const std::string& inputfile, std::vector<Vertex>* vertices, std::vector<UINT>* indices,
                            std::vector<Material>* mats, std::vector<UINT>* materialIDs, UINT* materialOffset,
                            UINT* materialVertexOffset, const std::string& material_search_path) {
    loadObjFileEx(inputfile, vertices, indices, mats, materialIDs, materialOffset, materialVertexOffset, nullptr, nullptr, material_search_path);
}

void ObjLoader::loadObjFileEx(const std::string& inputfile, std::vector<Vertex>* vertices, std::vector<UINT>* indices,
                              std::vector<Material>* mats, std::vector<UINT>* materialIDs, UINT* materialOffset,
                              UINT* materialVertexOffset, std::vector<MaterialExt>* ext, std::vector<std::string>* textures,
                              const std::string& material_search_path) {
    ObjParsed P; std::string err;
    if (!parse_obj(inputfile, material_search_path, P, err)) throw std::runtime_error("ObjLoader: " + err);   // ObjLoader.h:399-404
    // default material for faces without one (ObjLoader.h:415-417)
    Material defaultMaterial(XMFLOAT4(1.0f, 1.0f, 1.0f, 1.0f), XMFLOAT4(1.0f, 0.0f, 0.0f, 0.0f));
    mats->push_back(defaultMaterial);
    if (ext) ext->push_back(MaterialExt());
    (*materialOffset)++;
    for (const MtlRec& m : P.mats) {                                                                        // :420-444
        Material t(XMFLOAT4(m.Kd[0], m.Kd[1], m.Kd[2], m.d), XMFLOAT4(m.Pr, m.Pm, m.Ps, m.Pc));
        t.Ke = XMFLOAT3(m.Ke); t.Ks = XMFLOAT3(m.Ks);
        GenerateEssLUT(t);
        mats->push_back(t);
        if (ext) {                         // everything else tinyobj parsed: beside the record, map ids index *textures (one entry per distinct file name)
            MaterialExt x;
            x.Ni = m.Ni; x.Ns = m.Ns; x.Pcr = m.Pcr; x.aniso = m.aniso; x.anisor = m.anisor; x.illum = m.illum;
            for (int k = 0; k < 3; k++) { x.Ka[k] = m.Ka[k]; x.Tf[k] = m.Tf[k]; }
            for (int k = 0; k < kNumMapSlots; k++) {
                x.map[k] = -1;
                if (m.tex[k].empty() || !textures) continue;
                auto it = std::find(textures->begin(), textures->end(), m.tex[k]);
                if (it == textures->end()) { textures->push_back(m.tex[k]); it = textures->end() - 1; }
                x.map[k] = (int)(it - textures->begin());
            }
            ext->push_back(x);
        }
    }
    std::unordered_map<Vertex, uint32_t> unique;
    const size_t nfaces = P.idx.size() / 3;
    for (size_t f = 0; f < nfaces; f++) {                                                                   // :449-491
        for (int v = 0; v < 3; v++) materialIDs->push_back((UINT)(P.face_mat[f] + (int)*materialOffset));
        for (int v = 0; v < 3; v++) {
            const ObjIdx ix = P.idx[f * 3 + v];
            XMFLOAT3 pos(P.v[ix.v * 3], P.v[ix.v * 3 + 1], P.v[ix.v * 3 + 2]);
            XMFLOAT4 normal(0.0f, 0.0f, 0.0f, (float)*materialVertexOffset);
            if (ix.vn >= 0 && (size_t)ix.vn * 3 + 2 < P.vn.size()) normal = XMFLOAT4(P.vn[ix.vn * 3], P.vn[ix.vn * 3 + 1], P.vn[ix.vn * 3 + 2], (float)*materialVertexOffset);
            Vertex vert(pos, normal);
            auto it = unique.find(vert);
            if (it == unique.end()) { it = unique.emplace(vert, (uint32_t)vertices->size()).first; vertices->push_back(vert); }
            indices->push_back(it->second);
        }
    }
    *materialOffset += (UINT)P.mats.size();                                                                 // :494
}
