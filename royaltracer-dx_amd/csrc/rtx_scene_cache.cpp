// rtx_scene_cache.cpp — binary scene cache ("a binary scene cache (BVH + LUTs) so Bistro-class loads in seconds").
//
// One file holds what rtx_set_materials / rtx_add_mesh / rtx_add_instance handed over (the reference's m_materials, m_materialIDs,
// per-model VB / IB, m_instances: Renderer.cpp:363-407, 908-921) AND what rtx_commit_scene derived from it for the device: the
// MaterialOptimized table with its Ess LUTs, the compressed 8-wide BVH, the world-space triangles in its leaf order, the per-triangle
// shading records, instance matrices, the emissive-triangle list / CDF, the tiny-scene pre-test records.  Loading it replaces the
// binned-SAH build + collapse (2.2 s for 3.8 M triangles) by a read, a checksum and the upload.
//
// Layout (little endian, everything 16-byte aligned):
//   Header  { magic "RTXSCN01", u32 version, u32 endian = 0x01020304, u32 layout[8] = sizeof of the record types, u64 payload bytes,
//             u64 checksum of the payload, u64 section count }
//   Section { u32 tag, u32 element size, u64 element count } + data, padded to 16 bytes — in a fixed order, see write_all().
// The checksum is a 4-lane multiply-xorshift hash over 64-bit words, computed per 4 MiB chunk (chunks in parallel) and folded in order.
// A file with another version, another record layout, a wrong length or a wrong checksum is refused; nothing is partially loaded.
//
// Not stored (re-derived on demand): the binary build tree and its leaf order (only a HOST refit needs them: after a load the first
// topology-preserving host commit rebuilds instead), the leaf-order triangle copy and the object-space triangles of the GPU refit
// (re-derived from the meshes when the first transform-only commit asks for them).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "rtx_scene_host.hpp"

namespace rtx {
namespace {

constexpr char kMagic[8] = {'R', 'T', 'X', 'S', 'C', 'N', '0', '1'};
constexpr uint32_t kVersion = 3;              // 2: + the host layer's MaterialExt records and texture names (two sections at the end); 3: TriGPU::e1.w carries the determinant floor of the hit definition
struct Header { char magic[8]; uint32_t version, endian; uint32_t layout[8]; uint64_t payload, checksum, nsections; };
struct SecHead { uint32_t tag, elem; uint64_t count; };
static_assert(sizeof(Header) == 72 && sizeof(SecHead) == 16, "cache header layout");
constexpr size_t kHeaderBytes = 80;            // Header padded to 16

void fill_layout(uint32_t l[8]) {
    l[0] = sizeof(Node8GPU); l[1] = sizeof(TriGPU); l[2] = sizeof(TriShade); l[3] = sizeof(MatGPU);
    l[4] = sizeof(InstGPU); l[5] = sizeof(LightGPU); l[6] = sizeof(SmallRecPair); l[7] = sizeof(InstHost);
}

// ---- checksum: per-chunk 4-lane word hash, chunks folded in order (so it can be computed by several threads) ----
constexpr size_t kChunk = 4u << 20;
inline uint64_t mix(uint64_t h, uint64_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
uint64_t hash_chunk(const uint8_t* p, size_t n) {
    uint64_t h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) { uint64_t w[4]; memcpy(w, p + i, 32); for (int k = 0; k < 4; k++) h[k] = mix(h[k], w[k]); }
    uint64_t tail[4] = {0, 0, 0, 0};
    memcpy(tail, p + i, n - i);
    for (int k = 0; k < 4; k++) h[k] = mix(h[k], tail[k]);
    return mix(mix(mix(mix(n, h[0]), h[1]), h[2]), h[3]);
}
unsigned pool_size() { unsigned n = std::thread::hardware_concurrency(); return n ? std::min(n, 16u) : 4u; }
uint64_t hash_payload(const uint8_t* p, size_t n) {
    const size_t nchunks = (n + kChunk - 1) / kChunk;
    std::vector<uint64_t> part(nchunks);
    const unsigned T = (unsigned)std::min<size_t>(pool_size(), std::max<size_t>(nchunks, 1));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] { for (size_t c = t; c < nchunks; c += T) part[c] = hash_chunk(p + c * kChunk, std::min(kChunk, n - c * kChunk)); });
    for (auto& x : th) x.join();
    uint64_t h = 0x452821E638D01377ull ^ n;
    for (uint64_t v : part) h = mix(h, v);
    return h;
}
void parallel_copy(void* dst, const void* src, size_t n) {
    if (!n) return;                                        // (empty vector: data() may be null)
    if (n < (8u << 20)) { memcpy(dst, src, n); return; }
    const unsigned T = pool_size();
    const size_t per = ((n / T) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) {
        const size_t off = (size_t)t * per;
        if (off >= n) break;
        th.emplace_back([=] { memcpy((uint8_t*)dst + off, (const uint8_t*)src + off, std::min(per, n - off)); });
    }
    for (auto& x : th) x.join();
}

// ---- writer -------------------------------------------------------------------------------------------------------------------
struct Writer {
    std::vector<uint8_t> buf; uint64_t nsec = 0;
    void raw(uint32_t tag, uint32_t elem, uint64_t count, const void* data) {
        SecHead s{tag, elem, count};
        const size_t at = buf.size(), bytes = (size_t)elem * count, padded = (bytes + 15) & ~(size_t)15;
        buf.resize(at + sizeof(s) + padded, 0);
        memcpy(&buf[at], &s, sizeof(s));
        if (bytes) memcpy(&buf[at + sizeof(s)], data, bytes);
        nsec++;
    }
    template <class T> void vec(uint32_t tag, const std::vector<T>& v) { raw(tag, (uint32_t)sizeof(T), v.size(), v.data()); }
};
struct Scalars { uint32_t stack8, small_nrec, small_nocc, max_depth, nmesh, has_cam, pad1, pad2; float bvh_pad, small_cm, small_delta, small_hull_margin, total_weight, f0, f1, f2;
                 float cam[12]; };       // cam: eye, center, up, fovY (degrees), znear, zfar — the host layer's scene camera (Renderer.cpp:46-48, 1730-1731); has_cam = 0 for a context-level save

// ---- reader -------------------------------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t* p; size_t n, at = 0; std::string* err;
    bool head(uint32_t tag, uint32_t elem, SecHead& s) {
        if (at + sizeof(SecHead) > n) { *err = "scene cache: truncated section table"; return false; }
        memcpy(&s, p + at, sizeof(s)); at += sizeof(s);
        const size_t bytes = (size_t)s.elem * s.count, padded = (bytes + 15) & ~(size_t)15;
        if (s.tag != tag || s.elem != elem || s.count > n || at + padded > n) { *err = "scene cache: unexpected section (tag " + std::to_string(s.tag) + ")"; return false; }
        return true;
    }
    template <class T> bool vec(uint32_t tag, std::vector<T>& v) {
        SecHead s; if (!head(tag, (uint32_t)sizeof(T), s)) return false;
        v.resize(s.count);
        parallel_copy(v.data(), p + at, sizeof(T) * s.count);
        at += (sizeof(T) * s.count + 15) & ~(size_t)15;
        return true;
    }
};

enum : uint32_t { T_SCAL = 1, T_MATS128, T_MATIDS, T_INSTH, T_MESHV, T_MESHI, T_MESHB, T_BMATS, T_NODES8, T_SLOTS8, T_TRIS8, T_LEVELS, T_SMALLR, T_SMALLT, T_SMALLP,
                T_SHADE, T_BINST, T_LIGHTS, T_LIGHTS80, T_AUXREC, T_AUXTXT };

}  // namespace

bool save_scene_cache(const SceneHost& H, const BuiltScene& B, const char* path, std::string& err, const float* cam12, const CacheAux* aux) {
    if (!path || !*path) { err = "scene cache: empty path"; return false; }
    if (H.topo_dirty || H.mats_dirty || B.tris8.size() < B.shade.size() || B.tris8.size() != B.tri_slots8.size() || (!B.tris8.empty() && B.nodes8.empty())) { err = "scene cache: scene not built"; return false; }
    Writer w;
    Scalars sc{}; sc.stack8 = B.stack8; sc.small_nrec = B.small_nrec; sc.small_nocc = B.small_nocc; sc.max_depth = B.max_depth; sc.nmesh = (uint32_t)H.meshes.size();
    sc.bvh_pad = B.bvh_pad; sc.small_cm = B.small_cm; sc.small_delta = B.small_delta; sc.small_hull_margin = B.small_hull_margin; sc.total_weight = B.total_weight;
    if (cam12) { sc.has_cam = 1; memcpy(sc.cam, cam12, sizeof(sc.cam)); }
    w.raw(T_SCAL, sizeof(Scalars), 1, &sc);
    w.vec(T_MATS128, H.mats128); w.vec(T_MATIDS, H.matids); w.vec(T_INSTH, H.insts);
    for (const MeshHost& m : H.meshes) { w.vec(T_MESHV, m.verts); w.vec(T_MESHI, m.idx); w.raw(T_MESHB, 4, 1, &m.matid_base); }
    w.vec(T_BMATS, B.mats); w.vec(T_NODES8, B.nodes8); w.vec(T_SLOTS8, B.tri_slots8); w.vec(T_TRIS8, B.tris8); w.vec(T_LEVELS, B.level_start8);
    w.vec(T_SMALLR, B.small_recs); w.vec(T_SMALLT, B.small_tris); w.vec(T_SMALLP, B.small_poly);
    w.vec(T_SHADE, B.shade); w.vec(T_BINST, B.insts); w.vec(T_LIGHTS, B.lights); w.vec(T_LIGHTS80, B.lights80);
    const CacheAux none;                                                       // what rides beside the material table in the host layer (MaterialExt, texture names); empty for a context-level save
    const CacheAux& ax = aux ? *aux : none;
    w.raw(T_AUXREC, ax.rec_bytes ? ax.rec_bytes : 1u, ax.rec_bytes ? ax.records.size() / ax.rec_bytes : 0u, ax.records.data()); w.vec(T_AUXTXT, ax.text);
    Header h{}; memcpy(h.magic, kMagic, 8); h.version = kVersion; h.endian = 0x01020304u; fill_layout(h.layout);
    h.payload = w.buf.size(); h.checksum = hash_payload(w.buf.data(), w.buf.size()); h.nsections = w.nsec;
    uint8_t hb[kHeaderBytes] = {0}; memcpy(hb, &h, sizeof(h));
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) { err = "scene cache: cannot create " + tmp; return false; }
    const bool ok = fwrite(hb, 1, kHeaderBytes, f) == kHeaderBytes && fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
    if (fclose(f) != 0 || !ok) { remove(tmp.c_str()); err = "scene cache: write failed"; return false; }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); err = "scene cache: cannot move into place"; return false; }   // readers never see a half-written file
    return true;
}

static bool load_scene_cache_impl(const char* path, SceneHost& H, BuiltScene& B, std::string& err, float* cam12, CacheAux* aux) {
    const int fd = path ? open(path, O_RDONLY) : -1;
    if (fd < 0) { err = std::string("scene cache: cannot open ") + (path ? path : "(null)"); return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < kHeaderBytes) { close(fd); err = "scene cache: file too short"; return false; }
    const size_t total = (size_t)st.st_size;
    void* map = mmap(nullptr, total, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { err = "scene cache: mmap failed"; return false; }
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } guard{map, total};
    const uint8_t* base = (const uint8_t*)map;
    Header h; memcpy(&h, base, sizeof(h));
    uint32_t lay[8]; fill_layout(lay);
    if (memcmp(h.magic, kMagic, 8) != 0) { err = "scene cache: not a scene cache file"; return false; }
    if (h.version != kVersion) { err = "scene cache: version " + std::to_string(h.version) + ", this library reads version " + std::to_string(kVersion); return false; }
    if (h.endian != 0x01020304u || memcmp(h.layout, lay, sizeof(lay)) != 0) { err = "scene cache: written with another record layout"; return false; }
    if (h.payload != total - kHeaderBytes) { err = "scene cache: length does not match the header"; return false; }
    // (the checksum is a word hash: it catches accidental damage, not a crafted file — hence every index below is checked against its array)
    if (hash_payload(base + kHeaderBytes, h.payload) != h.checksum) { err = "scene cache: checksum mismatch (corrupt file)"; return false; }
    // everything is read into temporaries first: a malformed section table must not leave a half-loaded scene behind
    SceneHost Hn; BuiltScene Bn;
    Reader r{base + kHeaderBytes, (size_t)h.payload, 0, &err};
    SecHead s; Scalars sc;
    if (!r.head(T_SCAL, sizeof(Scalars), s) || s.count != 1) { if (err.empty()) err = "scene cache: bad scalar block"; return false; }
    memcpy(&sc, r.p + r.at, sizeof(sc)); r.at += sizeof(sc);
    if (!r.vec(T_MATS128, Hn.mats128) || !r.vec(T_MATIDS, Hn.matids) || !r.vec(T_INSTH, Hn.insts)) return false;
    // a mesh is three sections (vertices, indices, material-id base): at least 3 section heads + 16 bytes of what is left, and never more than the header's count
    if ((uint64_t)sc.nmesh * (3 * sizeof(SecHead) + 16) > h.payload - r.at || (uint64_t)sc.nmesh * 3 > h.nsections) { err = "scene cache: mesh count does not fit the file"; return false; }
    Hn.meshes.resize(sc.nmesh);
    for (MeshHost& m : Hn.meshes) {
        if (!r.vec(T_MESHV, m.verts) || !r.vec(T_MESHI, m.idx)) return false;
        if (!r.head(T_MESHB, 4, s) || s.count != 1) return false;
        memcpy(&m.matid_base, r.p + r.at, 4); r.at += 16;
    }
    if (!r.vec(T_BMATS, Bn.mats) || !r.vec(T_NODES8, Bn.nodes8) || !r.vec(T_SLOTS8, Bn.tri_slots8) || !r.vec(T_TRIS8, Bn.tris8) || !r.vec(T_LEVELS, Bn.level_start8) ||
        !r.vec(T_SMALLR, Bn.small_recs) || !r.vec(T_SMALLT, Bn.small_tris) || !r.vec(T_SMALLP, Bn.small_poly) || !r.vec(T_SHADE, Bn.shade) || !r.vec(T_BINST, Bn.insts) ||
        !r.vec(T_LIGHTS, Bn.lights) || !r.vec(T_LIGHTS80, Bn.lights80)) return false;
    CacheAux ax;
    {   // opaque to this layer: fixed-size records + a text blob (the host layer checks their shape against its own structs)
        if (r.at + sizeof(SecHead) > r.n) { err = "scene cache: truncated section table"; return false; }
        SecHead a; memcpy(&a, r.p + r.at, sizeof(a));
        if (a.tag != T_AUXREC || !a.elem || a.elem > 4096u || !r.head(T_AUXREC, a.elem, a)) { if (err.empty()) err = "scene cache: bad auxiliary section"; return false; }
        ax.rec_bytes = a.elem; ax.records.assign(r.p + r.at, r.p + r.at + (size_t)a.elem * a.count);
        r.at += ((size_t)a.elem * a.count + 15) & ~(size_t)15;
        if (!r.vec(T_AUXTXT, ax.text)) return false;
    }
    if (r.at != h.payload) { err = "scene cache: trailing bytes"; return false; }
    // consistency of what was read (indices stay inside their arrays: the kernels and the refit trust these)
    const size_t nt = Bn.shade.size(), nmat = Hn.mats128.size() / 32;
    const size_t nrec_pad = ((size_t)sc.small_nrec + 1) & ~(size_t)1;      // records are stored in pairs (rtx_scene_host.cpp)
    const size_t nrefs = Bn.tris8.size();                                      // leaf entries: the triangle count, plus the references spatial splits added
    bool ok = nrefs >= nt && Bn.tri_slots8.size() == nrefs && Bn.mats.size() == nmat && Bn.insts.size() == Hn.insts.size() && Bn.lights80.size() == Bn.lights.size() * 20 &&
              Hn.mats128.size() % 32 == 0 && (Bn.nodes8.empty() || Bn.level_start8.size() >= 2) && sc.small_nocc <= sc.small_nrec && sc.small_nrec <= kSmallSceneMaxTris &&
              Bn.small_recs.size() * 2 >= nrec_pad && (sc.small_nrec == 0 || (Bn.small_tris.size() >= nrec_pad * 2 && Bn.small_poly.size() >= nrec_pad * 4)) &&
              nrefs < (1u << 31);
    for (uint32_t id : Hn.matids) ok = ok && id < nmat;                       // (what rtx_commit_scene checks before a build)
    for (const MeshHost& m : Hn.meshes) {
        ok = ok && m.verts.size() % 7 == 0 && m.idx.size() % 3 == 0 && (size_t)m.matid_base + m.idx.size() <= Hn.matids.size();
        const size_t nv = m.verts.size() / 7;
        for (size_t i = 0; ok && i < m.idx.size(); i++) ok = m.idx[i] < nv;   // fill_objtris / a later host build gather vertices through these
    }
    size_t tri_at = 0;                                                        // instances own consecutive global triangle ranges (SceneHost::build)
    for (const InstHost& in : Hn.insts) {
        ok = ok && in.mesh < Hn.meshes.size() && in.tri_base == tri_at;
        if (ok) tri_at += Hn.meshes[in.mesh].idx.size() / 3;
    }
    ok = ok && tri_at == nt;
    for (size_t i = 0; ok && i < nt; i++) ok = Bn.shade[i].inst < Bn.insts.size();
    for (size_t i = 0; ok && i < nrefs; i++) ok = f2u(Bn.tris8[i].v0.w) < nt && Bn.tri_slots8[i] < nrefs;
    for (size_t i = 0; ok && i < Bn.small_tris.size(); i++) { const uint32_t g = f2u(Bn.small_tris[i].v0.w); ok = g < nt || g == kMissPrim; }
    for (size_t i = 0; ok && i < Bn.lights.size(); i++) ok = f2u(Bn.lights80[i * 20 + 7]) < Bn.insts.size();     // LightTriangle::instanceID (Renderer.h:113-124)
    for (size_t i = 0; ok && i < Bn.nodes8.size(); i++) {
        const Node8GPU& N = Bn.nodes8[i];
        const uint32_t ninternal = (uint32_t)__builtin_popcount(N.e_imask >> 24);
        uint32_t ntri = 0; for (int sl = 0; sl < 8; sl++) ntri += (uint32_t)__builtin_popcount((N.trivalid >> (4 * sl)) & 0xfu);
        ok = (ninternal == 0 || ((size_t)N.child_base + ninternal <= Bn.nodes8.size() && N.child_base > i)) && (size_t)N.tri_base + ntri <= nrefs;
    }
    if (!ok) { err = "scene cache: inconsistent contents"; return false; }
    // The traversal stack depth and the breadth-first levels are RE-DERIVED from the nodes (collapse_bvh8's two sweeps; children have larger indices, checked
    // above), never taken from the file: the per-lane LDS stack column has exactly stack8 entries, and the GPU refit sweeps the levels bottom-up.
    {
        const size_t nn = Bn.nodes8.size();
        std::vector<uint32_t> need(nn, 0), level(nn, 0);
        for (size_t i = nn; i-- > 0;) {
            const uint32_t nint = (uint32_t)__builtin_popcount(Bn.nodes8[i].e_imask >> 24);
            uint32_t deep = 0;
            for (uint32_t k = 0; k < nint; k++) deep = std::max(deep, need[(size_t)Bn.nodes8[i].child_base + k]);
            need[i] = (nint > 1 ? 1u : 0u) + deep;
        }
        const uint32_t stack8 = nn ? need[0] : 0u;
        std::vector<uint32_t> levels;
        for (size_t i = 0; i < nn; i++) {
            const uint32_t nint = (uint32_t)__builtin_popcount(Bn.nodes8[i].e_imask >> 24);
            for (uint32_t k = 0; k < nint; k++) level[(size_t)Bn.nodes8[i].child_base + k] = level[i] + 1;
        }
        for (size_t i = 0; i < nn; i++) {
            if (i && level[i] < level[i - 1]) { err = "scene cache: nodes are not in breadth-first order"; return false; }
            if (i == 0 || level[i] != level[i - 1]) levels.push_back((uint32_t)i);
        }
        if (nn) levels.push_back((uint32_t)nn);
        if (stack8 != sc.stack8 || (nn && levels != Bn.level_start8)) { err = "scene cache: stack depth / level table do not match the nodes"; return false; }
    }
    Bn.stack8 = sc.stack8; Bn.small_nrec = sc.small_nrec; Bn.small_nocc = sc.small_nocc; Bn.max_depth = sc.max_depth; Bn.bvh_pad = sc.bvh_pad;
    Bn.small_cm = sc.small_cm; Bn.small_delta = sc.small_delta; Bn.small_hull_margin = sc.small_hull_margin; Bn.total_weight = sc.total_weight; Bn.refit_count = 0;
    Bn.any_order = probe_anyhit_order(Bn);                                   // derived, like the stack depth: not taken from the file
    Hn.topo_dirty = false; Hn.mats_dirty = false;
    if (cam12) { if (!sc.has_cam) { err = "scene cache: the file holds no camera (written by rtx_save_scene_cache, not rtxh_scene_save)"; return false; } memcpy(cam12, sc.cam, sizeof(sc.cam)); }
    H = std::move(Hn); B = std::move(Bn);
    if (aux) *aux = std::move(ax);
    return true;
}

// nothing may throw across the extern "C" entry points that call this (rtx_load_scene_cache, rtxh_scene_load): a section count that passes the bounds but
// exhausts memory ends as an error string, not as std::terminate
bool load_scene_cache(const char* path, SceneHost& H, BuiltScene& B, std::string& err, float* cam12, CacheAux* aux) {
    try { return load_scene_cache_impl(path, H, B, err, cam12, aux); }
    catch (const std::bad_alloc&) { err = "scene cache: out of memory while reading"; }
    catch (const std::exception& e) { err = std::string("scene cache: ") + e.what(); }
    return false;
}

// object-space triangles of the GPU refit, re-derived from the meshes (what build() fills; a loaded cache does not carry them)
void SceneHost::fill_objtris(BuiltScene& B) const {
    size_t nt = 0; for (const InstHost& in : insts) nt += meshes[in.mesh].idx.size() / 3;
    B.objtris.resize(nt * 3);
    for (const InstHost& in : insts) {
        const MeshHost& m = meshes[in.mesh];
        for (uint32_t t = 0; t < m.idx.size() / 3; t++)
            for (int k = 0; k < 3; k++) { const float* p = &m.verts[(size_t)m.idx[t * 3 + k] * 7]; B.objtris[((size_t)in.tri_base + t) * 3 + k] = {p[0], p[1], p[2], 0.0f}; }
    }
}

}  // namespace rtx
