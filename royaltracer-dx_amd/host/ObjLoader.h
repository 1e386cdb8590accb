// ObjLoader.h — the reference's scene-loader surface (Pathtracer/src/Util/ObjLoader.h:393-495) on top of our
// own OBJ/MTL reader (the reference uses the vendored tinyobjloader v2.0.0; we re-implement the subset it
// relies on: v / vn / f with v, v/vt, v//vn, v/vt/vn and negative indices, o / g, usemtl, mtllib;
// MTL newmtl Kd Ks Ke d Tr Ni Pr Pm Ps Pc (into Material) and Ka Tf Ns illum Pcr aniso anisor + every map_* / bump / disp / refl / norm
// statement with its options (into MaterialExt, beside the record); quads split along the shorter diagonal as tinyobj does
// (tiny_obj_loader.h:1511-1608), larger polygons as a fan).
//
// Same signature and semantics as the reference: a default material is pushed first (ObjLoader.h:415-417),
// vertices are de-duplicated by POSITION only (Vertex.h:31-33,37-51), one material id per index
// (ObjLoader.h:455-460), Vertex.normal.w = *materialVertexOffset (:466), *materialOffset advances by
// 1 + #materials (:417,494).  The per-material 16-entry GGX energy LUT (:351-387) is generated with a
// FIXED seed instead of std::random_device so that scenes are reproducible.
#pragma once
#include <string>
#include <vector>
#include "Vertex.h"

constexpr int LUT_SIZE_THETA = 16;       // ObjLoader.h:23
constexpr int NUM_SAMPLES_MC = 16000;    // ObjLoader.h:24

// ObjLoader.h:351-387 (deterministic: seed = 0x9E3779B9 ^ thetaIdx)
void GenerateEssLUT(Material& mat);
// ObjLoader.h:294-330
float ComputeEss(const XMFLOAT3& N, const XMFLOAT3& V, float roughness, XMFLOAT3 Ks, int numSamples, Material& mat, uint32_t seed);

// What tinyobj's material_t holds beyond the reference's 128-byte Material ("ADD MAP IDs LATER", Vertex.h:21): the remaining
// MTL scalars (Ni is parsed by the reference's loader too but never copied into Material.Ni, ObjLoader.h:428-435) and one texture id per
// map statement.  Carried beside the material table, index-aligned with it; no shader reads any of it.
enum MapSlot { MAP_KA = 0, MAP_KD, MAP_KS, MAP_KE, MAP_NS, MAP_BUMP, MAP_D, MAP_DISP, MAP_REFL, MAP_PR, MAP_PM, MAP_PS, MAP_NORM, kNumMapSlots };
struct MaterialExt {
    float Ni = 1.0f, Ns = 1.0f;               // ior, shininess
    float Pcr = 0.0f, aniso = 0.0f, anisor = 0.0f;   // clearcoat roughness, anisotropy, anisotropy rotation (PBR extension)
    int illum = 0;
    float Ka[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};      // ambient, transmittance (Kt / Tf)
    int map[kNumMapSlots] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // index into the scene's texture-name list, -1 = none
};

class ObjLoader {
public:
    // the same call with the extra outputs: *ext grows in step with *mats (default material first), *textures collects distinct file names
    static void loadObjFileEx(const std::string& inputfile, std::vector<Vertex>* vertices, std::vector<UINT>* indices,
                              std::vector<Material>* mats, std::vector<UINT>* materialIDs, UINT* materialOffset,
                              UINT* materialVertexOffset, std::vector<MaterialExt>* ext, std::vector<std::string>* textures,
                              const std::string& material_search_path = "./");
    // throws std::runtime_error where the reference calls exit(1) (ObjLoader.h:399-404)
    static void loadObjFile(const std::string& inputfile, std::vector<Vertex>* vertices, std::vector<UINT>* indices,
                            std::vector<Material>* mats, std::vector<UINT>* materialIDs, UINT* materialOffset,
                            UINT* materialVertexOffset, const std::string& material_search_path = "./");
};
