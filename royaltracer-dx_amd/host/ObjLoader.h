// ObjLoader.h — the reference's scene-loader surface (Pathtracer/src/Util/ObjLoader.h:393-495) on top of our
// own OBJ/MTL reader (the reference uses the vendored tinyobjloader v2.0.0; we re-implement the subset it
// relies on: v / vn / f with v, v/vt, v//vn, v/vt/vn and negative indices, o / g, usemtl, mtllib;
// MTL newmtl Kd Ks Ke d Tr Ni Pr Pm Ps Pc; quads split along the shorter diagonal as tinyobj does
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

class ObjLoader {
public:
    // throws std::runtime_error where the reference calls exit(1) (ObjLoader.h:399-404)
    static void loadObjFile(const std::string& inputfile, std::vector<Vertex>* vertices, std::vector<UINT>* indices,
                            std::vector<Material>* mats, std::vector<UINT>* materialIDs, UINT* materialOffset,
                            UINT* materialVertexOffset, const std::string& material_search_path = "./");
};
