// Scenes.h — host-side scene container + the synthetic scenes BASELINE.json names.  None of Cornell / Sponza /
// Bistro ship with the reference (only garage.obj / monke.obj do, Renderer.cpp:363), so they are generated here,
// deterministically, in the reference's own data model: a global Material table, per-model Vertex / index /
// materialID arrays (ObjLoader.h:393-495) and an instance list of (model, matrix) pairs (Renderer.h:106).
#pragma once
#include <string>
#include <vector>
#include "Vertex.h"
#include "ObjLoader.h"
#include "../../include/rtx.h"

struct SceneModel { std::vector<Vertex> vertices; std::vector<UINT> indices; std::vector<UINT> materialIDs; };
struct SceneInstance { UINT model; XMMATRIX transform; };
struct Scene {
    std::string name;
    std::vector<Material> materials;         // [default_0, mats of model 0..., default_1, ...]  (ObjLoader.h:415-417,494)
    std::vector<MaterialExt> materialExt;    // index-aligned with `materials` when the scene came from OBJ / MTL files (empty otherwise): the MTL fields and
    std::vector<std::string> textures;       // texture map ids the 128-byte record has no room for (Vertex.h:21 "ADD MAP IDs LATER"), and the distinct map file names
    std::vector<SceneModel> models;
    std::vector<SceneInstance> instances;
    XMFLOAT3 eye{0, 0, 1}, center{0, 0, 0}, up{0, 1, 0};
    float fovY_deg = 60.0f, zn = 0.1f, zf = 1000.0f;     // Renderer.cpp:1730-1731
    size_t triangles() const { size_t n = 0; for (auto& i : instances) n += models[i.model].indices.size() / 3; return n; }
};

Scene MakeCornellBox();                                             // 32 triangles, 2 emissive
// hard = false: uniformly tessellated stand-ins (every surface a grid of centimetre quads: the EASY case for a BVH builder); hard = true: the same shell, materials, light,
// camera and triangle budget with the size distribution of the real assets — a few triangles metres long beside ornament tessellated to millimetres, long thin trims,
// overlapping cloth, foliage (size ratio > 1000 : 1) — the case tree quality is FOR (Scenes.cpp: sponza_hard_build)
Scene MakeSponzaClass(uint32_t target_tris = 262144, uint32_t seed = 260, bool hard = false);
Scene MakeBistroClass(uint32_t target_tris = 3800000, uint32_t seed = 3800, bool hard = false);
// the reference's own startup scene: each file through ObjLoader::loadObjFile, one instance per model,
// instance 1 rotated 1.57 rad about Y (Renderer.cpp:363-407, 444-449)
Scene LoadObjScene(const std::vector<std::string>& files, const std::string& mtl_dir);
// rtx_set_materials / rtx_add_mesh / rtx_add_instance / rtx_commit_scene / rtx_set_camera for `aspect`
int UploadScene(const Scene&, rtx_ctx*, float aspect);
void SceneViewProj(const Scene&, float aspect, float view[16], float proj[16]);
