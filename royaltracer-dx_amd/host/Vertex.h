// Vertex.h — Material (128 B) and Vertex (28 B) exactly as Pathtracer/src/Components/Vertex.h:14-35 declares
// them (same field names, defaults, constructors and position-only equality/hash).
#pragma once
#include <functional>
#include "DirectXMathLite.h"

struct Material {
    XMFLOAT4 Kd = {1, 1, 1, 1};
    XMFLOAT3 Ks = {1, 1, 1}; float Ni = 1;
    XMFLOAT3 Ke = {0, 0, 0}; float pad0 = 0;
    XMFLOAT4 Pr_Pm_Ps_Pc = {0, 0, 0, 0};
    float LUT[16] = {0};
    Material() {}
    Material(XMFLOAT4 kd, XMFLOAT4 pr_pm_ps_pc) : Kd(kd), Pr_Pm_Ps_Pc(pr_pm_ps_pc) {}
};
static_assert(sizeof(Material) == 128, "Material must be 128 bytes (Vertex.h:14-23)");

struct Vertex {
    XMFLOAT3 position;
    XMFLOAT4 normal_material = {1, 1, 1, 0};   // xyz normal (0,0,0 = flat), w = base of this model in materialIDs[]
    Vertex() {}
    Vertex(XMFLOAT3 pos, XMFLOAT4 norm) : position(pos), normal_material(norm.x, norm.y, norm.z, norm.w) {}
    bool operator==(const Vertex& o) const { return position.x == o.position.x && position.y == o.position.y && position.z == o.position.z; }
};
static_assert(sizeof(Vertex) == 28, "Vertex must be 28 bytes (Vertex.h:25-35)");

namespace std {
template <> struct hash<Vertex> {
    size_t operator()(const Vertex& v) const {     // Vertex.h:37-51: position only
        size_t h = std::hash<float>()(v.position.x) ^ std::hash<float>()(v.position.y) << 1 ^ std::hash<float>()(v.position.z) << 2;
        return h << 1;
    }
};
}
