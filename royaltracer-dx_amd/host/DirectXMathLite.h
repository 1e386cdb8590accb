// DirectXMathLite.h — the handful of DirectXMath POD types the reference's scene API is written against
// (XMFLOAT3 / XMFLOAT4 / XMMATRIX as used by Vertex.h, ObjLoader.h, Renderer.h), so that host code written
// for the reference's loader / material / camera surface compiles unchanged on Linux.
#pragma once
#include <stdint.h>
#include <string.h>

typedef uint32_t UINT;

struct XMFLOAT3 {
    float x, y, z;
    XMFLOAT3() : x(0), y(0), z(0) {}
    XMFLOAT3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit XMFLOAT3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
};
struct XMFLOAT4 {
    float x, y, z, w;
    XMFLOAT4() : x(0), y(0), z(0), w(0) {}
    XMFLOAT4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
};
// Row-vector 4x4 stored row-major, exactly DirectXMath's XMMATRIX / XMFLOAT4X4 memory image.  Read as 16
// floats it is the column-major column-vector layout of include/rtx.h (what the reference uploads raw).
struct XMMATRIX {
    float m[4][4];
    const float* data() const { return &m[0][0]; }
    float* data() { return &m[0][0]; }
};
inline XMMATRIX XMMatrixIdentity() { XMMATRIX r; memset(&r, 0, sizeof(r)); r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f; return r; }
XMMATRIX XMMatrixMultiply(const XMMATRIX& a, const XMMATRIX& b);              // row-vector convention: v * a * b
XMMATRIX XMMatrixScaling(float sx, float sy, float sz);
XMMATRIX XMMatrixTranslation(float x, float y, float z);
XMMATRIX XMMatrixRotationAxis(const XMFLOAT3& axis, float angle);
XMMATRIX XMMatrixPerspectiveFovRH(float fovY, float aspect, float zn, float zf);
inline XMMATRIX operator*(const XMMATRIX& a, const XMMATRIX& b) { return XMMatrixMultiply(a, b); }
constexpr float XM_PI = 3.141592654f;
