#include "DirectXMathLite.h"
#include <math.h>

XMMATRIX XMMatrixMultiply(const XMMATRIX& a, const XMMATRIX& b) {
    XMMATRIX r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
XMMATRIX XMMatrixScaling(float sx, float sy, float sz) { XMMATRIX r = XMMatrixIdentity(); r.m[0][0] = sx; r.m[1][1] = sy; r.m[2][2] = sz; return r; }
XMMATRIX XMMatrixTranslation(float x, float y, float z) { XMMATRIX r = XMMatrixIdentity(); r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r; }
// row-vector rotation about a (normalised) axis, as DirectXMath's XMMatrixRotationNormal builds it
XMMATRIX XMMatrixRotationAxis(const XMFLOAT3& axis, float angle) {
    float len = sqrtf(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
    float x = axis.x / len, y = axis.y / len, z = axis.z / len;
    float s = sinf(angle), c = cosf(angle), t = 1.0f - c;
    XMMATRIX r = XMMatrixIdentity();
    r.m[0][0] = t * x * x + c;     r.m[0][1] = t * x * y + s * z; r.m[0][2] = t * x * z - s * y;
    r.m[1][0] = t * x * y - s * z; r.m[1][1] = t * y * y + c;     r.m[1][2] = t * y * z + s * x;
    r.m[2][0] = t * x * z + s * y; r.m[2][1] = t * y * z - s * x; r.m[2][2] = t * z * z + c;
    return r;
}
// XMMatrixPerspectiveFovRH (Renderer.cpp:1730-1731): Height = cos/sin of half the fov, Width = Height/aspect,
// fRange = zf/(zn - zf); rows (W,0,0,0) (0,H,0,0) (0,0,fRange,-1) (0,0,fRange*zn,0)
XMMATRIX XMMatrixPerspectiveFovRH(float fovY, float aspect, float zn, float zf) {
    float s = sinf(0.5f * fovY), c = cosf(0.5f * fovY);
    float h = c / s, w = h / aspect, fr = zf / (zn - zf);
    XMMATRIX r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = 0.0f;
    r.m[0][0] = w; r.m[1][1] = h; r.m[2][2] = fr; r.m[2][3] = -1.0f; r.m[3][2] = fr * zn;
    return r;
}
