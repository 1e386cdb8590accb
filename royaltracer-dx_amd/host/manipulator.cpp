#include "manipulator.h"
#include <math.h>

namespace nv_helpers_dx12 {
namespace {
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline V3 normalize(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return {a.x * inv, a.y * inv, a.z * inv}; }   // glm: v * inversesqrt(dot(v,v))
}
void Manipulator::setLookat(const XMFLOAT3& pos, const XMFLOAT3& center, const XMFLOAT3& up) {      // manipulator.cpp:26-32
    m_pos = pos; m_int = center; m_up = up; update();
}
// manipulator.cpp:305-314: glm::lookAt (right-handed, glm/gtc/matrix_transform.inl:521-545)
void Manipulator::update() {
    V3 eye{m_pos.x, m_pos.y, m_pos.z}, cen{m_int.x, m_int.y, m_int.z}, up{m_up.x, m_up.y, m_up.z};
    V3 f = normalize(sub(cen, eye)), s = normalize(cross(f, up)), u = cross(s, f);
    float* M = m_matrix;    // column-major: M[c*4 + r]
    M[0] = s.x; M[4] = s.y; M[8] = s.z;   M[12] = -dot(s, eye);
    M[1] = u.x; M[5] = u.y; M[9] = u.z;   M[13] = -dot(u, eye);
    M[2] = -f.x; M[6] = -f.y; M[10] = -f.z; M[14] = dot(f, eye);
    M[3] = 0.0f; M[7] = 0.0f; M[11] = 0.0f; M[15] = 1.0f;
    if (m_roll != 0.0f) {                // m_matrix = m_matrix * rotate(roll, z)
        float c = cosf(m_roll), sn = sinf(m_roll);
        for (int r = 0; r < 4; r++) { float a = M[r], b = M[4 + r]; M[r] = a * c + b * sn; M[4 + r] = -a * sn + b * c; }
    }
}
}  // namespace nv_helpers_dx12
