// manipulator.h — the part of nv_helpers_dx12::Manipulator that sits on the path boundary
// (Pathtracer/rdn/manipulator.h:50-82, manipulator.cpp:26-32, 305-314): lookat state -> RH view matrix.
// Interactive orbit/pan/dolly input handling is out of scope.
#pragma once
#include "DirectXMathLite.h"

namespace nv_helpers_dx12 {
class Manipulator {
public:
    void setLookat(const XMFLOAT3& cameraPosition, const XMFLOAT3& centerPosition, const XMFLOAT3& upVector);
    void getLookat(XMFLOAT3& eye, XMFLOAT3& center, XMFLOAT3& up) const { eye = m_pos; center = m_int; up = m_up; }
    void setWindowSize(int w, int h) { m_width = w; m_height = h; }
    int getWidth() const { return m_width; }
    int getHeight() const { return m_height; }
    void setRoll(float roll) { m_roll = roll; update(); }
    // 16 floats, column-major (glm::value_ptr of glm::lookAt) — memcpy'd into CameraParams.view (Renderer.cpp:1726-1727)
    const float* getMatrix() const { return m_matrix; }
    static Manipulator& Singleton() { static Manipulator m; return m; }
private:
    void update();
    XMFLOAT3 m_pos{10, 10, 10}, m_int{0, 0, 0}, m_up{0, 1, 0};
    float m_roll = 0.0f;
    int m_width = 1, m_height = 1;
    float m_matrix[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};
#define CameraManip Manipulator::Singleton()
}  // namespace nv_helpers_dx12
