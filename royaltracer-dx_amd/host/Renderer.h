// Renderer.h — headless counterpart of the reference's Renderer class (Pathtracer/rdn/Renderer.h:46-51):
// same construction and OnInit / OnUpdate / OnRender / OnDestroy life cycle, no window, no DX12.  Everything
// the reference's Renderer feeds its shaders goes through the C-ABI of include/rtx.h instead.
#pragma once
#include <string>
#include <vector>
#include "Scenes.h"
#include "manipulator.h"

class Renderer {
public:
    Renderer(UINT width, UINT height, std::string name);
    ~Renderer();
    // headless configuration (the reference hard-codes these: Renderer.cpp:363, 46-48; Common_v6.hlsl:8-12)
    void SetModels(const std::vector<std::string>& obj_files, const std::string& mtl_dir) { m_models = obj_files; m_mtlDir = mtl_dir; m_haveScene = false; }
    void SetScene(const Scene& s) { m_scene = s; m_haveScene = true; }
    void SetDevice(int ordinal) { m_device = ordinal; }
    rtx_params& Params() { return m_params; }
    // What OnRender issues.  ReSTIR = the reference's shipping frame, its three DispatchRays (Renderer.cpp:646-673: RayGen = pass 1, RayGen2 = temporal reuse,
    // RayGen3 = spatial reuse + shade) with its shader defines nee_samples 4 / bounces 3 (Common_v6.hlsl:8-12) unless RestirParams() is changed; one frame per
    // OnRender, history carried in the context.  PathTracer (the default of this headless build) = Params().spp samples of the bounce-loop estimator.
    enum class Mode { PathTracer, ReSTIR };
    void SetMode(Mode m) { m_mode = m; }
    Mode GetMode() const { return m_mode; }
    rtx_params& RestirParams() { return m_restir; }
    // m_instances[i].second = ... of the reference's OnUpdate (Renderer.cpp:444-449): the matrix takes effect in the next OnUpdate, which hands it to the context and
    // re-commits — a transform-only commit, i.e. a REFIT of the resident tree on the GPU (the reference refits its TLAS every frame, Renderer.cpp:594, 2091-2121)
    void SetInstanceTransform(UINT instance, const XMMATRIX& objectToWorld);
    double LastRefitMs() const { return m_refitMs; }

    void OnInit();      // Renderer.cpp:44-103: camera lookat, load models, build acceleration structures, upload
    void OnUpdate();    // Renderer.cpp:431-452: camera buffer, instance 1 rotation, instance properties
    void OnRender();    // Renderer.cpp:468-506 / 556-715: one frame = Params().spp samples per pixel, accumulated
    void OnDestroy();   // Renderer.cpp:546-552

    UINT GetWidth() const { return m_width; }
    UINT GetHeight() const { return m_height; }
    const std::string& GetTitle() const { return m_title; }
    uint32_t FrameIndex() const { return m_time; }
    std::vector<float> ReadAccumulation();          // gPermanentData (RGBA32F)
    std::vector<uint8_t> ReadOutput();              // the DISPLAYED layer of gOutput (RGBA8): what the reference copies to the back buffer (Renderer.cpp:690-698)
    void OnKeyUp(uint8_t key);                      // 'C' cycles m_displayLevels (Renderer.cpp:748-754); other keys do nothing here (VK_SPACE toggles a raster path that does not exist)
    UINT CurrentDisplayLayer() const { return m_displayLevels[m_currentDisplayLevel]; }
    rtx_stats Stats();
    rtx_ctx* Context() { return m_ctx; }
private:
    void UpdateCameraBuffer();                      // Renderer.cpp:1722-1768
    void Check(int rc, const char* what);
    UINT m_width, m_height; float m_aspectRatio; std::string m_title;
    std::vector<std::string> m_models; std::string m_mtlDir;
    Scene m_scene; bool m_haveScene = false;
    int m_device = 0;
    rtx_ctx* m_ctx = nullptr;
    rtx_params m_params{}, m_restir{};
    Mode m_mode = Mode::PathTracer;
    uint32_t m_time = 0;                             // Renderer.h: m_time
    UINT m_currentDisplayLevel = 0;                  // Renderer.h:298
    std::vector<UINT> m_displayLevels = {0, 10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 23, 24, 25, 26, 27, 28};   // Renderer.h:299
    float m_prevView[16]; bool m_havePrev = false;   // m_prevViewMatrix
    std::vector<UINT> m_movedInstances; double m_refitMs = 0.0;
};
