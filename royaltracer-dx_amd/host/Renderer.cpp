#include "Renderer.h"
#include <chrono>
#include <cmath>
#include <cstring>
#include <stdexcept>

using nv_helpers_dx12::Manipulator;

Renderer::Renderer(UINT width, UINT height, std::string name) : m_width(width), m_height(height), m_aspectRatio((float)width / (float)height), m_title(std::move(name)) {
    m_params.width = width; m_params.height = height;
    m_params.spp = 1; m_params.sample_base = 1;
    m_params.max_bounces = 8; m_params.nee_samples = 1; m_params.rr_start = 3;     // rr_threshold = 3, RayGen.hlsl:69
    m_params.frame_seed = 0; m_params.flags = 0; m_params.tile_size = 64; m_params.shard_rank = 0; m_params.shard_count = 1;
    m_restir = m_params;
    m_restir.spp = 1;                                                               // one frame per OnRender
    m_restir.nee_samples = 4; m_restir.max_bounces = 3;                             // nee_samples / nee_samples_DI = 4, bounces = 3 (Common_v6.hlsl:8-12)
}
Renderer::~Renderer() { OnDestroy(); }

void Renderer::Check(int rc, const char* what) {
    if (rc != RTX_OK) throw std::runtime_error(std::string(what) + ": " + rtx_last_error(m_ctx));   // ThrowIfFailed, DXSampleHelper.h:17-23
}

void Renderer::OnInit() {
    if (!m_haveScene) {
        if (m_models.empty()) throw std::logic_error("Renderer::OnInit: no models (SetModels) and no scene (SetScene)");
        m_scene = LoadObjScene(m_models, m_mtlDir);                                 // CreateVB per model, Renderer.cpp:363-370
        m_haveScene = true;
    }
    CameraManip.setWindowSize((int)m_width, (int)m_height);                         // Renderer.cpp:45
    CameraManip.setLookat(m_scene.eye, m_scene.center, m_scene.up);                 // Renderer.cpp:46-48
    Check(rtx_create(m_device, &m_ctx), "rtx_create");
    Check(rtx_set_materials(m_ctx, m_scene.materials.data(), (uint32_t)m_scene.materials.size()), "rtx_set_materials");
    for (const SceneModel& m : m_scene.models) {
        uint32_t id;
        Check(rtx_add_mesh(m_ctx, m.vertices.data(), (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(), m.materialIDs.data(), &id), "rtx_add_mesh");
    }
    for (const SceneInstance& in : m_scene.instances) { uint32_t id; Check(rtx_add_instance(m_ctx, in.model, in.transform.data(), &id), "rtx_add_instance"); }
    Check(rtx_commit_scene(m_ctx), "rtx_commit_scene");                             // CreateAccelerationStructures, Renderer.cpp:893-946
    Check(rtx_clear_accum(m_ctx, m_width, m_height), "rtx_clear_accum");
}

void Renderer::UpdateCameraBuffer() {
    float view[16];
    memcpy(view, CameraManip.getMatrix(), 64);                                      // Renderer.cpp:1726-1727
    XMMATRIX proj = XMMatrixPerspectiveFovRH(60.0f * XM_PI / 180.0f, m_aspectRatio, 0.1f, 1000.0f);   // :1730-1731
    Check(rtx_set_camera(m_ctx, view, proj.data()), "rtx_set_camera");
    // accumulation reset when the view changed by more than s_bias in any element (RayGen_v6_pass3.hlsl:407-423)
    bool different = !m_havePrev;
    if (m_havePrev) for (int i = 0; i < 16; i++) if (fabsf(view[i] - m_prevView[i]) > 0.00002f) { different = true; break; }
    if (different) Check(rtx_clear_accum(m_ctx, m_width, m_height), "rtx_clear_accum");
    memcpy(m_prevView, view, 64); m_havePrev = true;                                // :1766-1767
}

void Renderer::SetInstanceTransform(UINT instance, const XMMATRIX& objectToWorld) {
    if (instance >= m_scene.instances.size()) throw std::out_of_range("Renderer::SetInstanceTransform: no such instance");
    m_scene.instances[instance].transform = objectToWorld;
    for (UINT i : m_movedInstances) if (i == instance) return;
    m_movedInstances.push_back(instance);
}

void Renderer::OnUpdate() {
    UpdateCameraBuffer();
    m_time++;                                                                       // Renderer.cpp:438
    // The reference re-sets instance 1 every frame (Renderer.cpp:444-449), rebuilds InstanceProperties (:451, 2091-2121) and refits the TLAS (:594).  Here an instance
    // moves when SetInstanceTransform was called since the last update: its matrix goes to the context (which keeps the old one as prevObjectToWorld for the temporal
    // pass) and ONE transform-only commit refits the resident tree (k_refit_tris / k_refit_nodes).  Nothing moved: nothing to do.
    if (!m_movedInstances.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        for (UINT i : m_movedInstances) Check(rtx_set_instance_transform(m_ctx, i, m_scene.instances[i].transform.data()), "rtx_set_instance_transform");
        Check(rtx_commit_scene(m_ctx), "rtx_commit_scene");
        m_movedInstances.clear();
        m_refitMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
}

void Renderer::OnRender() {
    if (m_mode == Mode::ReSTIR) {                                                   // PopulateCommandList: DispatchRays x 3 (Renderer.cpp:646-673)
        m_restir.width = m_width; m_restir.height = m_height;
        m_restir.frame_seed = m_time;
        Check(rtx_render_restir(m_ctx, &m_restir), "rtx_render_restir");
        return;
    }
    m_params.frame_seed = m_time;                                                   // stands in for uint(time), Renderer.cpp:1754-1760
    Check(rtx_render(m_ctx, &m_params), "rtx_render");
}

void Renderer::OnDestroy() { if (m_ctx) { rtx_destroy(m_ctx); m_ctx = nullptr; } }

std::vector<float> Renderer::ReadAccumulation() {
    std::vector<float> v((size_t)m_width * m_height * 4);
    Check(rtx_read_accum(m_ctx, v.data(), v.size() * 4), "rtx_read_accum");
    return v;
}
std::vector<uint8_t> Renderer::ReadOutput() {
    std::vector<uint8_t> v((size_t)m_width * m_height * 4);
    Check(rtx_read_layer(m_ctx, m_displayLevels[m_currentDisplayLevel], m_width, m_height, v.data(), v.size()), "rtx_read_layer");   // selectedLayer, Renderer.cpp:690
    return v;
}
void Renderer::OnKeyUp(uint8_t key) {
    if (key == 'C') m_currentDisplayLevel = (m_currentDisplayLevel + 1) % (UINT)m_displayLevels.size();                             // Renderer.cpp:750-753
}
rtx_stats Renderer::Stats() { rtx_stats s{}; Check(rtx_get_stats(m_ctx, &s), "rtx_get_stats"); return s; }
