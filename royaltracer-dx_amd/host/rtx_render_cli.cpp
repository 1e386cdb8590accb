// rtx_render — headless CLI over the Renderer facade (replaces the Win32 window loop, Main.cpp:18-27).
// usage: rtx_render [--scene cornell|sponza|bistro|obj] [--obj a.obj,b.obj --mtl dir] [--w 1920 --h 1080]
//                   [--spp 64] [--frames 1] [--bounces 8] [--nee 1] [--lambert] [--out image.{png,ppm,exr}] [--device 0]
//                   [--mode pt|restir]   restir = the reference's shipping frame (3 DispatchRays, Renderer.cpp:646-673: pass 1 + temporal + spatial reuse), one per --frames,
//                   nee 4 / bounces 3 as in Common_v6.hlsl:8-12 unless --nee / --bounces are given; with --gpus N the shards own one tile rectangle each (RTX_FLAG_BLOCK_TILES,
//                   32-px tiles) and exchange history + framebuffer tiles once per frame; [--literal] = the thread-per-pixel kernels instead of the wavefront stages;
//                   [--halo px] with --gpus N --mode restir: the history travels as border strips of `px` pixels between neighbouring rectangles (one send + receive per
//                   neighbour, rtx_restir_pack_halo) instead of the all-gather of 140 B per pixel; the frame line then reports the bytes the busiest rank sent and the stale reads (must be 0);
//                   [--force-gather] with --gpus 1: pack -> RCCL all-gather of a one-rank communicator -> unpack all the same (exercises the collective path on one GPU);
//                   [--orbit deg] moves the camera about the look-at point between frames (exercises the reprojection)
//                   [--spin deg] turns instance 1 (the reference's moving instance, Renderer.cpp:444-449) by `deg` about the vertical axis before every frame after the first: a
//                   transform-only commit = a refit of the resident tree on the GPU, on every rank with --gpus N (the reference refits its TLAS every frame, Renderer.cpp:594)
//                   [--only-rank r] with --gpus N --gather copy: rank r of N alone, through the same host path (measurement on one GPU: tools/shard_time.py native=1)
//                   [--gpus N [--devices 0,1,..] [--gather rccl|copy]]   the native N-GPU frame (MultiGpu.h): one process, N contexts, pixel tiles
//                   round-robin, ONE RCCL all-gather per frame; `--gather copy` replaces the collective by device copies (several ranks on one GPU: tests)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include "Renderer.h"
#include "ImageIO.h"
#include "MultiGpu.h"

int main(int argc, char** argv) {
    std::string scene = "cornell", out, objs, mtl = "./";
    UINT w = 1920, h = 1080, spp = 1, frames = 1, bounces = 8, nee = 1; int device = 0; bool lambert = false;
    int gpus = 1; std::string devlist, gather = "rccl", mode = "pt"; bool literal = false, nee_set = false, bounces_set = false, force_gather = false; float orbit = 0.0f, spin = 0.0f; int only_rank = -1; UINT halo = 0;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char* k) { return !strcmp(argv[i], k) && i + 1 < argc; };
        if (arg("--scene")) scene = argv[++i]; else if (arg("--obj")) { objs = argv[++i]; scene = "obj"; } else if (arg("--mtl")) mtl = argv[++i];
        else if (arg("--w")) w = atoi(argv[++i]); else if (arg("--h")) h = atoi(argv[++i]); else if (arg("--spp")) spp = atoi(argv[++i]);
        else if (arg("--frames")) frames = atoi(argv[++i]); else if (arg("--bounces")) { bounces = atoi(argv[++i]); bounces_set = true; } else if (arg("--nee")) { nee = atoi(argv[++i]); nee_set = true; }
        else if (arg("--mode")) mode = argv[++i]; else if (arg("--orbit")) orbit = (float)atof(argv[++i]); else if (arg("--spin")) spin = (float)atof(argv[++i]); else if (arg("--only-rank")) only_rank = atoi(argv[++i]); else if (!strcmp(argv[i], "--literal")) literal = true; else if (!strcmp(argv[i], "--force-gather")) force_gather = true;
        else if (arg("--halo")) halo = (UINT)atoi(argv[++i]); else if (arg("--gpus")) gpus = atoi(argv[++i]); else if (arg("--devices")) devlist = argv[++i]; else if (arg("--gather")) gather = argv[++i];
        else if (arg("--out")) out = argv[++i]; else if (arg("--device")) device = atoi(argv[++i]); else if (!strcmp(argv[i], "--lambert")) lambert = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (mode != "pt" && mode != "restir") { fprintf(stderr, "--mode must be pt or restir\n"); return 2; }
    const bool restir = mode == "restir";
    if (restir) { if (!nee_set) nee = 4; if (!bounces_set) bounces = 3; }
    // camera of frame f: the scene's eye rotated by f * orbit degrees about the vertical axis through the look-at point
    auto orbit_eye = [&](const Scene& sc, UINT f) {
        const float a = orbit * 3.14159265f / 180.0f * (float)f, cs = cosf(a), sn = sinf(a);
        const float dx = sc.eye.x - sc.center.x, dz = sc.eye.z - sc.center.z;
        return XMFLOAT3(sc.center.x + cs * dx + sn * dz, sc.eye.y, sc.center.z - sn * dx + cs * dz);
    };
    // instance 1 at frame f: its start-up matrix turned by f * spin degrees about the vertical axis (row-vector convention: M * R)
    auto spun = [&](const Scene& sc, UINT f) { return sc.instances[1].transform * XMMatrixRotationAxis({0.f, 1.f, 0.f}, spin * 3.14159265f / 180.0f * (float)f); };
    auto write_image = [&](const std::vector<float>& acc, const std::vector<uint8_t>& px) {
        const bool exr = out.size() > 4 && out.substr(out.size() - 4) == ".exr", png = out.size() > 4 && out.substr(out.size() - 4) == ".png";
        return exr ? WriteEXR(out, acc.data(), w, h) : png ? WritePNG(out, px.data(), w, h) : WritePPM(out, px.data(), w, h);
    };
    if (gpus > 1 || !devlist.empty()) {          // ---- native N-GPU frame (also `--gpus 1 --devices 0`: the same code path with one rank): one process, N contexts on N threads, one all-gather per frame ----
        try {
            std::vector<int> devs;
            if (devlist.empty()) for (int k = 0; k < gpus; k++) devs.push_back(k);
            else { std::stringstream ss(devlist); std::string t; while (std::getline(ss, t, ',')) devs.push_back(atoi(t.c_str())); }
            if ((int)devs.size() != gpus) { fprintf(stderr, "--devices must list %d ordinals\n", gpus); return 2; }
            Scene sc = scene == "cornell" ? MakeCornellBox() : scene == "sponza" ? MakeSponzaClass() : scene == "bistro" ? MakeBistroClass() : Scene();
            if (scene == "obj") { std::vector<std::string> f; std::stringstream ss(objs); std::string t; while (std::getline(ss, t, ',')) f.push_back(t); sc = LoadObjScene(f, mtl); }
            if (scene == "cornell") lambert = true;
            MultiGpuFrame mg(devs, gather == "copy" ? MultiGpuFrame::Gather::COPY : MultiGpuFrame::Gather::RCCL, force_gather, only_rank);      // --force-gather: the collective also with one rank
            mg.SetScene(sc, (float)w / (float)h);
            mg.Clear(w, h);
            rtx_params p{}; p.width = w; p.height = h; p.spp = spp; p.max_bounces = bounces; p.nee_samples = nee; p.rr_start = 3;
            p.tile_size = gpus > 1 ? 32 : 64;      // round-robin deal: 32-px tiles even out the background across 8 ranks (max / mean 1.04 instead of 1.15, tools/shard_time.py)
            p.flags = lambert ? RTX_FLAG_LAMBERT_ONLY : (scene == "bistro" ? RTX_FLAG_TRANSMISSION : 0);
            if (restir) { p.spp = 1; p.flags = (lambert ? RTX_FLAG_LAMBERT_ONLY : 0u) | RTX_FLAG_BLOCK_TILES; p.tile_size = 32; mg.SetOption(RTX_OPT_RESTIR_WAVEFRONT, literal ? 0 : 1); mg.ResetRestir(); mg.SetHaloExchange(halo); }
            float prev_view[16] = {0};
            if (spin != 0.0f && sc.instances.size() < 2) { fprintf(stderr, "--spin needs a scene with an instance 1\n"); return 2; }
            for (UINT f = 0; f < frames; f++) {
                p.sample_base = 1 + f * spp; p.frame_seed = f + 1;
                if (spin != 0.0f && f > 0) { mg.SetInstanceTransform(1, spun(sc, f).data()); if (!restir) mg.Clear(w, h); printf("refit on %d ranks: %.3f ms\n", gpus, mg.LastRefitMs()); }
                if (restir) {
                    nv_helpers_dx12::Manipulator cam; cam.setLookat(orbit_eye(sc, f), sc.center, sc.up);
                    XMMATRIX proj = XMMatrixPerspectiveFovRH(sc.fovY_deg * XM_PI / 180.0f, (float)w / (float)h, sc.zn, sc.zf);
                    mg.SetCamera(cam.getMatrix(), proj.data());
                    if (f == 0) mg.SetCamera(cam.getMatrix(), proj.data());            // previous view = current view for the first frame
                    bool moved = false;                                                // the reference's accumulation reset (RayGen_v6_pass3.hlsl:407-423), as Renderer::UpdateCameraBuffer applies it
                    for (int k = 0; k < 16 && f > 0; k++) moved = moved || fabsf(cam.getMatrix()[k] - prev_view[k]) > 0.00002f;
                    if (moved) mg.Clear(w, h);
                    memcpy(prev_view, cam.getMatrix(), 64);
                    mg.RenderRestir(p);
                } else mg.Render(p);
                double rays = 0; for (int r = 0; r < gpus; r++) { if (only_rank >= 0 && r != only_rank) continue; rtx_stats s = mg.Stats(r); rays += (double)(s.rays_primary + s.rays_extension + s.rays_shadow); }
                printf("frame %u on %d GPUs: %.3f ms (gather included), %.1f Mrays/s\n", f, gpus, mg.LastFrameMs(), rays / (mg.LastFrameMs() * 1e3));
                if (restir && gpus > 1) printf("  history exchange: %s, %.2f MB sent by the busiest rank, stale history reads %llu\n", halo ? "border strips (halo)" : "all-gather", (double)mg.LastExchangeBytes() / 1e6, (unsigned long long)mg.StaleHistoryReads());
            }
            const int rr = only_rank >= 0 ? only_rank : 0;
            if (!out.empty() && !write_image(mg.ReadAccumulation(rr), mg.ReadOutput(rr))) { fprintf(stderr, "error: could not write %s\n", out.c_str()); return 1; }
        } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
        return 0;
    }
    try {
        Renderer r(w, h, "rtx_render");
        r.SetDevice(device);
        if (scene == "cornell") { r.SetScene(MakeCornellBox()); lambert = true; }
        else if (scene == "sponza") r.SetScene(MakeSponzaClass());
        else if (scene == "bistro") r.SetScene(MakeBistroClass());
        else { std::vector<std::string> f; std::stringstream ss(objs); std::string t; while (std::getline(ss, t, ',')) f.push_back(t); r.SetModels(f, mtl); }
        r.Params().spp = spp; r.Params().max_bounces = bounces; r.Params().nee_samples = nee; r.Params().flags = lambert ? RTX_FLAG_LAMBERT_ONLY : (scene == "bistro" ? RTX_FLAG_TRANSMISSION : 0);
        if (restir) {
            r.SetMode(Renderer::Mode::ReSTIR);
            r.RestirParams().nee_samples = nee; r.RestirParams().max_bounces = bounces; r.RestirParams().flags = lambert ? RTX_FLAG_LAMBERT_ONLY : 0u;
        }
        r.OnInit();
        if (restir && rtx_set_option(r.Context(), RTX_OPT_RESTIR_WAVEFRONT, literal ? 0 : 1) != RTX_OK) throw std::runtime_error(rtx_last_error(r.Context()));
        XMFLOAT3 eye0, ctr0, up0; nv_helpers_dx12::CameraManip.getLookat(eye0, ctr0, up0);
        Scene inst0; bool have_inst0 = false;
        for (UINT f = 0; f < frames; f++) {
            if (spin != 0.0f && f > 0) {
                if (!have_inst0) { inst0 = scene == "cornell" ? MakeCornellBox() : scene == "sponza" ? MakeSponzaClass() : scene == "bistro" ? MakeBistroClass() : Scene(); if (scene == "obj") { std::vector<std::string> fl; std::stringstream ss(objs); std::string t; while (std::getline(ss, t, ',')) fl.push_back(t); inst0 = LoadObjScene(fl, mtl); } have_inst0 = true; }
                if (inst0.instances.size() < 2) { fprintf(stderr, "--spin needs a scene with an instance 1\n"); return 2; }
                r.SetInstanceTransform(1, spun(inst0, f));
                if (!restir) rtx_clear_accum(r.Context(), w, h);               // a moved scene restarts the progressive accumulation (the ReSTIR frame reprojects instead)
            }
            if (orbit != 0.0f) { Scene tmp; tmp.eye = eye0; tmp.center = ctr0; tmp.up = up0; nv_helpers_dx12::CameraManip.setLookat(orbit_eye(tmp, f), ctr0, up0); }
            r.OnUpdate(); r.Params().sample_base = 1 + f * spp; r.OnRender();
            rtx_stats s = r.Stats();
            double rays = (double)(s.rays_primary + s.rays_extension + s.rays_shadow);
            printf("frame %u: %.3f ms, %.1f Mrays/s (primary %llu, extension %llu, shadow %llu)\n", f, s.render_ms, rays / (s.render_ms * 1e3),
                   (unsigned long long)s.rays_primary, (unsigned long long)s.rays_extension, (unsigned long long)s.rays_shadow);
        }
        if (!out.empty()) {
            const bool exr = out.size() > 4 && out.substr(out.size() - 4) == ".exr", png = out.size() > 4 && out.substr(out.size() - 4) == ".png";
            bool ok;
            if (exr) { std::vector<float> acc = r.ReadAccumulation(); ok = WriteEXR(out, acc.data(), w, h); }
            else { std::vector<uint8_t> px = r.ReadOutput(); ok = png ? WritePNG(out, px.data(), w, h) : WritePPM(out, px.data(), w, h); }
            if (!ok) { fprintf(stderr, "error: could not write %s\n", out.c_str()); return 1; }
        }
        r.OnDestroy();
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
