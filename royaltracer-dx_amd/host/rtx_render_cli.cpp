// rtx_render — headless CLI over the Renderer facade (replaces the Win32 window loop, Main.cpp:18-27).
// usage: rtx_render [--scene cornell|sponza|bistro|obj] [--obj a.obj,b.obj --mtl dir] [--w 1920 --h 1080]
//                   [--spp 64] [--frames 1] [--bounces 8] [--nee 1] [--lambert] [--out image.{png,ppm,exr}] [--device 0]
//                   [--gpus N [--devices 0,1,..] [--gather rccl|copy]]   the native N-GPU frame (MultiGpu.h): one process, N contexts, pixel tiles
//                   round-robin, ONE RCCL all-gather per frame; `--gather copy` replaces the collective by device copies (several ranks on one GPU: tests)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include "Renderer.h"
#include "ImageIO.h"
#include "MultiGpu.h"

int main(int argc, char** argv) {
    std::string scene = "cornell", out, objs, mtl = "./";
    UINT w = 1920, h = 1080, spp = 1, frames = 1, bounces = 8, nee = 1; int device = 0; bool lambert = false;
    int gpus = 1; std::string devlist, gather = "rccl";
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char* k) { return !strcmp(argv[i], k) && i + 1 < argc; };
        if (arg("--scene")) scene = argv[++i]; else if (arg("--obj")) { objs = argv[++i]; scene = "obj"; } else if (arg("--mtl")) mtl = argv[++i];
        else if (arg("--w")) w = atoi(argv[++i]); else if (arg("--h")) h = atoi(argv[++i]); else if (arg("--spp")) spp = atoi(argv[++i]);
        else if (arg("--frames")) frames = atoi(argv[++i]); else if (arg("--bounces")) bounces = atoi(argv[++i]); else if (arg("--nee")) nee = atoi(argv[++i]);
        else if (arg("--gpus")) gpus = atoi(argv[++i]); else if (arg("--devices")) devlist = argv[++i]; else if (arg("--gather")) gather = argv[++i];
        else if (arg("--out")) out = argv[++i]; else if (arg("--device")) device = atoi(argv[++i]); else if (!strcmp(argv[i], "--lambert")) lambert = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
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
            MultiGpuFrame mg(devs, gather == "copy" ? MultiGpuFrame::Gather::COPY : MultiGpuFrame::Gather::RCCL);
            mg.SetScene(sc, (float)w / (float)h);
            mg.Clear(w, h);
            rtx_params p{}; p.width = w; p.height = h; p.spp = spp; p.max_bounces = bounces; p.nee_samples = nee; p.rr_start = 3; p.tile_size = 64;
            p.flags = lambert ? RTX_FLAG_LAMBERT_ONLY : (scene == "bistro" ? RTX_FLAG_TRANSMISSION : 0);
            for (UINT f = 0; f < frames; f++) {
                p.sample_base = 1 + f * spp; p.frame_seed = f + 1;
                mg.Render(p);
                double rays = 0; for (int r = 0; r < gpus; r++) { rtx_stats s = mg.Stats(r); rays += (double)(s.rays_primary + s.rays_extension + s.rays_shadow); }
                printf("frame %u on %d GPUs: %.3f ms (gather included), %.1f Mrays/s\n", f, gpus, mg.LastFrameMs(), rays / (mg.LastFrameMs() * 1e3));
            }
            if (!out.empty() && !write_image(mg.ReadAccumulation(0), mg.ReadOutput(0))) { fprintf(stderr, "error: could not write %s\n", out.c_str()); return 1; }
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
        r.OnInit();
        for (UINT f = 0; f < frames; f++) {
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
