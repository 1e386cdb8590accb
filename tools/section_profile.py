#!/usr/bin/env python3
"""Where the fused tiny-scene bounce kernel spends its time, by code section.

Needs the PROFILE build (`make -C royaltracer-dx_amd PROFILE=1` -> librtx_hip_prof.so: s_memtime deltas per section, summed per
wave).  Renders Cornell frames of configs[1] and prints each section's share.  Tooling only; the product library has no counters.
usage: python tools/section_profile.py [frames]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTX_LIB_PATH"] = os.path.join(ROOT, "royaltracer-dx_amd", "librtx_hip_prof.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

NAMES = ["load state", "closest: pre-test", "closest: exact tests", "surface + emissive", "NEE sample", "shadow push + barrier",
         "shadow: pre-test", "shadow: exact tests", "barrier 2", "radiance + BSDF sample + store", "-", "-"]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rt = graft.load_package()
    dev = torch.device("cuda", 0)
    scene = rt.Scene.cornell()
    ctx = rt.Context(0)
    W, H = 1920, 1080
    ctx.upload(scene, W / H)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
    p = rt.Params(width=W, height=H, spp=64, sample_base=1, max_bounces=8, nee_samples=1, rr_start=3, frame_seed=1, flags=1,
                  tile_size=64, shard_rank=0, shard_count=1)
    ctx.render(p)                                   # warm-up
    out = (C.c_ulonglong * 36)()
    rt.lib.rtx_debug_sections.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert rt.lib.rtx_debug_sections(out, 1) == 0
    for i in range(frames):
        p.frame_seed = 2 + i
        ctx.render(p)
    assert rt.lib.rtx_debug_sections(out, 0) == 0
    tot = float(sum(out[:12]))
    print(f"k_bounce_small, {frames} frame(s) of Cornell 1080p 64 spp 8 bounces: share of wave time by section; active lanes where counted")
    LANES = {2: "per exact-test round (closest)", 3: "surface reconstruction", 4: "NEE sample", 7: "per exact-test round (shadow)", 9: "BSDF sample", 10: "store path"}
    for i, (n, v) in enumerate(zip(NAMES, out[:12])):
        if v:
            print(f"  {n:34s} {100.0 * v / tot:5.1f} %")
    for i, what in LANES.items():
        if out[24 + i]:
            print(f"  lanes {what:34s} {out[12 + i] / out[24 + i]:5.1f} of 64   ({out[24 + i] / 1e6:.1f} M wave-level calls)")
    ctx.close()


if __name__ == "__main__":
    main()
