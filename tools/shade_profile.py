#!/usr/bin/env python3
"""Where k_shade (general path) spends its time, by code section, and with how many lanes (PROFILE build, see tools/section_profile.py).
usage: python tools/shade_profile.py [sponza|bistro]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTX_LIB_PATH"] = os.path.join(ROOT, "royaltracer-dx_amd", "librtx_hip_prof.so")
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "bistro"
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H = 1920, 1080
c = rt.Context(0); c.upload(sc, W / H); c.clear(W, H)
p = rt.Params(width=W, height=H, spp=4, max_bounces=8, nee_samples=1, flags=1 if kind == "sponza" else 4)
c.render(p)
out = (C.c_ulonglong * 36)()
rt.lib.rtx_debug_sections.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert rt.lib.rtx_debug_sections(out, 1) == 0
c.render(p)
assert rt.lib.rtx_debug_sections(out, 0) == 0
NAMES = ["load hit + state", "surface", "emissive / material setup", "NEE sample", "shadow entry push", "BSDF sample + throughput + RR", "store + compaction"]
LANES = {1: "enter surface (hits)", 3: "enter NEE sample (shading)", 4: "push a shadow ray", 5: "enter BSDF sample", 6: "store a continuing path"}
tot = float(sum(out[:12]))
print(f"k_shade, {kind}: share of wave time by section")
for i, n in enumerate(NAMES):
    print(f"  {n:34s} {100.0 * out[i] / tot:5.1f} %")
for i, what in LANES.items():
    if out[24 + i]:
        print(f"  lanes that {what:30s} {out[12 + i] / out[24 + i]:5.1f} of 64   ({out[24 + i] / 1e6:.2f} M wave-level calls)")
c.close()
