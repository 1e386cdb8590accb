#!/usr/bin/env python3
"""Lane utilisation of the persistent BVH traversal by iteration type (PROFILE build: make -C royaltracer-dx_amd PROFILE=1).
usage: python tools/traversal_profile.py [sponza|bistro] [trace_sched]     (RTX_LIB_PATH picks another PROFILE build, e.g. ..._prof_p3.so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTX_LIB_PATH"] = os.environ.get("RTX_LIB_PATH") or os.path.join(ROOT, "royaltracer-dx_amd", "librtx_hip_prof.so")
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "sponza"
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H = 1920, 1080
c = rt.Context(0); c.upload(sc, W / H); c.clear(W, H)
if len(sys.argv) > 2: c.set_option(rt.OPT_TRACE_SCHED, int(sys.argv[2]))
p = rt.Params(width=W, height=H, spp=4, max_bounces=8, nee_samples=1, flags=1 if kind == "sponza" else 4)
c.render(p)
out = (C.c_ulonglong * 8)()
rt.lib.rtx_debug_traversal.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert rt.lib.rtx_debug_traversal(out, 1) == 0
c.render(p)
assert rt.lib.rtx_debug_traversal(out, 0) == 0
ni, nl, ti, tl, busy, it, both = [float(x) for x in out[:7]]
max_sp = int(out[7])
st = c.stats(); rays = st.rays_primary + st.rays_extension
print(f"{os.path.basename(os.environ['RTX_LIB_PATH'])} sched {sys.argv[2] if len(sys.argv) > 2 else 'default'}")
print(f"{kind}: closest-hit traversal, {rays / 1e6:.1f} M rays, {it / 1e6:.1f} M wave iterations")
print(f"  node iterations {ni / it:.2%} of all, lanes taking part {nl / ni:.1f} of 64;  triangle iterations {ti / it:.2%}, lanes {tl / ti:.1f} of 64")
print(f"  busy lanes (holding an unfinished ray) {busy / it:.1f} of 64; lanes that could do either step {both / it:.1f}")
print(f"  per ray: {nl / rays:.2f} node steps, {tl / rays:.2f} triangle tests")
print(f"  deepest traversal stack seen: {max_sp} entries")
c.close()
