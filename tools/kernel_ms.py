#!/usr/bin/env python3
"""Tooling: frame time and per-kernel-class time of one BVH frame (1080p, 16 spp, 8 bounces) for the library in RTX_LIB_PATH.
usage: python tools/kernel_ms.py [sponza|bistro] [opt=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "sponza"
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H = 1920, 1080
c = rt.Context(0)
c.set_option(rt.OPT_KERNEL_TIMING, 1)
for a in sys.argv[2:]:
    k, v = a.split("="); c.set_option(int(k), int(v))
c.upload(sc, W / H)
p = rt.Params(width=W, height=H, spp=16, max_bounces=8, nee_samples=1, flags=1 if kind == "sponza" else 4)
for rep in range(3):
    c.clear(W, H); c.render(p)
    st = c.stats()
    if rep:
        print(f"{os.path.basename(rt.LIB_PATH)} {kind} {' '.join(sys.argv[2:])}: frame {st.render_ms:.2f} ms; " + ", ".join(f"{rt.KERNEL_NAMES[i]} {st.kernel_ms[i]:.2f}" for i in rt.KERNEL_NAMES if st.kernel_launches[i] > 0))
c.close()
