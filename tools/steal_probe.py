#!/usr/bin/env python3
"""Tooling: per-kernel-class time of one frame with work stealing off / on.  usage: python tools/steal_probe.py [sponza|bistro] [max_bounces]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "sponza"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H = 1920, 1080
c = rt.Context(0); c.upload(sc, W / H)
c.set_option(rt.OPT_KERNEL_TIMING, 1)
p = rt.Params(width=W, height=H, spp=16, max_bounces=mb, nee_samples=1, flags=1 if kind == "sponza" else 4)
OPT = int(os.environ.get("PROBE_OPT", rt.OPT_WORK_STEALING))
for steal in (0, 1, 0, 1):
    c.set_option(OPT, steal)
    c.clear(W, H); c.render(p); c.clear(W, H); c.render(p)
    st = c.stats()
    print(f"{kind} mb={mb} opt {OPT}={steal}: frame {st.render_ms:.2f} ms; " + ", ".join(f"{rt.KERNEL_NAMES[i]} {st.kernel_ms[i]:.2f}" for i in rt.KERNEL_NAMES if st.kernel_launches[i] > 0))
c.close()
