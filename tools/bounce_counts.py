#!/usr/bin/env python3
"""Tooling: rays per bounce of a BVH workload (queue lengths entering each bounce), from runs with max_bounces = 1..8.
usage: python tools/bounce_counts.py [sponza|bistro]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import __graft_entry__ as graft
rt = graft.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "sponza"
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H = 1920, 1080
c = rt.Context(0); c.upload(sc, W / H)
prev_ext = prev_sh = 0
for mb in range(1, 9):
    p = rt.Params(width=W, height=H, spp=4, max_bounces=mb, nee_samples=1, flags=1 if kind == "sponza" else 4)
    c.clear(W, H); c.render(p); st = c.stats()
    print(f"{kind} bounce {mb - 1}: {(st.rays_primary if mb == 1 else st.rays_extension - prev_ext) / st.rays_primary:.3f} of the primary rays traced, shadow rays {(st.rays_shadow - prev_sh) / st.rays_primary:.3f}")
    prev_ext, prev_sh = st.rays_extension, st.rays_shadow
c.close()
