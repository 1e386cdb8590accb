#!/usr/bin/env python3
"""Diagnostic: the sharded-ReSTIR parity scenario of tests/test_gpu_parity.py::test_restir_on_shards_equals_the_unsharded_frames[4-1-1] under builder / any-hit-order settings;
prints the pixels of each shard that differ from the unsharded frames.  usage: python tools/diag_shard.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as graft
rt = graft.load_package()
from royaltracer_dx_amd import sharding
gd = os.path.join(ROOT, "tests", "golden")
sc = rt.Scene.from_obj([os.path.join(gd, "garage.obj"), os.path.join(gd, "monke.obj")], gd + "/")
W, H, TS, nshards, blocks = 160, 96, 32, 4, 1
cams = [rt.lookat((-1.5 + 0.05 * k, 1.5, 3.5 - 0.04 * k), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)) for k in range(3)]
proj = rt.perspective_fov_rh(np.radians(60.0), W / H, 0.1, 1000.0)
base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=rt.FLAG_BLOCK_TILES if blocks else 0, tile_size=TS)
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
for reinsert, order, wave in ((0, 0, 1), (0, -1, 1), (2, 0, 1), (2, -1, 1), (2, -1, 0), (2, 0, 0)):
    def ctx():
        c = rt.Context(0); c.set_option(rt.OPT_BVH_REINSERT, reinsert); c.set_option(rt.OPT_ANYHIT_ORDER, order); return c
    ref = ctx(); ref.upload(sc, W / H); ref.restir_reset(); ref.clear(W, H)
    for k, v in enumerate(cams):
        ref.set_camera(v, proj); ref.render_restir(rt.Params(frame_seed=70 + k, **base))
    ref_img = ref.read_accum()
    ranks = []
    for r in range(nshards):
        c = ctx(); c.set_option(rt.OPT_RESTIR_WAVEFRONT, wave); c.upload(sc, W / H); c.restir_reset(); c.clear(W, H); ranks.append(c)
    for k, v in enumerate(cams):
        slabs = []
        for r, c in enumerate(ranks):
            p = rt.Params(frame_seed=70 + k, shard_rank=r, shard_count=nshards, **base)
            c.set_camera(v, proj); c.render_restir(p)
            slab = torch.empty(c.restir_state_slab_bytes(p) // 4, dtype=torch.float32, device="cuda:0")
            c.restir_pack_state(p, slab.data_ptr()); slabs.append(slab)
        torch.cuda.synchronize()
        gathered = torch.cat(slabs); torch.cuda.synchronize()
        for r, c in enumerate(ranks):
            c.restir_unpack_state(rt.Params(frame_seed=70 + k, shard_rank=r, shard_count=nshards, **base), gathered.data_ptr())
    own = sharding.owner_map(W, H, TS, nshards, bool(blocks))
    bad = []
    for r, c in enumerate(ranks):
        img = c.read_accum()
        d = (bits(img) != bits(ref_img)).any(-1) & (own == r)
        bad += [(r, int(y), int(x)) for y, x in zip(*np.nonzero(d))]
        c.close()
    print(f"reinsert {reinsert} any_order {order} wavefront {wave}: refs {ref.stats().bvh_refs} nodes {ref.stats().bvh_nodes} differing pixels {bad[:8]} ({len(bad)})", flush=True)
    ref.close()
