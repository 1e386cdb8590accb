#!/usr/bin/env python3
"""Probe (round 4): what would two PIPELINE LANES buy the path-traced frame?  Two contexts on one GPU render the two halves of a frame's samples at the same time (two host
threads, each context on its own stream), against one context rendering all of them: an upper bound of what splitting rtx_render's batch into two concurrent halves could give
(one lane's HBM-bound k_shade beside the other's VALU-bound traversal, one lane's launch tails under the other's launches).
usage: python tools/two_lane_probe.py [sponza|bistro] [shard=N] [frames=6]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import __graft_entry__ as graft  # noqa: E402
rt = graft.load_package()
kind = next((a for a in sys.argv[1:] if a in ("sponza", "bistro")), "sponza")
named = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
shard, frames = named.get("shard", 1), named.get("frames", 6)
sc = rt.Scene.sponza_class() if kind == "sponza" else rt.Scene.bistro_class()
W, H, spp = 1920, 1080, 16
flags = 1 if kind == "sponza" else 4
kw = dict(shard_rank=0, shard_count=shard, tile_size=32) if shard > 1 else {}


def params(n, base):
    return rt.Params(width=W, height=H, spp=n, max_bounces=8, nee_samples=1, flags=flags, sample_base=base, **kw)


one = rt.Context(0); one.upload(sc, W / H)
two = [rt.Context(0), rt.Context(0)]
for c in two: c.upload(sc, W / H)


def run_one():
    one.clear(W, H); t = time.perf_counter(); one.render(params(spp, 0)); return (time.perf_counter() - t) * 1e3


def run_two():
    for c in two: c.clear(W, H)
    bar = threading.Barrier(3); done = []
    def work(i):
        bar.wait(); two[i].render(params(spp // 2, i * (spp // 2))); done.append(time.perf_counter())
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    return (max(done) - t0) * 1e3


for _ in range(2): run_one(); run_two()
a = [run_one() for _ in range(frames)]; b = [run_two() for _ in range(frames)]
print(f"{kind} shard 1/{shard}: one context, {spp} spp: {min(a):.2f} ms (median {sorted(a)[len(a)//2]:.2f});  two contexts at once, {spp//2} spp each: {min(b):.2f} ms (median {sorted(b)[len(b)//2]:.2f})")
