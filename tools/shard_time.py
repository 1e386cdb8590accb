#!/usr/bin/env python3
"""Per-rank frame time of the headline workload when the image is sharded N ways (one GPU renders shard 0 of N, no gather):
what strong scaling can reach at best, and how much of a small shard's frame is fixed per-launch cost.
usage: python tools/shard_time.py [N ...] [option_id=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:] if "=" not in a] or [1, 2, 4, 8]
    opts = [a.split("=") for a in sys.argv[1:] if "=" in a]            # rtx option id=value
    rt = graft.load_package()
    dev = torch.device("cuda", 0)
    scene = rt.Scene.cornell()
    W, H = 1920, 1080
    ctx = rt.Context(0)
    for k, v in opts:
        ctx.set_option(int(k), int(v))
    ctx.upload(scene, W / H)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
    t1 = None
    for n in ns:
        p = rt.Params(width=W, height=H, spp=64, sample_base=1, max_bounces=8, nee_samples=1, rr_start=3, frame_seed=1, flags=1,
                      tile_size=64, shard_rank=0, shard_count=n)
        ctx.render(p)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(5):
            p.frame_seed = 2 + i
            ctx.render(p)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / 5
        t1 = t1 or ms
        print(f"shard 0 of {n}: {ms:7.3f} ms/frame   ideal {t1 / n:7.3f}   compute-only efficiency {t1 / (n * ms):.3f}")
    ctx.close()


if __name__ == "__main__":
    main()
