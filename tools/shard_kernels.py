#!/usr/bin/env python3
"""Where a small shard's frame time goes: wall ms per frame of shard r of N (launches back to back) beside the sum of its kernels (RTX_OPT_KERNEL_TIMING: launches serialised,
one event pair each) -- the difference is launch gaps + what overlapping launches hide.
usage: python tools/shard_kernels.py [cornell|sponza] [pt|restir] N [rank=0] [tile=32] [blocks=0|1] [frames=8] [option_id=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

args = sys.argv[1:]
kind = next((a for a in args if a in ("cornell", "sponza")), "cornell")
mode = next((a for a in args if a in ("pt", "restir")), "pt")
n = next((int(a) for a in args if a.isdigit()), 8)
named = {a.split("=")[0]: int(a.split("=")[1]) for a in args if "=" in a and not a.split("=")[0].isdigit()}
opts = [a.split("=") for a in args if "=" in a and a.split("=")[0].isdigit()]
rank, tile, blocks, frames = named.get("rank", 0), named.get("tile", 32), named.get("blocks", 0), named.get("frames", 8)
rt = graft.load_package()
dev = torch.device("cuda", 0)
scene = {"cornell": rt.Scene.cornell, "sponza": rt.Scene.sponza_class}[kind]()
W, H = 1920, 1080
ctx = rt.Context(0)
for k, v in opts:
    ctx.set_option(int(k), int(v))
ctx.upload(scene, W / H)
accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
flags = rt.FLAG_BLOCK_TILES if blocks else 0
if mode == "restir":
    p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=flags, tile_size=tile, shard_rank=rank, shard_count=n, frame_seed=1)
else:
    p = rt.Params(width=W, height=H, spp=64 if kind == "cornell" else 16, sample_base=1, max_bounces=8, nee_samples=1, rr_start=3, flags=flags | 1, tile_size=tile,
                  shard_rank=rank, shard_count=n, frame_seed=1)
render = ctx.render_restir if mode == "restir" else ctx.render


def run(k):
    if mode == "restir":
        ctx.restir_reset()
    render(p); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(k):
        p.frame_seed = 2 + i
        render(p)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3 / k


wall = min(run(frames) for _ in range(3))
ctx.set_option(rt.OPT_KERNEL_TIMING, 1)
run(1)
p.frame_seed = 2
render(p); torch.cuda.synchronize(dev)
st = ctx.stats()
ks = {rt.KERNEL_NAMES[i]: (st.kernel_ms[i], st.kernel_launches[i]) for i in rt.KERNEL_NAMES if st.kernel_launches[i] > 0}
tot, nl = sum(v[0] for v in ks.values()), sum(v[1] for v in ks.values())
print(f"{kind} {mode} shard {rank} of {n}, tile {tile}{' block deal' if blocks else ''}: wall {wall:.3f} ms per frame; kernels {tot:.3f} ms in {nl} launches "
      f"({(wall - tot) / max(nl, 1) * 1e3:.1f} us per launch not in kernels)")
print("  " + ", ".join(f"{k} {v[0]:.3f} ms / {v[1]}" for k, v in ks.items()))
ctx.close()
