#!/usr/bin/env python3
"""Per-rank frame times when the image is sharded N ways, measured on ONE GPU by rendering every shard r of N one after the other (no gather): what the slowest rank
of an N-GPU frame costs (max over ranks decides the frame), how uneven the tile deal is (max / mean) and how large the slab of the frame's one all-gather is.
usage: python tools/shard_time.py [cornell|sponza|bistro|sponza4k] [pt|restir] [N ...] [blocks=0|1] [tile=64] [frames=4] [option_id=value ...]
   pt      BASELINE's path-traced frame of that scene (cornell: 1080p 64 spp 8 bounces; sponza / bistro: 1080p 16 spp; sponza4k: C4, 3840x2160 64 spp)
   restir  the reference's ReSTIR frame (nee 4, bounces 3), 1080p; blocks=1 (RTX_FLAG_BLOCK_TILES, one tile rectangle per rank) is the deal meant for it;
           halo=32 adds what the HALO exchange of the history moves (rtx_restir_halo_plan: border strips to <= 8 neighbours) beside the all-gather's slab, with a link-rate
           estimate of both (7 xGMI links of ~153 GB/s per GPU; NO N > 1 RCCL run exists: estimates, not measurements)
   native=1  (pt, cornell | sponza | bistro) every rank THROUGH THE HOST PATH of the native N-GPU frame: `rtx_render --gpus N --only-rank r --gather copy` = MultiGpuFrame with
             its persistent worker thread, the enqueue-only render (RTX_OPT_ASYNC), pack, the rank's own slab copied into the gathered buffer, unpack, ONE host wait per frame"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def native(kind, ns, frames):
    import re, subprocess
    exe = os.path.join(ROOT, "royaltracer-dx_amd", "rtx_render")
    spp = 64 if kind == "cornell" else 16
    print(f"# {kind} pt 1920x1080 through the native host path (MultiGpuFrame: worker thread, enqueue-only render, pack, own slab -> gathered, unpack, one wait); 32-px tiles for N > 1; ms per frame of each rank alone on one MI355X")
    print("| N | max over ranks ms | mean ms | max / mean | compute + host-path efficiency t1 / (N max) | per rank ms |\n|---|---|---|---|---|---|")
    t1 = None
    for n in ns:
        per = []
        for r in range(n):
            total = max(8 * frames, 24)                                  # a fresh process per rank: the first frames allocate and run on a GPU that is still clocking up
            cmd = [exe, "--scene", kind, "--w", "1920", "--h", "1080", "--spp", str(spp), "--bounces", "8", "--nee", "1", "--frames", str(total), "--gpus", str(n),
                   "--devices", ",".join(["0"] * n), "--only-rank", str(r), "--gather", "copy"] + (["--force-gather"] if n == 1 else [])
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            ms = [float(m) for m in re.findall(r"frame \d+ on \d+ GPUs: ([0-9.]+) ms", out.stdout)]
            if out.returncode != 0 or len(ms) != total:
                raise SystemExit(f"{cmd}: rc {out.returncode}\n{out.stderr[-2000:]}")
            per.append(float(np.mean(ms[total // 2:])))                  # the second half of the run
        mx, mean = max(per), float(np.mean(per))
        t1 = t1 or mx
        eff = f"{t1 / (n * mx):.3f}" if ns[0] == 1 else "-"
        print(f"| {n} | {mx:.3f} | {mean:.3f} | {mx / mean:.3f} | {eff} | " + " ".join(f"{t:.2f}" for t in per) + " |", flush=True)


def main():
    args = sys.argv[1:]
    kind = next((a for a in args if a in ("cornell", "sponza", "bistro", "sponza4k")), "cornell")
    mode = next((a for a in args if a in ("pt", "restir")), "pt")
    ns = [int(a) for a in args if a.isdigit()] or [1, 2, 4, 8]
    named = {a.split("=")[0]: int(a.split("=")[1]) for a in args if "=" in a and not a.split("=")[0].isdigit()}
    opts = [a.split("=") for a in args if "=" in a and a.split("=")[0].isdigit()]            # rtx option id=value
    blocks, tile, frames = named.get("blocks", 0), named.get("tile", 64), named.get("frames", 4)
    if named.get("native"):
        return native(kind, ns, frames)
    rt = graft.load_package()
    dev = torch.device("cuda", 0)
    scene = {"cornell": rt.Scene.cornell, "sponza": rt.Scene.sponza_class, "sponza4k": rt.Scene.sponza_class, "bistro": rt.Scene.bistro_class}[kind]()
    W, H = (3840, 2160) if kind == "sponza4k" else (1920, 1080)
    ctx = rt.Context(0)
    for k, v in opts:
        ctx.set_option(int(k), int(v))
    ctx.upload(scene, W / H)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
    flags = (rt.FLAG_BLOCK_TILES if blocks else 0)
    if mode == "restir":
        base = dict(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=flags, tile_size=tile)
    else:
        spp = 64 if kind in ("cornell", "sponza4k") else 16
        base = dict(width=W, height=H, spp=spp, sample_base=1, max_bounces=8, nee_samples=1, rr_start=3,
                    flags=flags | (1 if kind in ("cornell", "sponza", "sponza4k") else 4), tile_size=tile)
    render = ctx.render_restir if mode == "restir" else ctx.render
    t1 = None
    print(f"# {kind} {mode} {W}x{H} tile {tile} {'block' if blocks else 'round-robin'} deal; ms per frame of each rank's shard, rendered alone on one MI355X")
    print("| N | max over ranks ms | mean ms | max / mean | ideal t1 / N ms | compute-only efficiency t1 / (N max) | slab MB / rank | per rank ms |\n|---|---|---|---|---|---|---|---|")
    for n in ns:
        per = []
        for r in range(n):
            p = rt.Params(frame_seed=1, shard_rank=r, shard_count=n, **base)
            if mode == "restir":
                ctx.restir_reset()
            render(p)                                              # warm-up of this shard (allocations, halo list)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(frames):
                p.frame_seed = 2 + i
                render(p)
            torch.cuda.synchronize(dev)
            per.append((time.perf_counter() - t0) * 1e3 / frames)
        mx, mean = max(per), float(np.mean(per))
        t1 = t1 or mx
        slab = ctx.slab_bytes(rt.Params(shard_rank=0, shard_count=n, **base)) / 1e6
        if mode == "restir":
            slab += ctx.restir_state_slab_bytes(rt.Params(shard_rank=0, shard_count=n, **base)) / 1e6
        ideal, eff = (f"{t1 / n:.3f}", f"{t1 / (n * mx):.3f}") if ns[0] == 1 else ("-", "-")          # (needs the N = 1 frame of the same run)
        print(f"| {n} | {mx:.3f} | {mean:.3f} | {mx / mean:.3f} | {ideal} | {eff} | {slab:.2f} | " + " ".join(f"{t:.2f}" for t in per) + " |", flush=True)
        if mode == "restir" and n > 1:
            LINK = 153e9                                          # bytes / s per xGMI link, 7 links per GPU (MI355X_MICROARCH.md / SURVEY section 5)
            hist = ctx.restir_state_slab_bytes(rt.Params(shard_rank=0, shard_count=n, **base))
            ag_ms = hist * (n - 1) / min(n - 1, 7) / LINK * 1e3    # all-gather: every rank receives the other n - 1 slabs, over min(n - 1, 7) links in parallel
            line = f"|   | history exchange, ESTIMATED from the link rate | all-gather: {hist / 1e6:.1f} MB per rank slab, {hist * (n - 1) / 1e6:.0f} MB received per rank, ~{ag_ms:.3f} ms (+{100 * ag_ms / mx:.0f} % of the slowest rank's frame)"
            if named.get("halo") and blocks:
                worst = (0, 0, 0)
                for r in range(n):
                    peers, sb, rb = rt.restir_halo_plan(rt.Params(shard_rank=r, shard_count=n, **base), named["halo"])
                    worst = max(worst, (sb, len(peers), max([e.send_bytes for e in peers] or [0])))
                h_ms = worst[2] / LINK * 1e3 + 0.01               # every pair on its own link: the largest strip decides; + ~10 us of launch / protocol latency
                line += f" | halo {named['halo']} px: {worst[0] / 1e6:.2f} MB sent by the busiest rank to {worst[1]} peers, largest strip {worst[2] / 1e6:.2f} MB, ~{h_ms:.3f} ms (+{100 * h_ms / mx:.1f} %)"
            print(line + " |", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
