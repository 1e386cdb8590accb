#!/usr/bin/env python3
"""bench.py — Mrays/s + ms/frame of the wavefront path tracer on BASELINE.json's headline workload:
Cornell Box, 1920x1080, 64 spp, 8 bounces, diffuse + emissive (configs[1]), on N MI355X.

A "step" is one whole frame (all spp) of the hot path over synthetic, in-repo generated input that is
already resident in HBM when the timed region starts.  For N > 1 the driver launches one process per GPU
(torch.distributed, backend nccl == RCCL); 64x64 pixel tiles are dealt round-robin to the ranks (no
data-path collective inside the frame) and the frame ends with ONE all_gather of the tile slabs over xGMI.
`value` = rays traced by all ranks / max-over-ranks wall time.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# algorithmic bytes per work item of each kernel (DESIGN.md "Bytes model"; SURVEY.md §8d:
# 224 B per extension ray = 48 (trace) + 176 (shade); 96 B per shadow ray = 48 written by shade + 48 read
# by trace_shadow; 32 B per pixel-sample accumulation)
ALG_BYTES = {"trace_closest": 48, "shade": 176, "trace_shadow": 48, "accumulate": 32, "raygen": 64,
             "bounce_fused": 224}   # fused trace+shade+shadow kernel: 224 B per extension ray + 96 B per shadow ray it traces
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
TRAFFIC_PROFILE = "r01_cornell_c2_latest.json"   # written by tools/rocprof_summary.py from separate --pmc passes

WORKLOADS = {
    # name: (scene ctor, width, height, spp, bounces, nee, flags)
    "cornell_1080p_64spp_8b": ("cornell", 1920, 1080, 64, 8, 1, 1),
    "cornell_1080p_1spp_4b": ("cornell", 1920, 1080, 1, 4, 1, 1),
    "sponza_1080p_16spp_8b": ("sponza", 1920, 1080, 16, 8, 1, 1),
    "sponza_4k_64spp_8b": ("sponza", 3840, 2160, 64, 8, 1, 1),
    "bistro_1080p_16spp_8b": ("bistro", 1920, 1080, 16, 8, 1, 0),
}


def make_scene(rt, kind):
    if kind == "cornell":
        return rt.Scene.cornell()
    if kind == "sponza":
        return rt.Scene.sponza_class()
    if kind == "bistro":
        return rt.Scene.bistro_class()
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_64spp_8b", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--paths-per-batch", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: host-staged gather, for testing the N>1 flow on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="force this CUDA device on every rank (testing only)")
    ap.add_argument("--checksum", action="store_true", help="add a checksum of the final accumulation buffer to the JSON line")
    ap.add_argument("--opt", action="append", default=[], help="tuning: rtx option id=value (repeatable)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dev_index = args.device if args.device >= 0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rt = graft.load_package()
    from royaltracer_dx_amd import sharding
    dist = None
    if world > 1:
        dist, rank, world = sharding.init_process_group(args.dist_backend, dev)     # backend "nccl" is RCCL on ROCm
    kind, W, H, spp, bounces, nee, flags = WORKLOADS[args.workload]
    scene = make_scene(rt, kind)
    ctx = rt.Context(dev_index)
    for kv in args.opt:                      # tuning knobs must be set before the scene is committed
        k, v = kv.split("=")
        ctx.set_option(int(k), int(v))
    ctx.upload(scene, W / H)
    if args.paths_per_batch:
        ctx.set_option(rt.OPT_PATHS_PER_BATCH, args.paths_per_batch)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
    params = rt.Params(width=W, height=H, spp=spp, sample_base=1, max_bounces=bounces, nee_samples=nee, rr_start=3,
                       frame_seed=1, flags=flags, tile_size=64, shard_rank=rank, shard_count=world)
    slab = gathered = None
    if world > 1:
        nfl = ctx.slab_bytes(params) // 4
        slab = torch.empty(nfl, dtype=torch.float32, device=dev)
        gathered = torch.empty(nfl * world, dtype=torch.float32, device=dev)

    def frame(step):
        accum.zero_()
        params.frame_seed = 1 + step
        ctx.render(params)                                   # synchronous on the bound stream
        if world > 1:                                        # final framebuffer gather over xGMI (RCCL)
            ctx.pack_tiles(params, slab.data_ptr())
            if args.dist_backend == "gloo":                  # testing path: stage through the host
                g = sharding.gather_slabs(dist, slab.cpu())
                gathered.copy_(g)
            else:
                sharding.gather_slabs(dist, slab, gathered)
            ctx.unpack_tiles(params, gathered.data_ptr())
        return ctx.stats()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        frame(i)
    ctx.set_option(rt.OPT_KERNEL_TIMING, 0 if args.no_kernel_timing else 1)
    kms = np.zeros(rt.K_COUNT); kitems = np.zeros(rt.K_COUNT); klaunch = np.zeros(rt.K_COUNT)
    rays = np.zeros(3); phits = 0.0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = frame(args.warmup + i)
        kms += np.array(st.kernel_ms[:]); kitems += np.array(st.kernel_items[:], dtype=np.float64); klaunch += np.array(st.kernel_launches[:], dtype=np.float64)
        rays += np.array([st.rays_primary, st.rays_extension, st.rays_shadow], dtype=np.float64); phits += st.primary_hits
    barrier()
    dt = time.perf_counter() - t0
    dt_max, rays_all = dt, rays
    if world > 1:
        cdev = dev if args.dist_backend == "nccl" else None
        dt_max = sharding.max_over_ranks(dist, dt, cdev)
        rays_all = sharding.sum_over_ranks(dist, rays, cdev)
        # (phits stays rank-local: it only prices rank 0's own kernel launches below)
    ms_per_step = dt_max * 1e3 / max(args.steps, 1)
    value = float(rays_all.sum()) / dt_max / 1e6 if dt_max > 0 else 0.0

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP-event time per class, measured in the timed region) ----
        roof = None
        if not args.no_kernel_timing and kms.sum() > 0:
            k = int(np.argmax(kms))
            name = rt.KERNEL_NAMES[k]
            bytes_per_launch = ALG_BYTES[name] * kitems[k] / max(klaunch[k], 1)
            if name == "bounce_fused":      # bounce 0 items were traced by raygen (their 48 B belong to it); + 96 B per shadow ray
                bytes_per_launch += (96.0 * rays[2] - 48.0 * phits) / max(klaunch[k], 1)
            avg_ms = kms[k] / max(klaunch[k], 1)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic = None
            try:    # HBM bytes per launch from the rocprofv3 PMC passes of this same command (tools/rocprof_summary.py)
                prof = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
                if args.workload == "cornell_1080p_64spp_8b" and world == 1:
                    sym = {"bounce_fused": "k_bounce_small"}.get(name, "k_" + name)
                    rows = [v for k, v in prof["kernels"].items() if k.split("<")[0] == sym]      # template instantiations of one kernel
                    traffic = round(sum(v["hbm_bytes_per_launch"] * v["calls"] for v in rows) / sum(v["calls"] for v in rows))
            except Exception:
                traffic = None
            roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "alg_bytes_per_launch": round(bytes_per_launch),
                    "alg_bytes_per_item": ALG_BYTES[name], "items_per_launch": round(kitems[k] / max(klaunch[k], 1), 1),
                    "avg_launch_ms": round(avg_ms, 5), "launches": int(klaunch[k]),
                    "kernel_ms_by_class": {rt.KERNEL_NAMES[i]: round(float(kms[i]), 3) for i in rt.KERNEL_NAMES if klaunch[i] > 0}}
        # ---- CPU baseline: the oracle (a port of the reference's shader math) on this host's cores ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            orc = graft.load_oracle()
            o = orc.Oracle().load(scene, W / H)
            cores = os.cpu_count() or 1
            o.set_threads(cores)
            cw, ch = W, H
            cp = params.copy(spp=1, shard_rank=0, shard_count=1, frame_seed=1)
            # bounded sample: 1 spp of the same frame; shrink the image if this host is slow
            probe = cp.copy(width=W // 8, height=H // 8)
            o.set_camera(*scene.view_proj(W / H))
            tp = time.perf_counter(); _, pc = o.render(probe); tp = time.perf_counter() - tp
            est = tp * 64
            scale = 1
            while est / (scale * scale) > 30.0 and scale < 8:
                scale *= 2
            cp = cp.copy(width=W // scale, height=H // scale, spp=1)
            tc = time.perf_counter(); _, cc = o.render(cp); tc = time.perf_counter() - tc
            cpu_spp = 1
            if tc < 6.0:             # fast host: take more of the frame's samples, aiming at ~12 s of wall time on all cores
                cpu_spp = int(max(1, min(spp, 32, round(12.0 / max(tc, 1e-3)))))
                if cpu_spp > 1:
                    cp = cp.copy(spp=cpu_spp)
                    tc = time.perf_counter(); _, cc = o.render(cp); tc = time.perf_counter() - tc
            cpu = {"value": round(sum(cc) / tc / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                   "sample": f"oracle/rt_oracle.c (OpenMP), {cp.width}x{cp.height}, {cpu_spp} of {spp} spp of the same frame, {sum(cc)} rays in {tc:.2f} s"}
        out = {"metric": "Mrays/s + ms/frame at 1080p, 8-bounce Cornell Box" if kind == "cornell" else f"Mrays/s + ms/frame, {args.workload}",
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": args.workload, "scene": scene_name(kind), "width": W, "height": H, "spp": spp,
                          "max_bounces": bounces, "nee_samples": nee, "rr_start": 3, "flags": flags,
                          "triangles": int(scene.num_triangles), "parallelism": f"pixel-tiles/{world}", "tile_size": 64,
                          "rays_per_frame": {"primary": int(rays_all[0] / max(args.steps, 1)), "extension": int(rays_all[1] / max(args.steps, 1)),
                                             "shadow": int(rays_all[2] / max(args.steps, 1))}},
               "roofline": roof, "cpu_baseline": cpu}
        if args.checksum:
            import hashlib
            out["accum_sha1"] = hashlib.sha1(accum.cpu().numpy().tobytes()).hexdigest()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def scene_name(kind):
    return {"cornell": "Cornell Box (32 triangles, 2 emissive)", "sponza": "Sponza-class procedural atrium",
            "bistro": "Bistro-class procedural street"}[kind]


if __name__ == "__main__":
    main()
