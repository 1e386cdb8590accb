#!/usr/bin/env python3
"""bench.py — Mrays/s + ms/frame of the wavefront path tracer on BASELINE.json's headline workload:
Cornell Box, 1920x1080, 64 spp, 8 bounces, diffuse + emissive (configs[1]), on N MI355X.

A "step" is one whole frame (all spp) of the hot path over synthetic, in-repo generated input that is
already resident in HBM when the timed region starts.  For N > 1 there is one process per GPU
(torch.distributed, backend nccl == RCCL); 64x64 pixel tiles are dealt round-robin to the ranks (no
data-path collective inside the frame) and the frame ends with ONE all_gather of the tile slabs over xGMI.
`value` = rays traced by all ranks / max-over-ranks wall time.  Rank 0 prints one JSON line.

Launching: either under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, which sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or plainly as `python bench.py --gpus N`: then this process — before it imports
torch or touches the GPU — starts N children of itself with those variables set, relays rank 0's JSON line and exits with the
worst child exit code.  Nothing is ever exec'ed from a process that has initialised the GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402   (does not import torch)

# ALGORITHMIC bytes: 224 B per extension ray (ray 32 W + 32 R, path state 64 R + 64 W, hit 16 W + 16 R),
# 96 B per shadow ray (48 B entry W + R), 32 B per pixel-sample (accumulation R + W).  The separate kernels of the general path split
# them: trace_closest 48 (ray R + hit W) and shade 176 per extension ray; trace_shadow 48 of the 96 per shadow ray, the other 48 are
# shade's write.  Primary rays are priced like extension rays (their ray + state are written by raygen: 64 B of the 224).
B_EXT, B_SHADOW, B_PIX = 224.0, 96.0, 32.0
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# COMPUTE roofline (the fused tiny-scene kernels and the BVH traversal are VALU-bound, not HBM-bound): VALU instructions per SIMD-cycle of the kernel divided by what a
# SATURATED REPLAY LOOP of the kernel's own instruction-class mix reaches on this chip (tools/gen_mix.py + tools/valu_peak.hip, 8 waves / SIMD) — a fraction that cannot
# exceed 1 by construction.  (Round 2 reported 4 * SQ_ACTIVE_INST_VALU / SIMD-cycles, which read 1.05-1.10: that counter charges every VALU instruction 4 cycles, while
# v_fma_f32 / VOP2 integer ops take 2.3 / 3.0; and per-opcode costs do not add up either — profiles/r03_valu_calib.md — so the peak of a mix has to be measured.)
# The counters come from tracked profiles/r05_roof_<workload>.json (tools/roofline_run.sh: separate rocprofv3 --pmc passes of this command); each file carries the hash
# of the kernel sources it was recorded with, and a profile that no longer matches the loaded kernels is reported as stale instead of being priced.
ROOF_PROFILE = {"cornell_1080p_64spp_8b": "r05_roof_cornell.json", "sponza_1080p_16spp_8b": "r05_roof_sponza.json", "bistro_1080p_16spp_8b": "r05_roof_bistro.json"}
KERNEL_SYMBOL = {"bounce_fused": ("k_bounce_small", "k_bounce_bvh"), "trace_closest": ("k_trace_closest",), "shade": ("k_shade",),
                 "trace_shadow": ("k_trace_shadow",), "accumulate": ("k_accumulate",), "raygen": ("k_raygen", "k_raygen_trace_small")}

WORKLOADS = {
    # name: (scene ctor, width, height, spp, bounces, nee, flags)
    "cornell_1080p_64spp_8b": ("cornell", 1920, 1080, 64, 8, 1, 1),
    "cornell_1080p_1spp_4b": ("cornell", 1920, 1080, 1, 4, 1, 1),
    "sponza_1080p_16spp_8b": ("sponza", 1920, 1080, 16, 8, 1, 1),
    "sponza_4k_64spp_8b": ("sponza", 3840, 2160, 64, 8, 1, 1),
    "bistro_1080p_16spp_8b": ("bistro", 1920, 1080, 16, 8, 1, 4),     # flags 4 = RTX_FLAG_TRANSMISSION: GGX microfacet + dielectric panes (strategy 3) + NEE
    # the HARD stand-ins (round 5, host/Scenes.h): the same shells, materials, lights, cameras and triangle budgets with the size distribution of the real assets — a few
    # triangles metres long beside millimetre ornament, long thin trims, overlapping cloth, foliage (> 1000 : 1) — where a BVH builder's quality shows
    "sponza_hard_1080p_16spp_8b": ("sponza_hard", 1920, 1080, 16, 8, 1, 1),
    "bistro_hard_1080p_16spp_8b": ("bistro_hard", 1920, 1080, 16, 8, 1, 4),
}
EXTRA_WORKLOADS = ("sponza_1080p_16spp_8b", "bistro_1080p_16spp_8b", "sponza_hard_1080p_16spp_8b", "bistro_hard_1080p_16spp_8b")    # the general BVH path, timed beside the headline (GPU only)


ASSETS = {"sponza": "sponza.obj", "bistro": "bistro.obj", "sponza_hard": "sponza.obj", "bistro_hard": "bistro.obj"}       # SURVEY 8(d) / BASELINE.md: the real asset if it lies under assets/, else the procedural stand-in of its class


def asset_path(kind):
    f = ASSETS.get(kind)
    for d in (os.environ.get("RTX_ASSETS", ""), os.path.join(ROOT, "assets")):
        if f and d and os.path.isfile(os.path.join(d, f)):
            return os.path.join(d, f)
    return None


def make_scene(rt, kind):
    """-> (scene, source description).  Sponza / Bistro: assets/<name>.obj through the OBJ / MTL reader (ObjLoader::loadObjFile, as the reference loads its models:
    Renderer.cpp:363-370), looked at from the asset's own bounding box; without the file, the deterministic procedural scene of that class."""
    if kind == "cornell":
        return rt.Scene.cornell(), "generated: Cornell Box (32 triangles, 2 emissive)"
    a = asset_path(kind)
    if a:
        sc = rt.Scene.from_obj([a], os.path.dirname(a) + "/")
        sc.frame_bounds()                                     # camera inside / in front of the model's bounding box (an OBJ file carries no camera)
        return sc, f"asset: {os.path.relpath(a, ROOT)} ({sc.num_triangles} triangles)"
    if kind == "sponza":
        return rt.Scene.sponza_class(), "generated: Sponza-class procedural atrium (no assets/sponza.obj)"
    if kind == "bistro":
        return rt.Scene.bistro_class(), "generated: Bistro-class procedural street (no assets/bistro.obj)"
    if kind == "sponza_hard":
        return rt.Scene.sponza_class(hard=True), "generated: Sponza-class procedural atrium, HARD variant: the real asset's triangle-size distribution (no assets/sponza.obj)"
    if kind == "bistro_hard":
        return rt.Scene.bistro_class(hard=True), "generated: Bistro-class procedural street, HARD variant: the real asset's triangle-size distribution (no assets/bistro.obj)"
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------------------------------
# plain `python bench.py --gpus N`: start the ranks (this process never touches the GPU)
# ------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line); sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True); t.start()
    worst = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                worst = worst or rc
                for q in alive:               # one rank failed: the others would wait in the collective forever
                    procs[q].terminate()      # (exactly the children started above, by handle)
        time.sleep(0.05)
    t.join(timeout=5)
    return worst


def step_stats(ms):
    """min / median / max / mean of per-step times: a 20-step MEAN cannot tell a slow box from one disturbed step"""
    a = np.asarray(ms, dtype=np.float64)
    if a.size == 0:
        return None
    return {"min": round(float(a.min()), 3), "median": round(float(np.median(a)), 3), "max": round(float(a.max()), 3), "mean": round(float(a.mean()), 3), "n": int(a.size)}


def gpu_env(dev_index=0):
    """What explains box-to-box variance, read ONCE before the timed region (never inside it): shader / memory clocks in force, power cap and draw, temperatures, driver
    and ROCm versions — from sysfs where an ordinary user may read it, else from rocm-smi.  Every field is best-effort; a missing one is simply absent."""
    import glob, re
    env = {}

    def rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    try:
        env["rocm"] = rd("/opt/rocm/.info/version")
        env["amdgpu_driver"] = rd("/sys/module/amdgpu/version") or rd("/sys/module/amdgpu/srcversion")
        env["kernel"] = os.uname().release
        # the dev_index-th render-capable AMD card in sysfs order
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if (rd(os.path.join(d, "vendor")) or "") == "0x1002" and os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        if cards:
            d = cards[min(dev_index, len(cards) - 1)]
            for key, f in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
                t = rd(os.path.join(d, f))
                if t:
                    lv = [ln.strip() for ln in t.splitlines()]
                    cur = [ln for ln in lv if ln.endswith("*")]
                    env[key] = {"current": cur[0].rstrip("* ").split(":")[-1].strip() if cur else None, "levels": [ln.rstrip("* ").split(":")[-1].strip() for ln in lv][:8]}
            env["perf_level"] = rd(os.path.join(d, "power_dpm_force_performance_level"))
            for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for key, f, scale in (("power_cap_w", "power1_cap", 1e-6), ("power_avg_w", "power1_average", 1e-6), ("power_input_w", "power1_input", 1e-6),
                                      ("temp_edge_c", "temp1_input", 1e-3), ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                    v = rd(os.path.join(hw, f))
                    if v and re.fullmatch(r"-?\d+", v):
                        env[key] = round(int(v) * scale, 1)
        if "sclk" not in env:                                  # sysfs not readable: ask the tool (a child process; nothing is exec'ed in place)
            for cmd in (["rocm-smi", "-d", str(dev_index), "--showclocks", "--showtemp", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                        ["amd-smi", "metric", "-g", str(dev_index), "--clock", "--power", "--temperature", "--json"]):
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                    if r.returncode == 0 and r.stdout.strip().startswith(("{", "[")):
                        env[cmd[0].replace("-", "_")] = json.loads(r.stdout)
                        break
                except Exception:
                    continue
    except Exception as e:                                      # the bench line must survive whatever a box forbids
        env["error"] = str(e)[:120]
    return {k: v for k, v in env.items() if v is not None}


def kernel_source_hash():
    """hash of the kernel sources (tools/valu_calib.py writes the same into a profile): tells whether a tracked counter profile still describes the loaded kernels"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "royaltracer-dx_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(workload):
    """-> (kernel rows, source string, stale flag) from the tracked counter profile of this workload, or (None, None, None)"""
    f = ROOF_PROFILE.get(workload)
    if not f:
        return None, None, None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", f)))
    except Exception:
        return None, None, None
    return d["kernels"], "profiles/" + f, d.get("kernel_source_sha") != kernel_source_hash()


def pmc_for(rows, cls):
    """sum the template instantiations of one kernel class"""
    if not rows:
        return None
    sel = [r for r in rows if r["kernel"].split("<")[0] in KERNEL_SYMBOL.get(cls, ())]
    if not sel:
        return None
    launches = sum(r["launches"] for r in sel)
    valu = sum(r["valu_inst"] for r in sel)
    lanes = sum((r.get("lanes_per_valu") or 0.0) * r["valu_inst"] for r in sel) / valu if valu else None
    # compute fraction of the class = instruction-weighted mean over the instantiations that have a replay peak
    wf = [(r["compute_frac"], r["valu_inst"], r["inst_per_simd_cycle"], r["compute_peak_inst_per_simd_cycle"]) for r in sel if r.get("compute_frac") is not None]
    wsum = sum(w for _, w, _, _ in wf)
    return {"launches": launches, "valu_inst_per_launch": valu / launches, "lanes_per_valu": lanes,
            "compute_frac": sum(f * w for f, w, _, _ in wf) / wsum if wsum else None,
            "inst_per_simd_cycle": sum(a * w for _, w, a, _ in wf) / wsum if wsum else None,
            "peak_inst_per_simd_cycle": sum(pk * w for _, w, _, pk in wf) / wsum if wsum else None,
            "hbm_bytes_per_launch": sum(r["hbm_bytes"] for r in sel) / launches}


def roofline_record(rt, workload, kms, kitems, klaunch, rays, n_pixel_samples, steps):
    """roofline of the kernel class with the largest HIP-event time inside the timed region (events are recorded on the context's stream)"""
    if kms.sum() <= 0:
        return None
    k = int(np.argmax(kms))
    name = rt.KERNEL_NAMES[k]
    launches = max(klaunch[k], 1)
    n_prim, n_ext, n_sh = rays
    # algorithmic bytes of all launches of this class over the timed region, per SURVEY 8(d)
    if name == "bounce_fused":       # fused trace + shade + shadow + sample: every extension ray and every shadow ray of the frame
        alg = B_EXT * n_ext + B_SHADOW * n_sh
    elif name == "trace_closest":
        alg = 48.0 * (n_prim + n_ext)
    elif name == "shade":
        alg = 176.0 * (n_prim + n_ext) + 48.0 * n_sh
    elif name == "trace_shadow":
        alg = 48.0 * n_sh
    elif name == "raygen":
        alg = 64.0 * n_prim
    else:
        alg = B_PIX * n_pixel_samples
    avg_ms = kms[k] / launches
    bytes_per_launch = alg / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    rows, src, stale = load_pmc(workload)
    pm = pmc_for(rows, name) if not stale else None
    traffic = round(pm["hbm_bytes_per_launch"]) if pm else None
    roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "traffic_source": ((src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, 2*FETCH+WRITE KiB; not measured in this run)") if pm else
                               (src + ": STALE (recorded with other kernel sources) - not used" if stale else None)),
            "alg_bytes_per_launch": round(bytes_per_launch), "alg_bytes_model": "224*N_ext + 96*N_shadow (+ 32*N_px*spp for the frame)",
            "avg_launch_ms": round(avg_ms, 5), "launches": int(klaunch[k]), "frames": int(max(steps, 1)),
            "kernel_ms_by_class": {rt.KERNEL_NAMES[i]: round(float(kms[i]), 3) for i in rt.KERNEL_NAMES if klaunch[i] > 0}}
    # whole frame, all kernels, SURVEY 8(d): bytes_alg = 224*N_ext + 96*N_shadow + 32*N_px*spp (primary rays are not priced: their state is part of the first extension ray's 224 B)
    frame_ms = float(kms.sum()) / max(steps, 1)
    frame_alg = (B_EXT * n_ext + B_SHADOW * n_sh + B_PIX * n_pixel_samples) / max(steps, 1)
    roof["frame"] = {"alg_bytes": round(frame_alg), "kernel_ms": round(frame_ms, 4),
                     "frac": round(frame_alg / (frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if frame_ms > 0 else None}
    if pm and traffic:
        roof["traffic_frac"] = round(pm["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)   # measured HBM bytes / live launch time
    if pm and pm["compute_frac"] is not None:
        roof["compute"] = {"frac": round(pm["compute_frac"], 4), "unit": "VALU instructions per SIMD-cycle",
                           "achieved": round(pm["inst_per_simd_cycle"], 4), "peak": round(pm["peak_inst_per_simd_cycle"], 4),
                           "lanes_per_inst": round(pm["lanes_per_valu"], 2) if pm["lanes_per_valu"] else None,
                           "frac_x_lanes": round(pm["compute_frac"] * (pm["lanes_per_valu"] or 64.0) / 64.0, 4),
                           "valu_inst_per_launch": round(pm["valu_inst_per_launch"]),
                           "model": "peak = what a saturated replay loop of this kernel's own instruction-class mix issues on this chip (tools/gen_mix.py, tools/valu_peak.hip); "
                                    "achieved = SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) of the kernel; frac_x_lanes = frac x active lanes / 64",
                           "source": src + " (counters of separate rocprofv3 --pmc runs of this command, same kernel sources as loaded)"}
        if pm["compute_frac"] > max(roof["frac"], roof.get("traffic_frac", 0.0)):
            roof["bound"] = "valu"
    return roof


def trace_work(rt, ctx, params, roof):
    """WORK per ray of the BVH traversal, the number a tree-quality or visiting-order change moves: one more frame with RTX_OPT_TRACE_COUNTERS — the generic
    instantiations of the persistent kernels tally every lane's node steps and triangle tests — divided by the rays of that frame; and, where the tracked counter profile of
    this workload is current, the VALU lane-instructions per ray of the dominant kernel class (SQ_INSTS_VALU x active lanes / rays)."""
    ctx.set_option(rt.OPT_KERNEL_TIMING, 0); ctx.set_option(rt.OPT_TRACE_COUNTERS, 1)
    try:
        ctx.trace_counters()                                   # reset
        ctx.render(params)
        st = ctx.stats(); cn, ct, an, at = ctx.trace_counters()
    finally:
        ctx.set_option(rt.OPT_TRACE_COUNTERS, 0)
    n_closest, n_any = st.rays_primary + st.rays_extension, st.rays_shadow
    out = {"closest_hit": {"rays": int(n_closest), "node_steps_per_ray": round(cn / max(n_closest, 1), 3), "tri_tests_per_ray": round(ct / max(n_closest, 1), 3)},
           "any_hit": {"rays": int(n_any), "node_steps_per_ray": round(an / max(n_any, 1), 3), "tri_tests_per_ray": round(at / max(n_any, 1), 3)},
           "source": "RTX_OPT_TRACE_COUNTERS on one more frame (generic instantiation of k_trace_closest / k_trace_shadow, speculative schedule as timed)"}
    comp = (roof or {}).get("compute")
    if comp and comp.get("valu_inst_per_launch") and comp.get("lanes_per_inst") and roof.get("launches"):
        n_dom = {"trace_closest": n_closest, "trace_shadow": n_any}.get(roof["kernel"])
        if n_dom:       # the profile's launches are those of the timed frames: per frame = per launch x launches per frame
            per_frame_launches = roof["launches"] / max(roof.get("frames", 1), 1)
            out["valu_lane_inst_per_ray"] = round(comp["valu_inst_per_launch"] * per_frame_launches * comp["lanes_per_inst"] / n_dom, 1)
            out["valu_lane_inst_source"] = comp["source"]
    return out


def time_extra(rt, dev_index, workload, steps=5):
    """one BVH workload, GPU only: warm-up + `steps` timed frames on a context of its own -> small record for the JSON line"""
    import torch
    kind, W, H, spp, bounces, nee, flags = WORKLOADS[workload]
    scene, source = make_scene(rt, kind)
    ctx = rt.Context(dev_index)
    try:
        t_up = time.perf_counter()
        ctx.upload(scene, W / H)
        commit_s = time.perf_counter() - t_up                   # scene hand-over + rtx_commit_scene (BVH build) + upload
        ctx.clear(W, H)
        params = rt.Params(width=W, height=H, spp=spp, sample_base=1, max_bounces=bounces, nee_samples=nee, rr_start=3, frame_seed=1, flags=flags)
        ctx.render(params)                                        # warm-up (allocations)
        # frame time first, as a caller gets it (kernel timing off: the shadow-ray kernel of a bounce overlaps the closest-hit kernel of the next) ...
        torch.cuda.synchronize()
        t0 = time.perf_counter(); per_step = []
        for i in range(steps):
            params.frame_seed = 2 + i
            ts = time.perf_counter(); ctx.render(params); per_step.append((time.perf_counter() - ts) * 1e3)       # rtx_render returns after its own stream synchronise
        dt = time.perf_counter() - t0
        # ... then the same frames once more with per-kernel HIP events (no overlap: a launch needs the GPU to itself to have a duration)
        ctx.set_option(rt.OPT_KERNEL_TIMING, 1)
        kms = np.zeros(rt.K_COUNT); kl = np.zeros(rt.K_COUNT); rays = np.zeros(3)
        t1 = time.perf_counter()
        for i in range(steps):
            params.frame_seed = 2 + i
            ctx.render(params)
            st = ctx.stats()
            kms += np.array(st.kernel_ms[:]); kl += np.array(st.kernel_launches[:], dtype=np.float64)
            rays += np.array([st.rays_primary, st.rays_extension, st.rays_shadow], dtype=np.float64)
        dt_timed = time.perf_counter() - t1
        roof = roofline_record(rt, workload, kms, None, kl, rays, float(W) * H * spp * steps, steps)
        work = trace_work(rt, ctx, params, roof)
        # what a transform-only commit (instance 0 moved: GPU refit of the resident tree) costs, wall time; the first one uploads the object-space triangles
        refit = []
        for k in range(4):
            m = np.eye(4, dtype=np.float32); m[3, 0] = 0.001 * (k + 1)
            ctx.set_instance_transform(0, m.reshape(16))
            t2 = time.perf_counter(); ctx.commit(); refit.append((time.perf_counter() - t2) * 1e3)
        refit_ms = round(min(refit[1:]), 3)
        # the same scene committed with the tree built ON THE GPU (RTX_OPT_GPU_BUILD): commit wall time, phases, and the frame on that tree
        gpu_build = None
        try:
            cg = rt.Context(dev_index); cg.set_option(rt.OPT_GPU_BUILD, 1)
            t3 = time.perf_counter(); cg.upload(scene, W / H); commit_g = time.perf_counter() - t3
            bi = cg.build_info()
            cg.clear(W, H); cg.render(params)
            gms = []
            for i in range(3):
                params.frame_seed = 2 + i
                ts = time.perf_counter(); cg.render(params); gms.append((time.perf_counter() - ts) * 1e3)
            gpu_build = {"commit_s": round(commit_g, 3), "build_phases_ms": {k: round(v, 2) for k, v in zip(("boxes_keys", "sort", "ploc", "top_on_host", "layout"), bi["ms"])},
                         "ploc_rounds": bi["ploc_rounds"], "clusters_to_host": bi["clusters_top"], "wide_nodes": bi["nodes"], "ms_per_frame_stats": step_stats(gms),
                         "frame_vs_host_tree": round(min(gms) / min(per_step), 4)}
            cg.close()
        except Exception as e:
            gpu_build = {"error": str(e)[:200]}
        rec = {"ms_per_frame": round(dt * 1e3 / steps, 3), "ms_per_frame_stats": step_stats(per_step), "Mrays_s": round(float(rays.sum()) / dt / 1e6, 1), "triangles": int(scene.num_triangles), "scene": source,
               "rays_per_frame": int(rays.sum() / steps), "dominant_kernel": roof["kernel"] if roof else None,
               "frac": roof["frac"] if roof else None, "bound": roof["bound"] if roof else None,
               "compute_frac": roof.get("compute", {}).get("frac") if roof else None,
               "lanes_per_inst": roof.get("compute", {}).get("lanes_per_inst") if roof else None,
               "kernel_ms_per_frame": {k: round(v / steps, 3) for k, v in roof["kernel_ms_by_class"].items()} if roof else None,
               "work_per_ray": work,
               "commit_s": round(commit_s, 3), "refit_commit_ms": refit_ms, "gpu_build": gpu_build,
               "ms_per_frame_kernels_timed": round(dt_timed * 1e3 / steps, 3),
               "note": "ms_per_frame / Mrays_s: kernel timing off (shadow rays of bounce b overlap the closest-hit rays of bounce b + 1); kernel_ms_per_frame, frac: a second pass with per-kernel HIP events, which runs the launches one after the other"}
        return rec
    finally:
        ctx.close()


RESTIR_EXTRAS = {"restir_garage_1080p": "garage", "restir_atrium_1080p": "sponza"}     # the reference's shipping frame (3 DispatchRays: pass 1 + temporal + spatial reuse)


def time_restir(rt, dev_index, kind, frames=6):
    """The reference's own frame — rtx_render_restir: pass 1 + temporal + spatial reuse with its defines nee 4 / bounces 3 (Common_v6.hlsl:8-12) — at 1920x1080 on the
    reference's start-up scene (garage.obj + monke.obj, Renderer.cpp:363) or the Sponza-class atrium: ms per frame once the history is warm, rays by type, and the
    per-class kernel times of the wavefront stages (raygen / trace_closest = persistent closest-hit launches / shade = the stage kernels / trace_shadow = persistent
    any-hit launches).  `frac` prices the dominant class by its algorithmic bytes (rays: 48 B each as for the path tracer's separate kernels; stage kernels: the
    reference's per-pixel records), `frame_frac` the whole frame."""
    import torch
    W, H = 1920, 1080
    gd = os.path.join(ROOT, "tests", "golden")
    if kind == "garage":
        scene, source = rt.Scene.from_obj([os.path.join(gd, "garage.obj"), os.path.join(gd, "monke.obj")], gd + "/"), "tests/golden/garage.obj + monke.obj (the reference's start-up scene)"
    else:
        scene, source = make_scene(rt, kind)
    ctx = rt.Context(dev_index)
    try:
        ctx.upload(scene, W / H)
        vp = scene.view_proj(W / H)
        ctx.set_camera(*vp); ctx.set_camera(*vp)
        ctx.restir_reset(); ctx.clear(W, H)
        p = rt.Params(width=W, height=H, spp=1, max_bounces=3, nee_samples=4, flags=0)
        for f in range(2):
            ctx.render_restir(p.copy(frame_seed=1 + f))          # allocations + a history for the temporal pass
        torch.cuda.synchronize()
        t0 = time.perf_counter(); per_step = []
        for f in range(frames):
            ts = time.perf_counter(); ctx.render_restir(p.copy(frame_seed=3 + f)); per_step.append((time.perf_counter() - ts) * 1e3)
        dt = time.perf_counter() - t0
        ctx.set_option(rt.OPT_KERNEL_TIMING, 1)
        kms = np.zeros(rt.K_COUNT); rays = np.zeros(3)
        for f in range(frames):
            ctx.render_restir(p.copy(frame_seed=3 + frames + f))
            st = ctx.stats()
            kms += np.array(st.kernel_ms[:]); rays += np.array([st.rays_primary, st.rays_extension, st.rays_shadow], dtype=np.float64)
        by = {rt.KERNEL_NAMES[i]: float(kms[i]) / frames for i in rt.KERNEL_NAMES if kms[i] > 0}
        dom = max(by, key=by.get) if by else None
        # algorithmic bytes: 48 B per ray of a traversal launch; the stage kernels ("shade") move the reference's per-pixel records — pass 1 writes Reservoir_DI 40 +
        # Reservoir_GI 40 + SampleData 60 B; pass 2 reads them and the reprojected pixel's 140 B and writes the two reservoirs; pass 3 reads the pixel's 140 B and 3 + 3 neighbours'
        # reservoir + sample data (100 B each), writes both reservoirs and the sample data to the history and adds to the accumulation buffer (16 B read + 16 B written)
        px_bytes = 140 + (140 + 140 + 80) + (140 + 6 * 100 + 80 + 60 + 32)
        alg = {"trace_closest": 48.0 * (rays[0] + rays[1]) / frames, "trace_shadow": 48.0 * rays[2] / frames, "shade": float(px_bytes) * W * H}
        frac = alg[dom] / (by[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS if dom in alg else None
        frame_alg = (224.0 * rays[1] + 96.0 * rays[2]) / frames + float(px_bytes) * W * H            # the path tracer's per-ray prices + the records
        return {"ms_per_frame": round(dt * 1e3 / frames, 3), "ms_per_frame_stats": step_stats(per_step), "Mrays_s": round(float(rays.sum()) / frames / (dt / frames) / 1e6, 1), "triangles": int(scene.num_triangles), "scene": source,
                "rays_per_frame": {"primary": int(rays[0] / frames), "extension": int(rays[1] / frames), "shadow": int(rays[2] / frames)},
                "kernel_ms_per_frame": {k: round(v, 3) for k, v in by.items()}, "dominant_kernel": dom, "frac": round(frac, 5) if frac else None,
                "frame_frac": round(frame_alg / (dt / frames) / 1e9 / HBM_PEAK_GBS, 5),
                "alg_bytes_model": f"dominant class: 48 B per ray of a traversal launch, {px_bytes} B of reservoir / sample records per pixel for the stage kernels; frame: 224*N_ext + 96*N_shadow + {px_bytes}*N_px",
                "form": "wavefront stages (csrc/rtx_restir_wave.hpp); RTX_OPT_RESTIR_WAVEFRONT=0 is the thread-per-pixel form",
                "params": "1920x1080, nee_samples 4, bounces 3, 1 frame per step, history warm"}
    finally:
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_64spp_8b", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the BVH workloads timed beside the headline (N = 1, default workload only)")
    ap.add_argument("--paths-per-batch", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="gloo: host-staged gather, for testing the N>1 flow on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="force this CUDA device on every rank (testing only)")
    ap.add_argument("--checksum", action="store_true", help="add a checksum of the final accumulation buffer to the JSON line")
    ap.add_argument("--opt", action="append", default=[], help="tuning: rtx option id=value (repeatable)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:          # no launcher: be the launcher (before torch / the GPU are touched)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} set by the launcher")

    import torch
    dev_index = args.device if args.device >= 0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rt = graft.load_package()
    from royaltracer_dx_amd import sharding
    dist = None
    ranks_seen = 1
    if world > 1:
        dist, rank, world = sharding.init_process_group(args.dist_backend, dev)     # backend "nccl" is RCCL on ROCm
        ranks_seen = dist.get_world_size()
    kind, W, H, spp, bounces, nee, flags = WORKLOADS[args.workload]
    scene, scene_source = make_scene(rt, kind)
    ctx = rt.Context(dev_index)
    for kv in args.opt:                      # tuning knobs must be set before the scene is committed
        k, v = kv.split("=")
        ctx.set_option(int(k), int(v))
    ctx.upload(scene, W / H)
    if args.paths_per_batch:
        ctx.set_option(rt.OPT_PATHS_PER_BATCH, args.paths_per_batch)
    # tile edge of the round-robin deal: 64 on one GPU (the order of the path slots the headline was tuned with); 32 when the frame is sharded — four times as many tiles
    # per rank even out which ranks get the empty background (max / mean over 8 ranks 1.15 -> 1.04, slowest rank 3.16 -> 2.95 ms: tools/shard_time.py, profiles/r03_shard_time.md)
    TILE = 64 if world == 1 else 32
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.bind_accum(accum.data_ptr(), accum.numel() * 4)
    params = rt.Params(width=W, height=H, spp=spp, sample_base=1, max_bounces=bounces, nee_samples=nee, rr_start=3,
                       frame_seed=1, flags=flags, tile_size=TILE, shard_rank=rank, shard_count=world)
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

    # the warm-up frames run in the timed frames' mode (with per-kernel events the shadow-ray kernel of bounce b does not overlap the closest-hit kernel of bounce b + 1): a
    # rocprofv3 trace of this command then holds frames of ONE kind, and its per-kernel averages reproduce the event times of the line
    ctx.set_option(rt.OPT_KERNEL_TIMING, 0 if args.no_kernel_timing else 1)
    for i in range(args.warmup):
        frame(i)
    kms = np.zeros(rt.K_COUNT); kitems = np.zeros(rt.K_COUNT); klaunch = np.zeros(rt.K_COUNT)
    rays = np.zeros(3)
    env = gpu_env(dev_index) if rank == 0 else None            # clocks / power / temperature / versions: once, BEFORE the timed region
    # per-step times from events on the stream the frames run on (recording an event costs no synchronise; they are read after the closing barrier)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        st = frame(args.warmup + i)
        ev[i][1].record(stream)
        kms += np.array(st.kernel_ms[:]); kitems += np.array(st.kernel_items[:], dtype=np.float64); klaunch += np.array(st.kernel_launches[:], dtype=np.float64)
        rays += np.array([st.rays_primary, st.rays_extension, st.rays_shadow], dtype=np.float64)
    barrier()
    dt = time.perf_counter() - t0
    dt_max, rays_all, dt_ranks = dt, rays, [dt]
    if world > 1:
        cdev = dev if args.dist_backend == "nccl" else None
        dt_max = sharding.max_over_ranks(dist, dt, cdev)
        rays_all = sharding.sum_over_ranks(dist, rays, cdev)
        per = np.zeros(world); per[rank] = dt
        dt_ranks = list(sharding.sum_over_ranks(dist, per, cdev))
    ms_per_step = dt_max * 1e3 / max(args.steps, 1)
    value = float(rays_all.sum()) / dt_max / 1e6 if dt_max > 0 else 0.0
    step_ms = [a.elapsed_time(b) for a, b in ev]               # this rank's frames, GPU time on the frames' stream

    if rank == 0:
        sha = None
        if args.checksum:
            import hashlib
            sha = hashlib.sha1(accum.cpu().numpy().tobytes()).hexdigest()
        # ---- roofline of the dominant kernel (HIP-event time per class, measured in the timed region, on the context's stream) ----
        roof = None
        if not args.no_kernel_timing:
            local_px = float(W) * H * spp * args.steps / world
            roof = roofline_record(rt, args.workload if world == 1 else "", kms, kitems, klaunch, rays, local_px, args.steps)
            comp = (roof or {}).get("compute")
            if comp and comp.get("valu_inst_per_launch") and comp.get("lanes_per_inst"):
                # WORK metric beside the self-referential compute fraction: VALU lane-instructions the dominant class issues per ray it handles
                n_dom = {"bounce_fused": rays[1] + rays[2], "trace_closest": rays[0] + rays[1], "trace_shadow": rays[2]}.get(roof["kernel"])
                if n_dom:
                    comp["valu_lane_inst_per_ray"] = round(comp["valu_inst_per_launch"] * roof["launches"] * comp["lanes_per_inst"] / float(n_dom), 1)
                    comp["work_model"] = ("valu_lane_inst_per_ray = SQ_INSTS_VALU per launch x launches of the timed region x active lanes per instruction / rays the class handled "
                                          "(bounce_fused: extension + shadow rays; the tiny-scene path has no tree: node steps / triangle tests per ray are reported for C3 / C5 under extra.*.work_per_ray)")
        # ---- the general BVH path beside the headline: driver-timed numbers for C3 / C5 (GPU only, ~2 s each + scene build) ----
        extra = None
        if world == 1 and not args.no_extra and args.workload == "cornell_1080p_64spp_8b":
            ctx.close(); ctx = None
            del accum
            torch.cuda.empty_cache()
            extra = {}
            for wl in EXTRA_WORKLOADS:
                try:
                    extra[wl] = time_extra(rt, dev_index, wl)
                except Exception as e:                       # the headline line must survive a failing extra
                    extra[wl] = {"error": str(e)[:200]}
            for name, k in RESTIR_EXTRAS.items():            # the reference's shipping frame (ReSTIR DI + GI), ~1 s each
                try:
                    extra[name] = time_restir(rt, dev_index, k)
                except Exception as e:
                    extra[name] = {"error": str(e)[:200]}
        # ---- CPU baseline: the oracle (a port of the reference's shader math) on this host's cores ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            orc = graft.load_oracle()
            o = orc.Oracle().load(scene, W / H)
            cores = os.cpu_count() or 1
            o.set_threads(cores)
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
               "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "ms_per_step_by_rank": [round(t * 1e3 / max(args.steps, 1), 3) for t in dt_ranks],
               "ms_per_step_min": step_stats(step_ms)["min"] if step_ms else None, "ms_per_step_median": step_stats(step_ms)["median"] if step_ms else None,
               "ms_per_step_stats": dict(step_stats(step_ms), source="HIP events around every step on rank 0's stream (ms_per_step itself is wall time over all steps, max over ranks)") if step_ms else None,
               "env": env,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "asset (assets/*.obj loaded through the OBJ / MTL reader)" if scene_source.startswith("asset:") else "synthetic",
               "config": {"workload": args.workload, "scene": scene_source, "width": W, "height": H, "spp": spp,
                          "max_bounces": bounces, "nee_samples": nee, "rr_start": 3, "flags": flags,
                          "triangles": int(scene.num_triangles), "parallelism": f"pixel-tiles/{world}", "tile_size": TILE,
                          "rays_per_frame": {"primary": int(rays_all[0] / max(args.steps, 1)), "extension": int(rays_all[1] / max(args.steps, 1)),
                                             "shadow": int(rays_all[2] / max(args.steps, 1))}},
               "roofline": roof, "cpu_baseline": cpu}
        if extra is not None:
            out["extra"] = extra
        if sha:
            out["accum_sha1"] = sha
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
