#!/usr/bin/env python3
"""Copy what tools/record_round.sh left under gpurun_out/<tag>_* into tracked summaries under profiles/ (no GPU needed).   usage: collect_profiles.py <tag>"""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def kernel_table(d, per=1, what="launch"):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(<[^>(]*>)?|__amd_rocclr_\w+)", r["Name"])
            if m:
                rows.append((m.group(1), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6))
    regs = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+(<[^>(]*>)?|__amd_rocclr_\w+)", r["Kernel_Name"])
            if m:
                regs.setdefault(m.group(1), (r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
    rows.sort(key=lambda x: -x[2])
    out = [f"| kernel | launches / {what} | ms / {what} | avg us / launch | LDS B / workgroup | scratch B / lane |", "|---|---|---|---|---|---|"]
    for k, c, t in rows:
        v = regs.get(k, ("-", "-", "-"))
        out.append(f"| {k} | {c / per:g} | {t / per:.3f} | {t / c * 1e3:.1f} | {v[1]} | {v[2]} |")
    return "\n".join(out), sum(t for _, _, t in rows) / per


for wl, name in (("cornell", "c2"), ("sponza", "c3"), ("bistro", "c5")):
    src = os.path.join(G, f"{tag}_roof_{wl}")
    if os.path.isfile(os.path.join(src, "roofline.json")):
        shutil.copy(os.path.join(src, "roofline.json"), os.path.join(P, f"{tag}_roof_{wl}.json"))
        shutil.copy(os.path.join(src, "roofline.md"), os.path.join(P, f"{tag}_roof_{wl}.md"))
    kt = os.path.join(G, f"{tag}_kt_{wl}")
    if os.path.isdir(kt):
        tab, tot = kernel_table(kt, per=3, what="frame")            # bench.py --steps 2 --warmup 1: three frames are in the trace
        line = [l for l in open(os.path.join(kt, "bench.log")) if l.startswith("{")]
        with open(os.path.join(P, f"{tag}_{name}.md"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats of `python3 bench.py --workload <{wl}> --steps 2 --warmup 1 --no-cpu-baseline --no-extra` ({tag})\n\n"
                    f"Three frames are in the trace (one warm-up + two timed), all three with per-kernel events, i.e. the launches one after the other (bench.py warms up in the timed frames' mode since round 4); "
                    f"per frame = totals / 3.  Sum of kernel time per frame: {tot:.2f} ms.  Registers / occupancy: profiles/{tag}_kernel_resources.md.\n\n" + tab + "\n")
            if line:
                d = json.loads(line[-1])
                f.write(f"\nbench line of the same run: {d['value']} {d['unit']}, {d['ms_per_step']} ms per frame; roofline " + json.dumps(d.get("roofline"))[:1800] + "\n")
                # does the trace reproduce the line?  average duration of the dominant class's launches: trace vs HIP events of the same run; frac recomputed from the trace
                roof = d.get("roofline") or {}
                cls = {"trace_closest": "k_trace_closest", "trace_shadow": "k_trace_shadow", "shade": "k_shade", "bounce_fused": "k_bounce_small", "raygen": "k_raygen"}.get(roof.get("kernel"), "")
                calls = tms = 0.0
                for fcsv in glob.glob(os.path.join(kt, "**", "*kernel_stats.csv"), recursive=True):
                    for r in csv.DictReader(open(fcsv)):
                        if cls and cls in r["Name"]:
                            calls += int(r["Calls"]); tms += float(r["TotalDurationNs"]) / 1e6
                if calls and roof.get("avg_launch_ms"):
                    avg_tr = tms / calls
                    frac_tr = roof["alg_bytes_per_launch"] / (avg_tr * 1e-3) / 1e9 / roof["peak"]
                    f.write(f"\nCross-check (VERDICT r03 item 2): `{cls}` averages {avg_tr:.4f} ms per launch in this trace ({int(calls)} launches) against {roof['avg_launch_ms']:.4f} ms by the HIP events of the same run "
                            f"(ratio {avg_tr / roof['avg_launch_ms']:.3f}); `frac` recomputed from the trace {frac_tr:.4f}, the line says {roof['frac']:.4f}.\n")
for sc in ("garage", "sponza", "bistro"):
    kt = os.path.join(G, f"{tag}_kt_restir_{sc}")
    if os.path.isdir(kt):
        tab, tot = kernel_table(kt, per=4, what="frame")
        log = [l.strip() for l in open(os.path.join(kt, "run.log")) if " frame " in l]
        with open(os.path.join(P, f"{tag}_restir_{sc}.md"), "w") as f:
            f.write(f"# ReSTIR frame (pass 1 + 2 + 3 as wavefront stages, two pipeline lanes), 1920x1080, nee 4, bounces 3, `{sc}` — rocprofv3 --kernel-trace --stats of `python3 tools/restir_time.py {sc} frames=4` ({tag})\n\n"
                    f"Sum of kernel time per frame {tot:.2f} ms (the two lanes overlap, so the frame is shorter than the sum):\n\n" + tab + "\n\n```\n" + "\n".join(log) + "\n```\n")
# calibration of the compute roofline: single-opcode loops, 1:1 pairs (do class costs add up?), the replay loops of the hot kernels
cal = {}
for wl in ("cornell", "sponza", "bistro"):
    fn = os.path.join(P, f"{tag}_roof_{wl}.json")
    if os.path.isfile(fn):
        cal[wl] = json.load(open(fn))
if cal:
    any_ = next(iter(cal.values()))
    single = {r["what"]: r["simd_cycles_per_inst"] for r in any_["single_opcode_loops"]}
    pairs = {"fma+min": ("v_fma_f32", "v_min_f32"), "fma+add_u32": ("v_fma_f32", "v_add_u32"), "min+cvt": ("v_min_f32", "v_cvt_f32_ubyte0"), "fma+rcp": ("v_fma_f32", "v_rcp_f32"),
             "add_u32+min": ("v_add_u32", "v_min_f32"), "fma+mul": ("v_fma_f32", "v_mul_f32"), "fma+mov": ("v_fma_f32", "v_mov_b32"), "add_u32+cvt": ("v_add_u32", "v_cvt_f32_ubyte0")}
    L = [f"# Calibration of the compute roofline on gfx950 (MI355X) — `tools/valu_peak.hip calib` under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU ({tag})", "",
         "All loops: 8 waves per SIMD, independent destination registers, `SIMD-cycles / inst` = 1024 SIMDs x GRBM_GUI_ACTIVE / 8 / wave-instructions.", "",
         "**1. Why round 2's `valu_busy_frac` exceeded 1.**  `4 x SQ_ACTIVE_INST_VALU / SIMD-cycles` charges every VALU instruction 4 cycles: on the saturated loops of this tool it read 1.74 for",
         "`v_fma_f32` (2.3 cycles each), 1.25 for `v_mul_f32` / `v_add_u32` / `v_mov_b32` (3.0-3.2) and 0.96 for `v_min_f32` / `v_cvt_f32_ubyte0` / `v_cndmask_b32` (4.2) (first calibration run of the round).", "",
         "**2. Per-class costs do not add up** (so `sum n_class x c_class / SIMD-cycles` is no roofline either): 1:1 mixes of two opcodes against the mean of their single-opcode costs:", "",
         "| mix | measured SIMD-cycles / inst | mean of the two alone | measured / mean |", "|---|---|---|---|"]
    for name, (a, b) in pairs.items():
        if name in single and a in single and b in single:
            pred = 0.5 * (single[a] + single[b])
            L.append(f"| {name} | {single[name]:.3f} | {pred:.3f} | {single[name] / pred:.2f} |")
    if "cvt/fma/max3/cmp" in single:
        L.append(f"| cvt/fma/max3/cmp (1:1:1:1) | {single['cvt/fma/max3/cmp']:.3f} | - | - |")
    L += ["", "A stream of `v_fma_f32` and `v_min_f32` issues at the cost of `v_fma_f32` alone, `v_add_u32` + `v_min_f32` slower than their mean, `v_fma_f32` + `v_rcp_f32` slower than the sum: operand",
          "ports, encodings and the transcendental unit interact.  The peak of an instruction mix therefore has to be MEASURED.", "",
          "**3. The roofline used by bench.py**: for every hot kernel a saturated replay loop with the kernel's own dynamic class shares (tools/gen_mix.py); `compute frac` = the kernel's VALU",
          "instructions per SIMD-cycle / the replay's.  It cannot exceed 1; the replay loops themselves read 1.00 by construction.", "",
          "| workload | kernel | replay: inst / SIMD-cycle (SIMD-cycles / inst) | kernel: inst / SIMD-cycle | compute frac | lanes / 64 |", "|---|---|---|---|---|---|"]
    for wl, d in cal.items():
        for k in d["kernels"]:
            if k.get("compute_frac") is not None:
                L.append(f"| {wl} | {k['kernel']} | {k['compute_peak_inst_per_simd_cycle']:.3f} ({1.0 / k['compute_peak_inst_per_simd_cycle']:.2f}) | {k['inst_per_simd_cycle']:.3f} | {k['compute_frac']:.3f} | {(k.get('lanes_per_valu') or 0) / 64.0:.2f} |")
    L += ["", "## Single-opcode and pair loops of this run", "", "| loop | SIMD-cycles / inst |", "|---|---|"]
    for r in any_["single_opcode_loops"]:
        L.append(f"| {r['what']} | {r['simd_cycles_per_inst']:.3f} |")
    open(os.path.join(P, f"{tag}_valu_calib.md"), "w").write("\n".join(L) + "\n")
pm = os.path.join(G, f"{tag}_pmc_restir_sponza", "pmc_summary.md")
if os.path.isfile(pm):
    with open(os.path.join(P, f"{tag}_pmc_restir.md"), "w") as f:
        f.write(f"# Counters of the ReSTIR frame on the atrium, 2 frames (`PMC_CMD=\"tools/restir_time.py sponza frames=2\" tools/pmc_run.sh`, {tag}; columns: tools/pmc_summary.py)\n\n" + open(pm).read())
res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True).stdout
open(os.path.join(P, f"{tag}_kernel_resources.md"), "w").write(f"# Registers, spills, LDS and occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; tools/kernel_resources.py, {tag})\n\n" + res)
print("profiles written:", sorted(f for f in os.listdir(P) if f.startswith(tag)))
