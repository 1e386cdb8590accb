#!/usr/bin/env python3
"""Aggregate the counter passes written by tools/pmc_run.sh into one table per kernel (template instantiations kept apart).

Derived columns (MI355X_MICROARCH.md, HBM / rocprofv3 section): SQ_ACTIVE_INST_*, SQ_WAVE_CYCLES, SQ_WAIT_* count quad-cycles;
FETCH_SIZE is in KiB and under-reports by 2x on gfx950, WRITE_SIZE in KiB.
"""
import csv, glob, os, re, sys, json
from collections import defaultdict

def short(name):
    m = re.match(r"(?:void )?(?:rtx::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name

def main():
    out = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
    rows = []
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        if not k.startswith("k_"): continue
        g = lambda n: c.get(n, 0.0)
        ncall = min(v for v in calls[k].values() if v > 0)          # (some counters come as several rows per dispatch: the launch count is the smallest row count)
        cyc = g("GRBM_GUI_ACTIVE") / 8.0                         # summed over the 8 XCDs
        valu = g("SQ_INSTS_VALU")
        d = {"kernel": k, "launches": ncall, "gpu_cycles": cyc,
             "valu_inst": valu, "salu_inst": g("SQ_INSTS_SALU"), "lds_inst": g("SQ_INSTS_LDS"), "vmem_rd_inst": g("SQ_INSTS_VMEM_RD"), "vmem_wr_inst": g("SQ_INSTS_VMEM_WR"),
             "lanes_per_valu": g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") if g("SQ_ACTIVE_INST_VALU") else None,
             "cycles_per_valu": 4.0 * g("SQ_ACTIVE_INST_VALU") / valu if valu else None,
             "valu_pipe_util": 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * cyc) if cyc else None,      # 256 CUs x 4 SIMDs
             "waves_per_simd": 4.0 * g("SQ_WAVE_CYCLES") / (1024.0 * cyc) if cyc else None,
             "wave_active_frac": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
             "wave_wait_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
             "l2_hit": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else None,
             "l2_req": g("TCC_REQ_sum"),
             "hbm_bytes": 2.0 * 1024.0 * g("FETCH_SIZE") + 1024.0 * g("WRITE_SIZE"),
             "fma_f32": g("SQ_INSTS_VALU_FMA_F32"), "add_f32": g("SQ_INSTS_VALU_ADD_F32"), "mul_f32": g("SQ_INSTS_VALU_MUL_F32"), "trans_f32": g("SQ_INSTS_VALU_TRANS_F32"),
             "int32": g("SQ_INSTS_VALU_INT32"), "cvt": g("SQ_INSTS_VALU_CVT"),
             "wave_parked_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None}
        rows.append(d)
    json.dump(rows, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    cols = ["kernel", "launches", "gpu_cycles", "valu_inst", "lanes_per_valu", "cycles_per_valu", "valu_pipe_util", "waves_per_simd", "wave_active_frac", "wave_wait_frac", "salu_inst", "lds_inst", "vmem_rd_inst", "l2_hit", "hbm_bytes", "fma_f32", "add_f32", "mul_f32", "trans_f32", "int32", "cvt", "wave_parked_frac"]
    fmt = lambda v: "-" if v is None else (v if isinstance(v, str) else (f"{v:.3g}" if abs(v) < 1000 else f"{v:.4g}"))
    lines = ["| " + " | ".join(cols) + " |", "|" + "---|" * len(cols)]
    for d in rows: lines.append("| " + " | ".join(fmt(d[c]) for c in cols) + " |")
    md = "\n".join(lines) + "\n"
    open(os.path.join(out, "pmc_summary.md"), "w").write(md)
    print(md)

if __name__ == "__main__":
    main()
