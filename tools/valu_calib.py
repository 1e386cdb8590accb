#!/usr/bin/env python3
"""Merge the counter passes of one workload (tools/pmc_summary.py) with the instruction-mix replay of its kernels (tools/valu_peak.hip `calib`, tools/gen_mix.py):
-> <dir>/roofline.json / .md.   usage: valu_calib.py <dir> <workload>     (see tools/roofline_run.sh)"""
import csv, glob, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """what bench.py recomputes to tell whether a tracked profile still describes the loaded kernels"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "royaltracer-dx_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    d, wl = sys.argv[1], sys.argv[2]
    rows = json.load(open(os.path.join(d, "pmc_summary.json")))
    cyc = {}
    for f in glob.glob(os.path.join(d, "calib_g1", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
            cyc.setdefault(k, {})[r["Counter_Name"]] = cyc.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    calib, singles = {}, []
    for l in open(os.path.join(d, "calib_plain.log")):
        m = re.match(r"calib (k_\w+<\d+>) (.*?)\s+wave-inst=(\d+)\s+wave-cycles=\d+\s+clock=([\d.]+) GHz\s+wall=([\d.]+) ms", l)
        if not m:
            continue
        kid, name, ninst = m.group(1), m.group(2).strip(), float(m.group(3))
        c = cyc.get(kid, {})
        if not c.get("GRBM_GUI_ACTIVE"):
            continue
        simd_cycles = 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
        rec = {"loop": kid, "what": name, "wave_inst": ninst, "simd_cycles_per_inst": simd_cycles / ninst, "inst_per_simd_cycle": ninst / simd_cycles,
               "counter_inst": c.get("SQ_INSTS_VALU")}
        if kid.startswith("k_mix"):
            calib[name] = rec
        else:
            singles.append(rec)
    out = {"workload": wl, "kernel_source_sha": kernel_source_hash(), "kernels": [], "replay": calib, "single_opcode_loops": singles,
           "model": "compute_frac = (SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)) of the kernel / the same figure of a saturated replay loop of its own instruction-class mix"}
    for r in rows:
        k = dict(r)
        cycs = r.get("gpu_cycles") or 0.0
        k["inst_per_simd_cycle"] = r["valu_inst"] / (1024.0 * cycs) if cycs else None
        rp = calib.get(r["kernel"])
        if rp and k["inst_per_simd_cycle"]:
            k["compute_peak_inst_per_simd_cycle"] = rp["inst_per_simd_cycle"]
            k["compute_frac"] = k["inst_per_simd_cycle"] / rp["inst_per_simd_cycle"]
        out["kernels"].append(k)
    json.dump(out, open(os.path.join(d, "roofline.json"), "w"), indent=1)
    with open(os.path.join(d, "roofline.md"), "w") as f:
        f.write(f"# Roofline counters, {wl} (tools/roofline_run.sh; kernel sources {out['kernel_source_sha']})\n\n")
        f.write("`compute frac` = VALU instructions per SIMD-cycle of the kernel / of a saturated replay loop with the kernel's own instruction-class mix (<= 1 by construction); "
                "`lanes` = active lanes per VALU instruction of 64; `HBM bytes` = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes) per launch.\n\n")
        f.write("| kernel | launches | VALU inst | inst / SIMD-cycle | replay peak | compute frac | lanes | HBM bytes / launch | L2 hit |\n|---|---|---|---|---|---|---|---|---|\n")
        g = lambda v, fmt: "-" if v is None else format(v, fmt)
        for k in out["kernels"]:
            f.write(f"| {k['kernel']} | {k['launches']} | {k['valu_inst']:.4g} | {g(k.get('inst_per_simd_cycle'), '.3f')} | {g(k.get('compute_peak_inst_per_simd_cycle'), '.3f')} | "
                    f"{g(k.get('compute_frac'), '.3f')} | {g(k.get('lanes_per_valu'), '.1f')} | {k['hbm_bytes'] / max(k['launches'], 1):.4g} | {g(k.get('l2_hit'), '.2f')} |\n")
        f.write("\n## Saturated loops of this run (8 waves / SIMD, independent registers)\n\n| loop | what | SIMD-cycles / inst | inst / SIMD-cycle |\n|---|---|---|---|\n")
        for rec in list(calib.values()) + singles:
            f.write(f"| {rec['loop']} | {rec['what']} | {rec['simd_cycles_per_inst']:.3f} | {rec['inst_per_simd_cycle']:.3f} |\n")
    print(open(os.path.join(d, "roofline.md")).read())


if __name__ == "__main__":
    main()
