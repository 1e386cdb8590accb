#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes) into a
small JSON + markdown pair under profiles/.

usage: rocprof_summary.py <kt_dir> <fetch_dir> <write_dir> <out_prefix> [note]

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read, so it is doubled.  (Uncalibrated
for narrow / gathered accesses: treat as an upper estimate there.)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?|__amd_rocclr_\w+|vectorized_elementwise_kernel)", name)
    return m.group(1).replace(" ", "") if m else name[:40]


def load_counter(d, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


def main():
    kt, fd, wd, out = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    stats = {}
    for f in glob.glob(os.path.join(kt, "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            e = stats.setdefault(k, {"calls": 0, "total_ms": 0.0, "pct": 0.0})
            e["calls"] += int(r["Calls"]); e["total_ms"] += float(r["TotalDurationNs"]) / 1e6; e["pct"] += float(r["Percentage"])
            e["avg_us"] = e["total_ms"] * 1e3 / e["calls"]
    regs = {}
    for f in glob.glob(os.path.join(kt, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            regs.setdefault(short(r["Kernel_Name"]), {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "scratch": r.get("Scratch_Size"), "wg": r.get("Workgroup_Size")})
    fetch, write = load_counter(fd, "FETCH_SIZE"), load_counter(wd, "WRITE_SIZE")
    rows = []
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
        fk = sum(fetch.get(k, [])) / max(len(fetch.get(k, [])), 1)
        wk = sum(write.get(k, [])) / max(len(write.get(k, [])), 1)
        hbm = fk * 2 * 1024 + wk * 1024
        s.update({"fetch_kib_per_launch": round(fk, 1), "write_kib_per_launch": round(wk, 1), "hbm_bytes_per_launch": round(hbm),
                  "hbm_gbs": round(hbm / (s["avg_us"] * 1e-6) / 1e9, 1) if s["avg_us"] else 0.0, **regs.get(k, {})})
        rows.append((k, s))
    json.dump({"note": note, "kernels": dict(rows)}, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write(f"# rocprofv3 summary\n\n{note}\n\n")
        f.write("| kernel | calls | total ms | avg us | % | VGPR | LDS B | FETCH KiB/launch (raw) | WRITE KiB/launch | HBM bytes/launch (2*FETCH+WRITE) | HBM GB/s |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, s in rows:
            f.write(f"| {k} | {s['calls']} | {s['total_ms']:.3f} | {s['avg_us']:.1f} | {s['pct']:.2f} | {s.get('vgpr')} | {s.get('lds')} | {s['fetch_kib_per_launch']} | {s['write_kib_per_launch']} | {s['hbm_bytes_per_launch']} | {s['hbm_gbs']} |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
