#!/usr/bin/env python3
"""Registers, spills, LDS and occupancy of every kernel in csrc/rtx_kernels.hip as hipcc reports them (-Rpass-analysis=kernel-resource-usage); no GPU needed.
usage: python3 tools/kernel_resources.py [filter-substring ...] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
flt = args[:args.index("--")] if "--" in args else args
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "royaltracer-dx_amd", "csrc", "rtx_kernels.hip"), "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace("rtx::", "")
        cur = {"kernel": name}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("| kernel | VGPRs | AGPRs | SGPRs | spilled SGPRs | spilled VGPRs | scratch B/lane | LDS B | waves/SIMD |\n|---|---|---|---|---|---|---|---|---|")
for r in rows:
    if flt and not any(s in r["kernel"] for s in flt):
        continue
    print(f"| {r['kernel']} | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('SGPRs')} | {r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | {r.get('ScratchSize')} | {r.get('LDS Size')} | {r.get('Occupancy')} |")
