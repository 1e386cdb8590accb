#!/usr/bin/env python3
"""Static instruction statistics of one kernel in a gfx950 device assembly file (hipcc --cuda-device-only -S).
usage: asm_stats.py file.s <substring of the mangled kernel name> [top N]"""
import collections, re, sys
name, top = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0
on, h = False, collections.Counter()
for l in open(sys.argv[1]):
    if re.match(r"^_Z\w*:", l):
        on = name in l.split(":")[0]
        continue
    if on and l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", l)
    if on and m:
        h[re.sub(r"_e(32|64)$", "", m.group(1))] += 1
valu = sum(v for k, v in h.items() if k.startswith("v_"))
print(f"total {sum(h.values())} valu {valu} v_mov {h['v_mov_b32']} readlane {h['v_readlane_b32']} writelane {h['v_writelane_b32']} scratch {sum(v for k,v in h.items() if k.startswith('scratch'))} s_waitcnt {h['s_waitcnt']} s_nop {h['s_nop']}")
for k, v in h.most_common(top):
    print(f"  {v:5d} {k}")
