#!/usr/bin/env python3
"""Loops of ONE kernel in a gfx950 device assembly file (hipcc --cuda-device-only -S) and what they hold: VALU instructions, SGPR-spill traffic (v_readlane / v_writelane)
and the s_nop hazard padding -- to see whether spill code sits in a hot loop or around it.
usage: asm_loops.py file.s <substring of the mangled kernel name>"""
import re, sys
name = sys.argv[2]
on, body = False, []
for l in open(sys.argv[1]):
    if re.match(r"^_Z\w*:", l):
        on = name in l.split(":")[0]
        continue
    if on and l.startswith(".Lfunc_end"):
        break
    if on:
        body.append(l.rstrip("\n"))
lab, ins = {}, []
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = len(ins)
        continue
    m = re.match(r"^\s+([a-z]\w+)\s*(.*)", l)
    if m and not l.strip().startswith((".", ";")):
        ins.append((m.group(1), m.group(2)))
loops = set()
for i, (op, a) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = a.split()[0].strip() if a.split() else ""
        if t in lab and lab[t] <= i:
            loops.add((lab[t], i))
cnt = lambda seg, pre: sum(1 for o, _ in seg if o.startswith(pre))
print(f"{len(ins)} instructions, {cnt(ins, 'v_')} VALU, {cnt(ins, 'v_readlane')} v_readlane, {cnt(ins, 'v_writelane')} v_writelane, {cnt(ins, 's_nop')} s_nop; {len(loops)} backward branches")
print("| loop (instruction range) | instructions | VALU | v_readlane | v_writelane | s_nop |\n|---|---|---|---|---|---|")
for a, b in sorted(loops):
    seg = ins[a:b + 1]
    print(f"| {a}-{b} | {b - a + 1} | {cnt(seg, 'v_')} | {cnt(seg, 'v_readlane')} | {cnt(seg, 'v_writelane')} | {cnt(seg, 's_nop')} |")
