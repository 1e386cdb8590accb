#!/usr/bin/env python3
"""Rewrite VOP2-encoded FP32 mul / add / sub / fmac in a gfx950 device assembly file to their VOP3 forms.
Measured (profiles/r02_valu_peak.md): a wave64 v_mul_f32 / v_add_f32 / v_sub_f32 / v_fmac_f32 in the VOP2 (e32) encoding the compiler prefers occupies the SIMD
for 3.1 cycles, the same operation in the VOP3 (e64) encoding for 2.3.  Results are bit-identical (same operation, other encoding).
Lines with a literal constant operand (VOP3 takes none on gfx9), DPP / SDWA forms and anything else are left alone.
usage: vop3_rewrite.py in.s out.s"""
import re
import sys

LIT = re.compile(r"0x[0-9a-fA-F]+|(?<![\w.])\d+\.\d+e[+-]?\d+")
INLINE = {"0", "0.5", "1.0", "2.0", "4.0", "-0.5", "-1.0", "-2.0", "-4.0"}


def operand_ok(op):
    op = op.strip()
    if re.fullmatch(r"-?\|?[vs]\d+\|?|-?\|?[vs]\[\d+:\d+\]\|?|vcc_lo|vcc_hi|m0", op):
        return True
    if op in INLINE:
        return True
    if re.fullmatch(r"-?\d+", op) and -16 <= int(op) <= 64:
        return True
    return False                      # literal constants and anything unusual: keep the VOP2 form


def main():
    n = {"mul": 0, "add": 0, "sub": 0, "subrev": 0, "fmac": 0, "kept": 0}
    out = []
    pat = re.compile(r"^(\s*)v_(mul|add|sub|subrev|fmac)_f32_e32\s+(v\d+),\s*([^,]+),\s*([^;\n]+?)(\s*(;.*)?)$")
    for line in open(sys.argv[1]):
        m = pat.match(line.rstrip("\n"))
        if not m:
            out.append(line); continue
        ind, op, d, a, b, tail = m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6) or ""
        if not (operand_ok(a) and operand_ok(b)):
            n["kept"] += 1; out.append(line); continue
        if op == "fmac":
            out.append(f"{ind}v_fma_f32 {d}, {a}, {b}, {d}{tail}\n")
        else:
            out.append(f"{ind}v_{op}_f32_e64 {d}, {a}, {b}{tail}\n")
        n[op] += 1
    open(sys.argv[2], "w").writelines(out)
    print("vop3_rewrite:", n, file=sys.stderr)


if __name__ == "__main__":
    main()
