#!/usr/bin/env python
"""Static VALU instruction mix of a loop body in a hipcc -S listing, priced with the measured SIMD issue cost per
instruction class (tools/microbench/valu_issue.hip -> profiles/r02_valu_issue.md).

usage: python tools/isa_mix.py FILE.s KERNEL_SUBSTRING FIRST_LABEL LAST_LABEL
       (the body = the lines from FIRST_LABEL up to LAST_LABEL inside the kernel whose mangled name contains the substring)
prints {"classes": {class: count}, "instructions": n, "cycles": sum, "avg_cycles": ...} as JSON."""
import json
import re
import sys

# cycles one wave64 instruction occupies its SIMD at saturation (8 waves/SIMD launched), MI355X, measured
COST = {"fp32_basic": 2.13,      # v_add/sub/mul_f32, v_fma_f32 with <= 2 distinct VGPR sources
        "fp32_3src": 3.8,        # v_fma_f32 / v_fmac_f32 reading three VGPRs
        "packed": 4.17,          # v_pk_add/mul/fma_f32
        "dpp": 4.11,             # any VALU op with a DPP modifier
        "cmp": 4.1,              # v_cmp_* (to vcc or an SGPR pair)
        "other_valu": 4.1,       # v_min/max/med3, v_cndmask, v_mov_b64, v_cvt, v_readlane/readfirstlane, integer VALU, v_mad_u64_u32
        "mov32": 2.13,           # v_mov_b32 (assumed basic rate)
        "transcendental": 8.1,   # v_exp/log/rcp/rsq/sqrt_f32
        "permlane": 8.1}         # v_permlane32_swap / v_permlane16_swap


def classify(op, line):
    if "dpp" in op or "row_ror" in line or "quad_perm" in line or "row_shr" in line or "row_bcast" in line:
        return "dpp"
    if op.startswith("v_pk_"):
        return "packed"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_permlane"):
        return "permlane"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt)_f32", op):
        return "transcendental"
    if re.match(r"v_(fma|fmac|mad)_f32", op):
        regs = set(re.findall(r"\bv\d+\b", line.split(None, 1)[1])) if len(line.split(None, 1)) > 1 else set()
        dst = re.findall(r"\bv\d+\b", line)[:1]
        srcs = re.findall(r"\bv\d+\b", line)[1:]
        n = len(set(srcs)) + (1 if op.startswith("v_fmac") and dst and dst[0] not in srcs else 0)
        return "fp32_3src" if n >= 3 else "fp32_basic"
    if re.match(r"v_(add|sub|subrev|mul)_f32", op):
        return "fp32_basic"
    if op.startswith("v_mov_b32"):
        return "mov32"
    return "other_valu"


def main():
    path, kern, first, last = sys.argv[1:5]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().endswith(":") or (l.startswith("_Z") and kern in l and ":" in l))
    body, on = [], False
    for l in lines[start:]:
        if l.startswith(first + ":"):
            on = True
        elif l.startswith(last + ":") and on:
            break
        elif l.startswith("\t.end_amdhsa_kernel") or l.strip() == "s_endpgm":
            if on:
                break
        if on:
            body.append(l)
    classes, salu, lds = {}, 0, 0
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c = classify(op, t)
            classes[c] = classes.get(c, 0) + 1
        elif op.startswith("s_"):
            salu += 1
        elif op.startswith("ds_"):
            lds += 1
    n = sum(classes.values())
    cyc = sum(COST[c] * k for c, k in classes.items())
    print(json.dumps({"classes": classes, "instructions": n, "cycles": round(cyc, 1), "avg_cycles": round(cyc / max(n, 1), 3),
                      "salu": salu, "lds": lds, "cost_table": COST}))


if __name__ == "__main__":
    main()
