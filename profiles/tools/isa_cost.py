#!/usr/bin/env python3
"""Static issue-cost estimate of a basic-block range of a gfx950 .s file, using the per-instruction
SIMD cycles measured by profiles/tools/valu_ubench.hip on MI355X (profiles/r02/valu_ubench.txt):
  full rate (2 cycles / wave64): v_add/sub/mul/fma/fmac_f32, v_mov_b32, v_add/sub_u32, v_and/or/xor_b32
                                 -- only with VGPR / inline-constant / literal sources
  half rate (4 cycles):          everything else on the VALU, and ANY VALU instruction with an SGPR source
                                 (v_pk_*_f32 do two operations in those 4 cycles)
  quarter rate (8 cycles):       v_rcp_f32 and the other transcendentals
usage: isa_cost.py file.s first_line last_line
"""
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32",
         "v_rcp_iflag_f32"}


def base(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def classify(line):
    parts = line.strip().split(None, 1)
    op = parts[0]
    args = parts[1] if len(parts) > 1 else ""
    args = args.split(";")[0]
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        return None
    b = base(op)
    if b in TRANS:
        return "trans"
    sgpr_src = False
    ops = [a.strip() for a in args.split(",")]
    for a in ops[1:]:  # sources
        if re.match(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp)", a):
            sgpr_src = True
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "half"
    if b in FAST and not sgpr_src:
        return "full"
    return "half"


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    cost = {"full": 2, "half": 4, "trans": 8}
    n = {}
    ops = {}
    for i, line in enumerate(open(path), 1):
        if i < a or i > b:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c = classify(t)
        if c is None:
            continue
        n[c] = n.get(c, 0) + 1
        key = (c, base(t.split()[0]))
        ops[key] = ops.get(key, 0) + 1
    valu = sum(n.get(k, 0) for k in cost)
    cyc = sum(n.get(k, 0) * v for k, v in cost.items())
    print("lines %d-%d: VALU %d instructions = %d SIMD issue cycles (%.2f cycles/instruction); %s" %
          (a, b, valu, cyc, cyc / max(valu, 1), ", ".join("%s %d" % kv for kv in sorted(n.items()))))
    for (c, op), k in sorted(ops.items(), key=lambda kv: (-cost.get(kv[0][0], 0) * kv[1], kv[0])):
        if c in cost:
            print("  %-6s %-22s %4d  -> %5d cycles" % (c, op, k, k * cost[c]))


if __name__ == "__main__":
    main()
