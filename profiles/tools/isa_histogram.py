#!/usr/bin/env python3
"""Opcode histogram of the dominant kernel, whole kernel and select-free view loop, weighted with the issue
costs measured on MI355X (profiles/r02/valu_ubench.txt).

  profiles/tools/isa_histogram.py [update-mode]     (0 = kMax default kernel, 2 = unit-weight average + truncation)

Compiles vacancy_amd/csrc/carve_fused.hip to assembly (device only, the instantiations bench.py launches),
takes carve_fused_kernel<unsigned short, MODE, TRUNC, true, false, 16, false, 2> and prints
  * every opcode with its static count and issue class,
  * the same for ONE view of the select-free path: from the top of the view loop (tile wait, request of the
    next tile, per-view constants) through the straight-line run over a lane's eight voxels to the last
    update, i.e. instructions and issue cycles per 8 voxel*views of a wave.
Classes: full = 2 cycles per wave64 instruction (1.95 measured), half = 4 (3.5 measured; also ANY VALU
instruction with an SGPR source when it follows another half-rate one), trans = 8 (7.55 measured).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_cost  # noqa: E402

COST = {"full": 1.95, "half": 3.5, "trans": 7.55}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "0"
    trunc = "1" if mode == "2" else "0"
    src = os.path.join(ROOT, "vacancy_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
                        "-I" + os.path.join(ROOT, "include"), "-I" + src, "-fno-slp-vectorize",
                        "-DVCY_DEV_BENCH_KERNELS_ONLY", "--cuda-device-only", "-S", os.path.join(src, "carve_fused.hip"),
                        "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    tag = "carve_fused_kernelItLi%sELb%sELb1ELb0ELi16ELb0ELi2EEEv" % (mode, trunc)
    start = next(i for i, l in enumerate(lines) if tag in l and l.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]

    def histogram(block, title):
        ops = collections.Counter()
        cls = collections.Counter()
        for l in block:
            t = l.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            c = isa_cost.classify(t)
            if c is None:
                continue
            op = t.split()[0]
            ops[(c, op)] += 1
            cls[c] += 1
        print("== %s" % title)
        valu = sum(cls[k] for k in ("full", "half", "trans"))
        cyc = sum(cls[k] * COST[k] for k in ("full", "half", "trans"))
        print("   VALU %d (full %d, half %d, trans %d) = %.0f issue cycles; SALU %d, LDS %d, VMEM %d"
              % (valu, cls["full"], cls["half"], cls["trans"], cyc, cls["salu"], cls["lds"], cls["vmem"]))
        for (c, op), k in sorted(ops.items(), key=lambda kv: (-kv[1], kv[0])):
            print("   %6d  %-6s %s" % (k, c, op))
        return valu, cyc

    histogram(body, "whole kernel %s (static counts, %d lines)" % (tag, len(body)))
    # select-free loop: the longest stretch that holds eight update chains (kMax: v_cmp_gt_f32_e32 vcc;
    # average: v_cmp_ngt_f32_e32 vcc) -- from the first v_rcp_f32 before the first of them to the last one
    pat = "v_cmp_gt_f32_e32 vcc" if mode == "0" else "v_cmp_ngt_f32_e32 vcc"
    idx = [i for i, l in enumerate(body) if pat in l]
    groups, cur = [], []
    for i in idx:
        if cur and i - cur[-1] > 400:
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    g = next(x for x in groups if len(x) >= 8)[:8]
    first = max(i for i in range(g[0]) if "v_rcp_f32" in body[i] and sum("v_rcp_f32" in body[j] for j in range(i, g[0])) >= 1)
    # walk back over the run's earlier reciprocals (straight-line code: no label in between)
    i = first
    while i > 0 and not body[i].strip().endswith(":"):
        i -= 1
    last = g[-1]
    while "v_addc_co_u32" not in body[last] and "v_cndmask" not in body[last + 1] and last < len(body) - 1 and mode == "0":
        last += 1
    last += 3 if mode != "0" else 1
    valu, cyc = histogram(body[i + 1:last + 1], "one view of the select-free path: loop top (tile wait, next tile request, per-view constants) + the run over the 8 voxels of a lane (lines %d-%d of the kernel)" % (i + 1, last))
    print("   => %.1f VALU instructions and %.1f issue cycles per voxel*view of a wave" % (valu / 8.0, cyc / 8.0))


if __name__ == "__main__":
    main()
