#!/usr/bin/env python3
"""Opcode histogram of the dominant kernel, whole kernel and select-free view loop, weighted with the issue
costs measured on MI355X (profiles/r02/valu_ubench.txt).

  profiles/tools/isa_histogram.py [update-mode]     (0 = kMax default kernel, 2 = unit-weight average + truncation)

Compiles vacancy_amd/csrc/carve_fused.hip to assembly (device only, the instantiations bench.py launches),
takes carve_fused_kernel<unsigned short, MODE, TRUNC, true, false, 16, false, 2> and prints
  * every opcode with its static count and issue class,
  * the same for every flavour of the select-free run: the straight-line code over a lane's eight voxels in
    one view (eight exact divisions, sixteen tap reads), i.e. instructions and issue cycles per 8 voxel*views
    of a wave, without the per-view overhead at the top of the view loop.
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
                        "-DVCY_DEV_BENCH_KERNELS_ONLY", "--cuda-device-only", "-S", os.path.join(src, "carve_fused_u8.hip"),
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
    # the straight-line runs over a lane's eight voxels: basic blocks (no label inside) with eight exact
    # divisions (v_rcp_f32) and sixteen tap reads (ds_read2_b32): one per flavour of the select-free loop
    blocks, cur = [], []
    for l in body:
        t = l.strip()
        if re.match(r"^\.?[A-Za-z_][\w.$]*:", t):  # a label (possibly followed by a comment): a new basic block
            blocks.append(cur)
            cur = []
        else:
            cur.append(l)
    blocks.append(cur)
    for b in blocks:
        nrcp = sum("v_rcp_f32" in l for l in b)
        ntap = sum("ds_read2_b32" in l for l in b)
        if nrcp < 8 or ntap < 16:
            continue
        sig = []
        if any("v_cmp_gt_f32_e32 vcc" in l for l in b):
            sig.append("kMax update chain (brick touched everywhere)")
        if any("v_cmp_ngt_f32_e32 vcc" in l for l in b):
            sig.append("weighted average with the truncation test")
        if nrcp == 9:
            sig.append("weighted average, brick-wide weights (one count reciprocal per view)")
        if nrcp == 16 and not any("v_cmp_ngt_f32_e32 vcc" in l for l in b):
            sig.append("weighted average, per-voxel weights, truncation test compiled out")
        if nrcp == 8 and not sig:
            sig.append("first touch: plain store")
        valu, cyc = histogram(b, "run over the 8 voxels of a lane, one view -- " + "; ".join(sig))
        print("   => %.1f VALU instructions and %.1f issue cycles per voxel*view of a wave (per-view overhead not included)"
              % (valu / 8.0, cyc / 8.0))


if __name__ == "__main__":
    main()
