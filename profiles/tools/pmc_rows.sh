#!/bin/bash
# Runs ON the GPU box: instruction counts and wait cycles per wave of single-view launches (the first view on a fresh grid,
# then three views over the carved grid) with the few-view flavour of the fused kernel or the workgroup-per-block one.
#   [VCY_HIP_LIB=...] [PARAM=rowkernel] profiles/tools/pmc_rows.sh <out dir under gpurun_out> <mode: tsdf|default> <value of the knob: -1|0|1>
set -u
OUT=$1; MODE=${2:-tsdf}; RK=${3:--1}; REPO=$(pwd -P); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd); export TMPDIR=/tmp
cat > /tmp/rows_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
mode = "$MODE"
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
views, masks = synth.sphere_views(1024, 32, 1280, 720)
c = vc.VoxelCarver(synth.sphere_option(1024, uo)); assert c.Init()
d = c.upload_sdf(vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band))
c.set_param("defer", 0)
c.set_param("${PARAM:-rowkernel}", $RK)
for i in range(4):
    assert c.CarveDevice(views[i], d)
c.sync()
PY
TAG=rows_${MODE}_${PARAM:-rowkernel}${RK}
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT" -o $TAG --output-format csv -- python /tmp/rows_once.py ) > "$OUT/$TAG.log" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT" -o ${TAG}_trace --output-format csv -- python /tmp/rows_once.py ) >> "$OUT/$TAG.log" 2>&1
python - "$OUT" "$MODE" "$RK" "$TAG" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/" + sys.argv[4] + "_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "carve_fused" not in r["Kernel_Name"]: continue
        acc.setdefault(int(r["Dispatch_Id"]), collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
dur = {}
for f in glob.glob(sys.argv[1] + "/**/" + sys.argv[4] + "_trace_kernel_trace.csv", recursive=True):
    k = 0
    for r in csv.DictReader(open(f)):
        if "carve_fused" in r["Kernel_Name"]:
            dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            k += 1
for k, d in enumerate(sorted(acc)):
    c = acc[d]
    w = c["SQ_WAVES"] or 1
    print("%s knob %s view %d: %.3f ms  waves %.0f | per wave: VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f | per CU: SALU+SMEM %.2f M, VALU %.2f M per SIMD | wave cycles %.3g, waiting %.2f of them"
          % (sys.argv[2], sys.argv[3], k, dur.get(k, 0.0), w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_SMEM"] / w, c["SQ_INSTS_LDS"] / w,
             (c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"]) / 256 / 1e6, c["SQ_INSTS_VALU"] / 1024 / 1e6, c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1)))
PY
