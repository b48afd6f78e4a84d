#!/bin/bash
# Runs ON the GPU box: scalar / vector instruction counts per wave of single-view launches (the first view on a fresh
# grid, then views over the carved grid), default or tsdf.   profiles/tools/pmc_single_view.sh <out dir under gpurun_out> [mode]
set -u
OUT=$1; MODE=${2:-tsdf}; REPO=$(pwd -P); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd); export TMPDIR=/tmp
cat > /tmp/sv_once.py <<PY
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
for i in range(4):
    assert c.CarveDevice(views[i], d)
c.sync()
PY
( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA -d "$OUT" -o sv --output-format csv -- python /tmp/sv_once.py ) > "$OUT/sv.log" 2>&1
python - "$OUT" "$MODE" <<'PY'
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "carve_fused" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Dispatch_Id"], collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
for k, (d, c) in enumerate(acc.items()):
    w = c["SQ_WAVES"] or 1
    print("%s view %d: waves %.0f  per wave: VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f | SALU + SMEM per CU %.2f M instructions; SQ_ACTIVE_INST_SCA %.3g"
          % (sys.argv[2], k, w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_SMEM"] / w, c["SQ_INSTS_LDS"] / w,
             (c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"]) / 256 / 1e6, c["SQ_ACTIVE_INST_SCA"]))
PY
