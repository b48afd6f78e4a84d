#!/bin/bash
# Runs ON the GPU box: WRITE_SIZE / FETCH_SIZE of single-view TSDF launches at 1024^3 for several builds (prod = in-tree),
# and their ms per view.  usage: profiles/tools/pmc_write_variants.sh <out dir> <variant> ...
set -u
OUT=$1; shift
REPO=$(pwd -P); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd); export TMPDIR=/tmp
cat > /tmp/pv_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 5
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1)
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, 32, 1280, 720)
sdf0 = vc.make_sdf(masks[0], use_truncation=True, band=0.1)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(sdf0)
c.set_param("defer", 0)
for i in range(nv):
    assert c.CarveDevice(views[i], d)
c.sync()
PY
for v in "$@"; do
  lib=$REPO/build/variants/$v/libvacancy_hip.so; [ "$v" = prod ] && lib=$REPO/vacancy_amd/csrc/libvacancy_hip.so
  for ctr in WRITE_SIZE FETCH_SIZE; do
    ( cd /tmp && VCY_HIP_LIB=$lib rocprofv3 --pmc $ctr -d "$OUT" -o ${v}_$ctr --output-format csv -- python /tmp/pv_once.py ) > "$OUT/${v}_$ctr.log" 2>&1
  done
  python - "$OUT" $v <<'PY'
import csv, sys, collections
out, v = sys.argv[1], sys.argv[2]
for ctr in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open("%s/%s_%s_counter_collection.csv" % (out, v, ctr))):
        if "carve_fused" in r["Kernel_Name"]:
            acc.setdefault(r["Dispatch_Id"], 0.0); acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(acc.values())
    print("%-8s %s KiB per launch (launches 2..): %s" % (v, ctr, ", ".join("%.4g" % x for x in vals[1:])))
PY
  VCY_HIP_LIB=$lib python $REPO/profiles/tools/per_view.py 1024 tsdf | head -2
done
