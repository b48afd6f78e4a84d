#!/bin/bash
# Runs ON the GPU box: kernel trace of single-view launches over a carved grid ("defer" 0), per kernel avg us of views 2..N.
#   profiles/tools/per_view_trace.sh <out dir> [n] [mode] [views]
set -u
OUT=$1; N=${2:-1024}; MODE=${3:-default}; NV=${4:-12}
REPO=$(pwd -P); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd); export TMPDIR=/tmp
cat > /tmp/pv_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, mode, nv = $N, "$MODE", $NV
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if mode == "tsdf" else UpdateOption()
opt = synth.sphere_option(n, uo)
views, masks = synth.sphere_views(n, 32, 1280, 720)
sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
c = vc.VoxelCarver(opt); assert c.Init()
d = c.upload_sdf(sdf0)
c.set_param("defer", 0)
for i in range(nv):
    assert c.CarveDevice(views[i], d)
c.sync()
PY
( cd /tmp && rocprofv3 --kernel-trace -d "$OUT" -o trace --output-format csv -- python /tmp/pv_once.py ) > "$OUT/trace.log" 2>&1
python - "$OUT/trace_kernel_trace.csv" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.OrderedDict()
seen = collections.Counter()
for r in rows:
    m = re.search(r"(\w+_kernel|__amd_\w+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    seen[k] += 1
    if k.startswith("carve_fused") and seen[k] == 1: first_end = int(r["End_Timestamp"])
    per.setdefault(k, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in per.items():
    later = [e - s for s, e in v if s > first_end]
    if later: print("%-28s launches %3d  avg %8.1f us  min %8.1f  max %8.1f" % (k, len(later), sum(later) / len(later) / 1e3, min(later) / 1e3, max(later) / 1e3))
cf = [x for x in per.get("carve_fused_kernel", []) if x[0] > first_end]
if len(cf) > 1: print("period between carve kernel starts: %.1f us" % ((cf[-1][0] - cf[0][0]) / (len(cf) - 1) / 1e3))
PY
