#!/bin/bash
# Runs ON the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of marching cubes at 1024^3 for each
# build named on the command line ("prod" = vacancy_amd/csrc/libvacancy_hip.so); MCSWEEP=1 for the one-sweep cell search;
# N=512 NV=16 MODE=tsdf for BASELINE configs[1].
R=$(pwd -P); O=$R/gpurun_out/mck; mkdir -p $O
cat > /tmp/mc_drive.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["VCY_ROOT"])
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, nv = int(os.environ.get("N", "1024")), int(os.environ.get("NV", "32"))
tsdf = os.environ.get("MODE", "default") == "tsdf"
uo = UpdateOption(voxel_update=1, use_truncation=True, truncation_band=0.1) if tsdf else UpdateOption()
opt = synth.sphere_option(n, uo)
w, h = (640, 480) if n <= 512 else (1280, 720)
views, masks = synth.sphere_views(n, nv, w, h)
sdf0 = vc.make_sdf(masks[0], use_truncation=bool(uo.use_truncation), band=uo.truncation_band)
c = vc.VoxelCarver(opt)
assert c.Init()
d = [c.upload_sdf(sdf0)] * nv
assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
c.set_param("meshkeys", 0)
c.set_param("mcsweep", int(os.environ.get("MCSWEEP", "0")))
for it in range(6):
    m = c.ExtractIsoSurface(0.0, True)
print("device_ms", m["device_ms"], "faces", len(m["faces"]))
PY
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/build/variants/$v/libvacancy_hip.so
  [ "$v" = "prod" ] && lib=$R/vacancy_amd/csrc/libvacancy_hip.so
  VCY_ROOT=$R VCY_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $O -o $v --output-format csv -- python /tmp/mc_drive.py > $O/$v.log 2>&1
  echo "== $v: $(grep device_ms $O/$v.log)"
  python - $O/${v}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "mc_" in n or "scan" in n or "chunk" in n:
        import re; m_ = re.search(r"(mc_\w+|scan_\w+|add_chunk\w*)", n); short = m_.group(1) if m_ else n[:28]; print("   %-28s calls %4s  avg %9.1f us  min %9.1f" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
