#!/bin/bash
# Runs ON the GPU box: SQ counters of the marching-cubes kernels at 1024^3 (own rocprofv3 pass per counter set).
R=$(pwd -P); O=$R/gpurun_out/mcpmc; mkdir -p $O
cat > /tmp/mc_drive.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["VCY_ROOT"])
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, nv = 1024, 32
opt = synth.sphere_option(n, UpdateOption())
views, masks = synth.sphere_views(n, nv, 1280, 720)
c = vc.VoxelCarver(opt)
assert c.Init()
d = [c.upload_sdf(vc.make_sdf(masks[0]))] * nv
assert c.CarveBatchDevice(vc.VoxelCarver.prepare_batch(views, d))
c.set_param("meshkeys", 0)
for it in range(3):
    m = c.ExtractIsoSurface(0.0, True)
PY
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  VCY_ROOT=$R rocprofv3 --pmc $set -d $O -o set$i --output-format csv -- python /tmp/mc_drive.py > $O/set$i.log 2>&1
done
VCY_ROOT=$R rocprofv3 --kernel-trace --stats -d $O -o trace --output-format csv -- python /tmp/mc_drive.py > $O/trace.log 2>&1
python - $O <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/set*_counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(mc_\w+?)_kernel", r["Kernel_Name"])
        if m: per[(m.group(1), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, d), v in per.items(): acc[k][c].append(v)
dur = {}
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/trace_kernel_stats.csv")[0])):
    m = re.search(r"(mc_\w+?)_kernel", r["Name"])
    if m: dur[m.group(1)] = float(r["AverageNs"])
for k in sorted(acc):
    a = {c: sum(v) / len(v) for c, v in acc[k].items()}
    t = dur.get(k, 0) * 1e-9
    clk = a.get("GRBM_GUI_ACTIVE", 0) / 8 / t if t else 0
    line = "%-12s %7.1f us  clk %.2f GHz" % (k, t * 1e6, clk / 1e9)
    if a.get("SQ_WAVE_CYCLES") and clk: line += "  waves/SIMD %.2f" % (a["SQ_WAVE_CYCLES"] * 4 / (1024 * clk * t))
    if a.get("SQ_WAIT_INST_ANY"): line += "  wait %.2f" % (a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"])
    line += "  per wave: " + " ".join("%s %.1f" % (c.replace("SQ_INSTS_", "").lower(), a[c] / a["SQ_WAVES"]) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if c in a)
    line += "  waves %.0f" % a.get("SQ_WAVES", 0)
    print(line)
PY
