#!/bin/bash
# Runs ON the GPU box: counters of single-view launches over a carved grid (the reference's call pattern, "defer" 0).
#   profiles/tools/pmc_per_view.sh <out dir under gpurun_out> [n] [mode]
set -u
OUT=$1; N=${2:-1024}; MODE=${3:-tsdf}
REPO=$(pwd -P)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cat > /tmp/pv_once.py <<PY
import sys
sys.path.insert(0, "$REPO")
from vacancy_amd import synth
from vacancy_amd import carver as vc
from vacancy_amd.capi import UpdateOption
n, mode, nv = $N, "$MODE", 6
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
run() { local name=$1; shift; ( cd /tmp && rocprofv3 "$@" -d "$OUT" -o "$name" --output-format csv -- python /tmp/pv_once.py ) > "$OUT/$name.log" 2>&1; }
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES_EQ_64 SQ_INST_CYCLES_VMEM
run tcc1 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run tcc2 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcp1 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr
run grbm --pmc GRBM_GUI_ACTIVE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + "/*_counter_collection.csv")):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        key = (k, r["Counter_Name"])
        acc.setdefault(key, collections.OrderedDict()).setdefault(r["Dispatch_Id"], 0.0)
        acc[key][r["Dispatch_Id"]] += float(r["Counter_Value"])   # (one row per dimension of the counter)
    for (k, cn), disp in acc.items():
        res.setdefault(k, {})[cn] = list(disp.values())
with open(out + "/summary.txt", "w") as fo:
    for k, cs in res.items():
        if "carve_fused" not in k and "footprint" not in k: continue
        fo.write("== %s\n" % k)
        for cn, vals in cs.items():
            fo.write("  %-34s n=%d  last: %s\n" % (cn, len(vals), ", ".join("%.4g" % v for v in vals[-3:])))
print(open(out + "/summary.txt").read())
PY
tail -30 "$OUT"/trace_kernel_stats.csv 2>/dev/null | cut -c1-200
