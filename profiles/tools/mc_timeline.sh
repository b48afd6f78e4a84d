#!/bin/bash
# Runs ON the GPU box: kernel timeline (start, duration, gap to the previous kernel) of the LAST marching-cubes
# extraction of a bench.py run, from a rocprofv3 kernel trace.   profiles/tools/mc_timeline.sh <out dir under gpurun_out> [lib]
R=$(pwd -P)
OUT=$R/$1
LIB=${2:-$R/vacancy_amd/csrc/libvacancy_hip.so}
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
VCY_HIP_LIB=$LIB rocprofv3 --kernel-trace --stats -d "$OUT" -o t --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > "$OUT/bench.log" 2>&1
python - "$OUT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [r for r in rows if re.search(r"mc_|scan_chunks|add_chunk", r["Kernel_Name"])]
idx = max(i for i, r in enumerate(seq) if "mc_bits" in r["Kernel_Name"])
seq = seq[idx:]
t0 = int(seq[0]["Start_Timestamp"]); prev = t0
for r in seq:
    n = re.search(r"(mc_\w+|scan_chunks\w*|add_chunk\w*)", r["Kernel_Name"]).group(1)
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-26s start %8.1f us  dur %7.1f us  gap before %6.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3)); prev = e
print("span %.1f us" % ((prev - t0) / 1e3))
PY
