#!/bin/bash
# Runs ON the GPU box: bench.py (default, cull0, tsdf) for each build under build/variants/ named on the command line.
#   profiles/tools/ab_variants.sh <out_dir under gpurun_out> <variant>...
OUT=$1; shift
mkdir -p "$OUT"
for v in "$@"; do
  lib=build/variants/$v/libvacancy_hip.so
  [ "$v" = "prod" ] && lib=vacancy_amd/csrc/libvacancy_hip.so
  echo -n "$v  state: "; VCY_HIP_LIB=$lib python profiles/tools/state_hash.py 2>&1 | tail -1
  VCY_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-mc --no-configs --variants modes > "$OUT/$v.json" 2> "$OUT/$v.err"
  python - "$v" "$OUT/$v.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-10s default %9.0f (%.3f ms)  cull0 %9.0f  tsdf %9.0f" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("value_cull0") or 0, r.get("value_tsdf") or 0))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
