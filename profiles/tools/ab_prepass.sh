#!/bin/bash
# Runs ON the GPU box: bench.py for each build named; prints value, step, pre-pass and kernel ms (default / cull0 / tsdf).
for v in "$@"; do
  lib=build/variants/$v/libvacancy_hip.so
  [ "$v" = "prod" ] && lib=vacancy_amd/csrc/libvacancy_hip.so
  VCY_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-mc --variants modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s default %8.0f step %.3f prepass %.3f kernel %.3f   cull0 %8.0f  tsdf %8.0f' % ('$v', d['value'], r['step_device_ms'], r['prepass_ms_per_step'], r['avg_launch_ms'], r['value_cull0'], r['value_tsdf']))"
done
