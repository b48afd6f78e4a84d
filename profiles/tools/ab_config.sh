#!/bin/bash
# Runs ON the GPU box: one bench.py configuration for each build named on the command line.
#   profiles/tools/ab_config.sh "<bench args>" <variant>...
ARGS=$1; shift
for v in "$@"; do
  lib=build/variants/$v/libvacancy_hip.so
  [ "$v" = "prod" ] && lib=vacancy_amd/csrc/libvacancy_hip.so
  VCY_HIP_LIB=$lib python bench.py $ARGS --no-cpu-baseline --no-variants --no-mc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s %10.0f  %.3f ms' % ('$v', d['value'], d['ms_per_step']))"
done
