#!/bin/bash
# Runs ON the GPU box: the round's profile set, but only on a box of the pool's usual speed (the boxes differ by +-3 % in
# shader clock; one of the slow ones would put every number of the set 4 % below what the driver's run is likely to see).
#   profiles/tools/collect_on_typical_box.sh <round> [minimum headline Mvoxel*views/s, default 4400000]
R=$1; MIN=${2:-4400000}
V=$(python bench.py --no-variants --no-mc --no-cpu-baseline 2>/dev/null | python -c "import sys, json; print(int(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']))")
echo "probe: headline $V Mvoxel*views/s (threshold $MIN)"
if [ "${V:-0}" -lt "$MIN" ]; then echo "slow box: not collecting"; exit 7; fi
bash profiles/tools/run_round_profiles.sh $R > gpurun_out/${R}_run.log 2>&1
cat gpurun_out/$R/status.txt | tr "\n" ";"
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error") | tee gpurun_out/$R/pytest_gpu_final.txt
