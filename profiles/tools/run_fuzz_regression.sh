#!/bin/bash
# Runs ON the GPU box: the differential fuzzers on the build in the tree (state and meshes against the CPU oracle, bit for
# bit).  usage: profiles/tools/run_fuzz_regression.sh <out dir under gpurun_out> [budget seconds per fuzzer]
set -u
O=gpurun_out/${1:-fuzz}; B=${2:-900}; mkdir -p $O
python -c "from vacancy_amd import capi; print(capi.load().vcy_version().decode())" > $O/fuzz_regression.txt
run() {  # label, seconds, command...
  local label=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout $secs "$@" > $O/fuzz_last.log 2>&1; local rc=$?
  echo "$label  rc=$rc ($(( $(date +%s) - t0 )) s; rc 124 = stopped by the time budget)  $(grep -iE 'mismatch|seeds|bad' $O/fuzz_last.log | tail -1)" >> $O/fuzz_regression.txt
  tail -2 $O/fuzz_last.log >> $O/fuzz_regression.txt
}
run "tests/fuzz/fuzz_marching_cubes.py 0..2300" $B python tests/fuzz/fuzz_marching_cubes.py 0 2300
run "tests/fuzz/fuzz_incremental.py 2000..3200 (rows of whole bricks)" $B python tests/fuzz/fuzz_incremental.py 2000 3200
run "tests/fuzz/fuzz_incremental.py 0..1200" $B python tests/fuzz/fuzz_incremental.py 0 1200
run "tests/fuzz/fuzz_incremental.py 3000..4200 modes (nearest neighbour, weights != 1, update limits in reach)" $B python tests/fuzz/fuzz_incremental.py 3000 4200 modes
run "tests/fuzz/fuzz_random_scenes.py 0..4000" $B python tests/fuzz/fuzz_random_scenes.py 0 4000
run "tests/fuzz/fuzz_fine_grids.py 0..1500" $B python tests/fuzz/fuzz_fine_grids.py 0 1500
cat $O/fuzz_regression.txt
