#!/bin/bash
# Runs ON the GPU box (round 5): marching cubes with the single-pass prefix sums (look-back): the whole GPU suite, the MC
# fuzzer, per-kernel times at 1024^3 and 512^3, wall times.
set -u
O=gpurun_out/r05f; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" > $O/status.txt
( time timeout 900 python tests/fuzz/fuzz_marching_cubes.py 0 400 ) > $O/fuzz_mc.log 2>&1; echo "fuzz_mc rc=$?" >> $O/status.txt
bash profiles/tools/mc_kernel_times.sh prod > $O/mc_kernels_1024.txt 2>&1
N=512 NV=16 MODE=tsdf bash profiles/tools/mc_kernel_times.sh prod > $O/mc_kernels_512.txt 2>&1
bash profiles/tools/ab_mc.sh prod prod > $O/mc_wall_1024.txt 2>&1
cat $O/status.txt; tail -4 $O/pytest_gpu.log; tail -3 $O/fuzz_mc.log; cat $O/mc_kernels_1024.txt $O/mc_kernels_512.txt $O/mc_wall_1024.txt
