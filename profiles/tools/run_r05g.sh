#!/bin/bash
# Runs ON the GPU box: quick check of a marching-cubes change -- its GPU tests, a short fuzz, per-kernel times, wall.
set -u
O=gpurun_out/r05g; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q -k "march or mesh or slab or halo or bunny or sweep or extract" ) > $O/pytest_mc.log 2>&1; echo "pytest_mc rc=$?" > $O/status.txt
( time timeout 600 python tests/fuzz/fuzz_marching_cubes.py 0 ${FUZZ:-150} ) > $O/fuzz_mc.log 2>&1; echo "fuzz_mc rc=$?" >> $O/status.txt
bash profiles/tools/mc_kernel_times.sh prod > $O/mc_kernels_1024.txt 2>&1
N=512 NV=16 MODE=tsdf bash profiles/tools/mc_kernel_times.sh prod > $O/mc_kernels_512.txt 2>&1
bash profiles/tools/ab_mc.sh prod > $O/mc_wall_1024.txt 2>&1
cat $O/status.txt; tail -2 $O/pytest_mc.log; tail -2 $O/fuzz_mc.log; cat $O/mc_kernels_1024.txt $O/mc_kernels_512.txt $O/mc_wall_1024.txt
