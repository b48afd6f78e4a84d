#!/bin/bash
# Build machine: development builds of the fused carve kernel that leave at points 1..5 of a wave's path
# (1 after the block decode, 2 after the prologue, 3 after the early-return test, 4 before the view loop, 5 after it);
# on the GPU box: profiles/tools/pmc_single_view.sh with each gives scalar / vector instructions per wave up to that point.
set -eu
cd "$(dirname "$0")/../.."
for k in 1 2 3 4 5; do profiles/tools/build_variant.sh exit$k -DVCY_DEV_BENCH_KERNELS_ONLY -DVCY_DEV_EXIT_AT=$k > /dev/null; done
ls build/variants/
