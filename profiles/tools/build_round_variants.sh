#!/bin/bash
# Build machine (no GPU needed): the development builds profiles/tools/run_round_profiles.sh uses, into build/variants/
# (git-ignored; they travel to the GPU box with the snapshot).
#   devprod  the kernels bench.py launches, as shipped (only those instantiations: -DVCY_DEV_BENCH_KERNELS_ONLY)
#   floorT   ... without the tile loads, floorS without the stores, floorTS without both (the measured issue floor)
#   phase    s_memtime marks per phase of the fused kernel (profiles/tools/phase_timing.py)
set -eu
cd "$(dirname "$0")/../.."
profiles/tools/build_variant.sh devprod -DVCY_DEV_BENCH_KERNELS_ONLY
profiles/tools/build_variant.sh floorT -DVCY_DEV_BENCH_KERNELS_ONLY -DVCY_FLOOR_NO_TILE_LOADS
profiles/tools/build_variant.sh floorS -DVCY_DEV_BENCH_KERNELS_ONLY -DVCY_FLOOR_NO_STORES
profiles/tools/build_variant.sh floorTS -DVCY_DEV_BENCH_KERNELS_ONLY -DVCY_FLOOR_NO_TILE_LOADS -DVCY_FLOOR_NO_STORES
profiles/tools/build_variant.sh phase -DVCY_DEV_BENCH_KERNELS_ONLY -DVCY_PHASE_TIMING
ls -la build/variants/*/libvacancy_hip.so
