#!/bin/bash
# Runs ON the GPU box: the marching-cubes tests, then the sweep A/B at 1024^3.
O=gpurun_out/mcs; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -k "marching or slab or rccl or cpp_bunny or queued_views or odd_grid or degenerate" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
timeout 300 profiles/tools/ab_mc_sweep.sh 2>&1 | tee $O/ab.txt
