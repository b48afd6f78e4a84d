#!/bin/bash
# Runs ON the GPU box: the GPU test suite, the default bench line, and a kernel trace of the same command.
#   profiles/tools/run_round_check.sh <out dir under gpurun_out, e.g. r04check>
set -u
O=gpurun_out/${1:-check}; mkdir -p $O
R=$(pwd -P)
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $R/$O/trace.log 2>&1
cd $R; cat $O/status.txt; tail -3 $O/pytest.log; tail -c 3000 $O/bench.json
