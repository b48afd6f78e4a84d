#!/bin/bash
# Runs ON the GPU box (round 5): the new GPU tests, config 4 with the footprints in the carve kernel's prologue vs records, streamed emulation
set -u
O=gpurun_out/r05c; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded_silhouette or make_sdf_batch or cpp_ or select_free or view_dropping" --durations=10 ) > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" > $O/status.txt
( time timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=10 ) > $O/pytest_fullsize.log 2>&1; echo "pytest_fullsize rc=$?" >> $O/status.txt
timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --variants streamed > $O/bench_2048x64_config4.json 2> $O/bench_2048x64_config4.err; echo "config4 rc=$?" >> $O/status.txt
timeout 900 python bench.py --prologue 2 --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-mc > $O/bench_2048x64_config4_records.json 2> $O/bench_2048x64_config4_records.err; echo "config4 records rc=$?" >> $O/status.txt
timeout 600 python bench.py --prologue 1 --steps 10 --no-cpu-baseline --no-variants --no-mc > $O/bench_1024x32_prologue1.json 2> $O/err1.txt; echo "1024 prologue1 rc=$?" >> $O/status.txt
timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-variants --no-mc > $O/bench_1024x32_prologue0.json 2> $O/err0.txt; echo "1024 prologue0 rc=$?" >> $O/status.txt
timeout 1200 python profiles/tools/streamed_emulation.py > $O/streamed_emulation.txt 2>&1; echo "streamed emulation rc=$?" >> $O/status.txt
cat $O/status.txt; tail -5 $O/pytest_new.log; tail -3 $O/pytest_fullsize.log; cat $O/streamed_emulation.txt
for f in $O/bench_2048x64_config4.json $O/bench_2048x64_config4_records.json $O/bench_1024x32_prologue1.json $O/bench_1024x32_prologue0.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1], d['value'], d['ms_per_step'], 'launches', r.get('kernel_launches_per_step'), 'prepass', r.get('prepass_ms_per_step'), 'kernel', r.get('avg_launch_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
