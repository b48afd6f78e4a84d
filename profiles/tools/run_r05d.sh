#!/bin/bash
# Runs ON the GPU box (round 5, after the container was re-created): the whole GPU suite, the default bench line,
# config 4 with the footprints in the prologue vs records, the streamed emulation.
set -u
O=gpurun_out/r05d; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" > $O/status.txt
timeout 900 python bench.py > $O/bench_1024x32_default.json 2> $O/bench_default.err; echo "default rc=$?" >> $O/status.txt
timeout 900 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --variants streamed > $O/bench_2048x64_config4.json 2> $O/bench_2048x64_config4.err; echo "config4 rc=$?" >> $O/status.txt
timeout 900 python bench.py --prologue 2 --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-mc > $O/bench_2048x64_config4_records.json 2> $O/bench_2048x64_config4_records.err; echo "config4 records rc=$?" >> $O/status.txt
timeout 900 python bench.py --config 1 --no-cpu-baseline > $O/bench_512x16_tsdf_config1.json 2> $O/bench_config1.err; echo "config1 rc=$?" >> $O/status.txt
timeout 1200 python profiles/tools/streamed_emulation.py > $O/streamed_emulation.txt 2>&1; echo "streamed emulation rc=$?" >> $O/status.txt
cat $O/status.txt; tail -5 $O/pytest_gpu.log; cat $O/streamed_emulation.txt | tail -30
for f in $O/bench_1024x32_default.json $O/bench_2048x64_config4.json $O/bench_2048x64_config4_records.json $O/bench_512x16_tsdf_config1.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1], d['value'], d['ms_per_step'], 'launches', r.get('kernel_launches_per_step'), 'prepass', r.get('prepass_ms_per_step'), 'kernel', r.get('avg_launch_ms'), 'mc', d.get('mc'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
