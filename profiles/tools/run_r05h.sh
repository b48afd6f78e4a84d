#!/bin/bash
# Runs ON the GPU box: whole GPU suite + the default bench line (with the configs block).
set -u
O=gpurun_out/${1:-r05h}; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" > $O/status.txt
( time timeout 900 python bench.py ) > $O/bench_1024x32_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; tail -3 $O/pytest_gpu.log; tail -4 $O/bench_default.err
python - $O/bench_1024x32_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['mc']['device_ms'], d['mc']['wall_ms'])
c=d.get('configs',{})
for k,v in c.items():
    if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','sequence_wall_ms','carve_wall_ms','extract_voxel_wall_ms','mc_wall_ms','mc','error','wall_s_spent')}, (v.get('roofline') or {}).get('frac'))
print('configs wall', c.get('wall_s_spent'))
PY
