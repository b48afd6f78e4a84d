#!/bin/bash
# Runs ON the GPU box (round 5): the default bench line with the `configs` block; counters of the single-view launches
# (weighted average over a carved grid, u8 counters) and the first-view tool.
set -u
O=gpurun_out/r05e; mkdir -p $O
( time timeout 900 python bench.py --no-cpu-baseline --variants none ) > $O/bench_default_configs.json 2> $O/bench_default_configs.err; echo "bench rc=$?" > $O/status.txt
timeout 600 python profiles/tools/first_view.py 1024 default > $O/first_view.txt 2>&1
timeout 600 python profiles/tools/first_view.py 1024 tsdf >> $O/first_view.txt 2>&1
timeout 900 bash profiles/tools/pmc_per_view.sh gpurun_out/r05e/pv_tsdf 1024 tsdf > /dev/null 2>&1; cp $O/pv_tsdf/summary.txt $O/per_view_tsdf_pmc.txt
cat $O/status.txt; cat $O/first_view.txt; cat $O/per_view_tsdf_pmc.txt; tail -3 $O/bench_default_configs.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05e/bench_default_configs.json').read().strip().splitlines()[0])
print(json.dumps(d.get('configs'), indent=1)[:6000])
PY
rm -rf $O/pv_tsdf
