set -u
O=gpurun_out/r2g; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cd /tmp; export TMPDIR=/tmp
R=$(cd /root/repo && pwd -P)
rocprofv3 --kernel-trace --stats -d $R/$O -o mc_trace --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $R/$O/mc_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/$O -o mc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > $R/$O/mc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/$O -o mc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants > $R/$O/mc_write.log 2>&1
cd $R; cat $O/status.txt; tail -3 $O/pytest.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('mc'))
PY
