mkdir -p gpurun_out/r06k
SECONDS=0
python bench.py > gpurun_out/r06k/bench.json 2> gpurun_out/r06k/bench.err
echo "bench.py wall seconds: $SECONDS rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06k/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(json.dumps(d["cpu_baseline"])[:1200])
PY
