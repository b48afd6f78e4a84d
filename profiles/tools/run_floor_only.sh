#!/bin/bash
# Runs ON the GPU box: the issue-floor A/B of run_round_profiles.sh by itself (profiles/counters.json of the round is in
# the tree), then the headline bench line again with it, and the write ceiling of the device.
set -u
RND=${1:-r05}
O=gpurun_out/$RND; mkdir -p $O
cp profiles/counters.json $O/counters.json
profiles/tools/ab_variants.sh "$O/floor" devprod floorT floorS floorTS devprod floorT floorS floorTS > "$O/issue_floor.txt" 2>&1
python profiles/tools/summarize_floor.py "$O/issue_floor.txt" "$O/counters.json"
cp $O/counters.json profiles/counters.json
python bench.py > "$O/bench_1024x32_default.json" 2> "$O/bench_default.err"; echo "bench default rc=$?"
python profiles/tools/write_ceiling.py > "$O/write_ceiling.txt" 2>&1
cat $O/issue_floor.txt | grep -v state; cat $O/write_ceiling.txt
