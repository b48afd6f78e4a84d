#!/bin/bash
# Build machine: copies the summaries run_round_profiles.sh left under gpurun_out/<round>/ into profiles/<round>/ (the
# tracked copies the docs cite) and installs its counters.json.   profiles/tools/copy_round_profiles.sh r06
set -eu
R=$1; S=gpurun_out/$R; D=profiles/$R; mkdir -p $D
for f in bench_1024x32_default.json bench_512x16_tsdf_config1.json bench_2048x64_config4.json bench_2ranks_one_device_gloo.json \
  bench_8ranks_one_device_gloo.json bench_auto_fallback_2x_one_device.json bench_inprocess_2x_one_device.json \
  bench_inprocess_8x_one_device.json extract_voxel_phases.txt first_view.txt issue_floor.txt kernel_stats_1024x32_cull0.csv \
  kernel_stats_1024x32_default.csv kernel_stats_1024x32_default_with_mc.csv kernel_stats_1024x32_tsdf.csv \
  kernel_stats_2048x64_config4.csv kernel_stats_512x16_tsdf_config1.csv kernel_trace_carve_fused_per_dispatch.txt \
  marching_cubes_brick_minima.txt mc_kernels_512_tsdf.txt mc_sweep_vs_bit_planes.txt mc_unprofiled.txt mc_wall.txt \
  one_view_final.txt one_view_pmc_final.txt row_kernel_final.txt per_view_launches.txt per_view_trace_default.txt \
  per_view_trace_tsdf.txt per_view_tsdf_pmc_final.txt phase_timing.json pmc_1024_marching_cubes.json pmc_1024x32_cull0.json \
  pmc_1024x32_default.json pmc_1024x32_tsdf.json pmc_2048x64_config4.json pmc_512x16_tsdf_config1.json slab_emulation.txt \
  status.txt streamed_emulation.txt write_ceiling.txt pytest_gpu_final.txt class_api_extractions.txt; do
  cp $S/$f $D/ 2>/dev/null || echo "missing $f"
done
cp $S/bench_2ranks_one_device_rccl.err $D/bench_2ranks_one_device_rccl.err.txt
cp $S/counters.json profiles/counters.json
grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $D/mc_wall.txt > /tmp/x.$$ && mv /tmp/x.$$ $D/mc_wall.txt
python3 - <<PY
import json, csv
d = json.loads(open("$D/bench_1024x32_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "| frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "traffic", r["traffic"])
h = r["headline_kernel"]
print("headline kernel", h["avg_launch_ms"], "traffic", h["traffic"], "pairs frac", h.get("frac_processed_pairs"), "api-equivalent", h["per_view_api_equivalent"]["frac"])
print("issue floor", r["issue_floor"].get("cull0"), r["issue_floor"].get("default_without_stores_only"), "| mc", r["mc"]["device_ms"], r["mc"]["wall_ms"])
print({k: (v["carve_ms_first_view"], v["carve_ms_per_view_after_first"]) for k, v in r["per_view_launches"].items()})
c0 = r["other_configs"]["configs[0]"]
print("bunny", c0["value"], "sequence", c0["sequence_wall_ms"], "xv", c0["extract_voxel_wall_ms"], "mc", c0["mc"]["wall_ms"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["extrapolated"])
print(set(v.get("build") for v in json.load(open("profiles/counters.json")).values() if isinstance(v, dict)))
for f in ("kernel_stats_1024x32_cull0.csv", "kernel_stats_1024x32_default.csv"):
    for row in csv.DictReader(open("$D/" + f)):
        if "carve_fused" in row["Name"]:
            print(f, row["Calls"], "avg %.3f min %.3f" % (float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6))
PY
tail -3 $D/kernel_trace_carve_fused_per_dispatch.txt
grep "rep 2" $D/one_view_final.txt
