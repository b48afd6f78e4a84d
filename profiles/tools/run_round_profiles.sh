#!/bin/bash
# Runs ON the GPU box (through gpurun): everything profiles/rNN/ is built from.
#   profiles/tools/run_round_profiles.sh <round dir name, e.g. r02>
# Leaves under gpurun_out/<round>/: the rocprofv3 kernel trace + PMC passes of the three carve workloads
# (default, view dropping off, TSDF) and of marching cubes, their summaries, the phase breakdown of the fused
# kernel, the 2-rank run of bench.py on one device, and the bench lines themselves.
set -u
RND=$1
R=$(pwd -P)
O=$R/gpurun_out/$RND
mkdir -p "$O"
export TMPDIR=/tmp
CTR=$O/counters.json
rm -f "$CTR" "$O/status.txt"
# carve workloads: trace + counters (no marching cubes inside these runs)
bash profiles/tools/profile_gpu.sh gpurun_out/$RND/default --no-mc
python profiles/tools/summarize_pmc.py "$O/default" "$O/pmc_1024x32_default.json" --key default_1024_32_b1_c1 --counters "$CTR"
python profiles/tools/per_dispatch.py "$O/default/trace_kernel_trace.csv" > "$O/kernel_trace_carve_fused_per_dispatch.txt"
grep -o '"avg_launch_ms": [0-9.]*' "$O/default/trace.log" | head -1 | sed 's/^/bench line of the same (traced) run: /' >> "$O/kernel_trace_carve_fused_per_dispatch.txt"
bash profiles/tools/profile_gpu.sh gpurun_out/$RND/cull0 --no-mc --cull 0
python profiles/tools/summarize_pmc.py "$O/cull0" "$O/pmc_1024x32_cull0.json" --key default_1024_32_b1_c0 --counters "$CTR"
bash profiles/tools/profile_gpu.sh gpurun_out/$RND/tsdf --no-mc --mode tsdf
python profiles/tools/summarize_pmc.py "$O/tsdf" "$O/pmc_1024x32_tsdf.json" --key tsdf_1024_32_b1_c1 --counters "$CTR"
# the other single-GPU configurations of BASELINE.json (bench.py's `configs` block reads these): trace + HBM bytes +
# VALU counters + clock of configs[1] (512^3 x 16 TSDF) and of the configs[4] shape (2048^3 x 64, four launches per step)
PASSES="trace fetch write sq1 grbm" bash profiles/tools/profile_gpu.sh gpurun_out/$RND/config1 --no-mc --config 1
python profiles/tools/summarize_pmc.py "$O/config1" "$O/pmc_512x16_tsdf_config1.json" --key tsdf_512_16_b1_c1 --counters "$CTR"
PASSES="trace fetch write sq1 grbm" STEPS_TRACE="--steps 3 --warmup 1" bash profiles/tools/profile_gpu.sh gpurun_out/$RND/config4 --no-mc --config 4
python profiles/tools/summarize_pmc.py "$O/config4" "$O/pmc_2048x64_config4.json" --key default_2048_64_b1_c1 --counters "$CTR"
# marching cubes: trace with the extraction inside, HBM bytes of its kernels
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$O/mc" -o trace --output-format csv -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants ) > "$O/mc_trace.log" 2>&1
( cd /tmp && rocprofv3 --pmc FETCH_SIZE -d "$O/mc" -o fetch --output-format csv -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-variants ) > "$O/mc_fetch.log" 2>&1
( cd /tmp && rocprofv3 --pmc WRITE_SIZE -d "$O/mc" -o write --output-format csv -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-variants ) > "$O/mc_write.log" 2>&1
python profiles/tools/summarize_pmc.py "$O/mc" "$O/pmc_1024_marching_cubes.json" --mc-key mc_1024 --counters "$CTR"
profiles/tools/ab_mc.sh prod > "$O/mc_unprofiled.txt" 2>&1
# the one-sweep cell search against the bit planes in memory ("mcsweep" 1 / 0), alternating in one process
profiles/tools/ab_mc_sweep.sh prod 2>&1 | grep mcsweep > "$O/mc_sweep_vs_bit_planes.txt"
# marching cubes with / without the brick minima, the per-view call pattern with / without the live-workgroup list
python profiles/tools/mc_skip.py 2>&1 | grep mcskip > "$O/marching_cubes_brick_minima.txt"
for m in default tsdf; do python profiles/tools/per_view.py 1024 $m 2>&1 | grep livelist; done > "$O/per_view_launches.txt"
# single-view launches: kernel trace (per kernel us per view) and the counters of the weighted-average launch
for m in default tsdf; do bash profiles/tools/per_view_trace.sh "gpurun_out/$RND/pvt_$m" 1024 $m 16 > "$O/per_view_trace_$m.txt" 2>&1; done
bash profiles/tools/pmc_per_view.sh "gpurun_out/$RND/pv_tsdf" 1024 tsdf > /dev/null 2>&1; cp "$O/pv_tsdf/summary.txt" "$O/per_view_tsdf_pmc_final.txt"
# issue floor of the fused kernel: the same instruction stream without tile loads (T), without stores (S), without both
if [ -f build/variants/floorTS/libvacancy_hip.so ]; then
  profiles/tools/ab_variants.sh "$O/floor" devprod floorT floorS floorTS devprod floorT floorS floorTS > "$O/issue_floor.txt" 2>&1
  python profiles/tools/summarize_floor.py "$O/issue_floor.txt" "$CTR"
fi
# N GPUs from one process (threads + vcy_halo_allgather): two "GPUs" on this one device
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --launch inprocess --slabs-per-gpu 2 --steps 5 --warmup 1 > "$O/bench_inprocess_2x_one_device.json" 2> "$O/bench_inprocess_2x_one_device.err"
echo "in-process 2 x one device rc=$?" >> "$O/status.txt"
# --launch auto (the default): torch.distributed.run first; RCCL refuses two ranks on one device, so the parent falls
# back to the in-process form and says so in config.launch_note
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --slabs-per-gpu 2 --steps 5 --warmup 1 --no-mc > "$O/bench_auto_fallback_2x_one_device.json" 2> "$O/bench_auto_fallback_2x_one_device.err"
echo "launch auto -> in-process fallback rc=$?" >> "$O/status.txt"
# phase breakdown of the fused kernel (development build with s_memtime marks)
if [ -f build/variants/phase/libvacancy_hip.so ]; then
  VCY_HIP_LIB=build/variants/phase/libvacancy_hip.so python profiles/tools/phase_timing.py > "$O/phase_timing.log" 2>&1
  cp gpurun_out/phase_timing.json "$O/phase_timing.json"
fi
# the multi-rank path of bench.py on this one device: 2 ranks, 2 slabs each, halo exchange over gloo (RCCL refuses
# two ranks on one device; --allow-gloo is the documented escape for exactly this check)
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --launch torchrun --slabs-per-gpu 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --allow-gloo > "$O/bench_2ranks_one_device_gloo.json" 2> "$O/bench_2ranks_one_device_gloo.err"
echo "2 ranks (gloo) rc=$?" >> "$O/status.txt"
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --launch torchrun --steps 3 --warmup 1 --no-cpu-baseline --no-variants > "$O/bench_2ranks_one_device_rccl.json" 2> "$O/bench_2ranks_one_device_rccl.err"
echo "2 ranks (rccl on one device, expected to be refused) rc=$?" >> "$O/status.txt"
# round 5: the 8-rank launches rehearsed on this one device (gloo for the exchange: RCCL refuses two ranks per device), the
# in-process form with eight "GPUs", the streamed emulation (sharded against replicated SDF producer), ExtractVoxel phases
VCY_BENCH_FORCE_DEVICE=0 VCY_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --launch torchrun --allow-gloo --verify-mesh --variants streamed --steps 10 > "$O/bench_8ranks_one_device_gloo.json" 2> "$O/bench_8ranks_one_device_gloo.err"; echo "8 ranks torchrun gloo rc=$?" >> "$O/status.txt"
VCY_BENCH_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --launch inprocess --verify-mesh --variants streamed --steps 10 > "$O/bench_inprocess_8x_one_device.json" 2> "$O/bench_inprocess_8x_one_device.err"; echo "8x inprocess rc=$?" >> "$O/status.txt"
timeout 1200 python profiles/tools/streamed_emulation.py > "$O/streamed_emulation.txt" 2>&1; echo "streamed emulation rc=$?" >> "$O/status.txt"
timeout 1200 python profiles/tools/slab_emulation.py > "$O/slab_emulation.txt" 2>&1; echo "slab emulation rc=$?" >> "$O/status.txt"
VCY_XV_TIMING=1 timeout 300 python profiles/tools/xv_timing.py > "$O/extract_voxel_phases.txt" 2>&1
# the same extractions through the C++ class API, Mesh included (a fresh Mesh per view, as examples.cc has it)
(timeout 300 vacancy_amd/host/host_selftest tests/golden/bunny xvtime 2.5 2>&1 | grep XVTIME) > "$O/class_api_extractions.txt"
for m in default tsdf; do timeout 300 python profiles/tools/first_view.py 1024 $m; done > "$O/first_view.txt" 2>&1
# round 6: single-view launches through the instance compiled for one view against the general one, and the few-view
# flavour (a wave walks the bricks of a row segment) against both; what a marching-cubes CALL costs (wall) per size
PARAM=oneview VALUES=0,1 timeout 300 python profiles/tools/row_kernel.py 1024 tsdf default > "$O/one_view_final.txt" 2>&1
PARAM=rowkernel VALUES=0,-1 timeout 300 python profiles/tools/row_kernel.py 1024 tsdf default > "$O/row_kernel_final.txt" 2>&1
for m in tsdf default; do PARAM=oneview bash profiles/tools/pmc_rows.sh "gpurun_out/$RND/pmc_one_view" $m 1; PARAM=oneview bash profiles/tools/pmc_rows.sh "gpurun_out/$RND/pmc_one_view" $m 0; done 2>&1 | grep knob > "$O/one_view_pmc_final.txt"
VCY_MC_TIMING_ONCE=1 timeout 300 python profiles/tools/mc_wall.py > "$O/mc_wall.txt" 2>&1
N=512 NV=16 MODE=tsdf bash profiles/tools/mc_kernel_times.sh prod > "$O/mc_kernels_512_tsdf.txt" 2>&1
timeout 300 python profiles/tools/write_ceiling.py > "$O/write_ceiling.txt" 2>&1
# the summaries profiles/rNN/ keeps of the traced runs
for w in default cull0 tsdf; do cp "$O/$w/trace_kernel_stats.csv" "$O/kernel_stats_1024x32_$w.csv"; done
cp "$O/config1/trace_kernel_stats.csv" "$O/kernel_stats_512x16_tsdf_config1.csv"; cp "$O/config4/trace_kernel_stats.csv" "$O/kernel_stats_2048x64_config4.csv"
cp "$O/mc/trace_kernel_stats.csv" "$O/kernel_stats_1024x32_default_with_mc.csv"
# the bench lines: with the counters of this session next to them
mkdir -p profiles && cp "$CTR" profiles/counters.json
python bench.py > "$O/bench_1024x32_default.json" 2> "$O/bench_default.err"; echo "bench default rc=$?" >> "$O/status.txt"
python bench.py --config 1 --no-variants > "$O/bench_512x16_tsdf_config1.json" 2> "$O/bench_config1.err"; echo "bench config1 rc=$?" >> "$O/status.txt"
python bench.py --config 4 --steps 5 --warmup 1 --variants streamed --no-cpu-baseline > "$O/bench_2048x64_config4.json" 2> "$O/bench_config4.err"; echo "bench config4 rc=$?" >> "$O/status.txt"
cat "$O/status.txt"
ls "$O"
