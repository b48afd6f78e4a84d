#!/bin/bash
# Runs ON the GPU box (round 5): new GPU tests, the 8-rank rehearsals on one device, the streamed emulation.
set -u
O=gpurun_out/r05b; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded_silhouette or make_sdf_batch or cpp_ or widen or counter_width or native_rccl" --durations=10 ) > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" > $O/status.txt
# eight processes (torch.distributed.run) on the one device, gloo for the exchange (RCCL refuses two ranks per device)
VCY_BENCH_FORCE_DEVICE=0 VCY_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --launch torchrun --allow-gloo --verify-mesh --variants streamed --steps 10 > $O/bench_8ranks_one_device_gloo.json 2> $O/bench_8ranks_one_device_gloo.err; echo "8 ranks torchrun gloo rc=$?" >> $O/status.txt
# eight host threads, eight contexts, native library all-gather (one communicator rank)
VCY_BENCH_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --launch inprocess --verify-mesh --variants streamed --steps 10 > $O/bench_inprocess_8x_one_device.json 2> $O/bench_inprocess_8x_one_device.err; echo "8x inprocess rc=$?" >> $O/status.txt
VCY_BENCH_FORCE_DEVICE=0 VCY_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 4 --launch torchrun --allow-gloo --verify-mesh --variants streamed --steps 10 > $O/bench_4ranks_one_device_gloo.json 2> $O/bench_4ranks_one_device_gloo.err; echo "4 ranks torchrun gloo rc=$?" >> $O/status.txt
timeout 1200 python profiles/tools/streamed_emulation.py > $O/streamed_emulation.txt 2>&1; echo "streamed emulation rc=$?" >> $O/status.txt
cat $O/status.txt; tail -5 $O/pytest_new.log; tail -c 1500 $O/bench_8ranks_one_device_gloo.json; tail -5 $O/bench_8ranks_one_device_gloo.err; cat $O/streamed_emulation.txt
