O=gpurun_out/r03
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --launch inprocess --slabs-per-gpu 2 --steps 5 --warmup 1 > $O/bench_inprocess_2x_one_device.json 2> $O/bench_inprocess_2x_one_device.err; echo rc=$?
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --slabs-per-gpu 2 --steps 5 --warmup 1 --no-mc > $O/bench_auto_fallback_2x_one_device.json 2> $O/bench_auto_fallback_2x_one_device.err; echo rc=$?
VCY_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --launch torchrun --slabs-per-gpu 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --allow-gloo > $O/bench_2ranks_one_device_gloo.json 2> $O/bench_2ranks_one_device_gloo.err; echo rc=$?
python bench.py > $O/bench_1024x32_default.json 2> $O/bench_default.err; echo rc=$?
