set -x
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python bench.py > gpurun_out/r06_bench_default.log 2>&1
tail -1 gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_bf16.json
cut -c1-300 gpurun_out/r06_bench_bf16.json
bash tools/collect_config_evidence.sh r06 > gpurun_out/r06_config_evidence.log 2>&1
tail -9 gpurun_out/r06_config_evidence.log
python bench.py --frames 8 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r06_frames8.json
bash tools/collect_train_profiles.sh r06 > gpurun_out/r06_train_profiles.log 2>&1
python tools/power_trace.py bf16 250 > gpurun_out/r06_power_trace.txt 2>&1
python tools/soak_determinism.py 12 > gpurun_out/r06_soak_determinism.txt 2>&1; tail -2 gpurun_out/r06_soak_determinism.txt
