set -x
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python bench.py > gpurun_out/r06_bench_default.log 2>&1
tail -1 gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_bf16.json
cut -c1-300 gpurun_out/r06_bench_bf16.json
bash tools/collect_train_profiles.sh r06 > gpurun_out/r06_train_profiles.log 2>&1
python tools/soak_determinism.py 12 > gpurun_out/r06_soak_determinism.txt 2>&1; tail -3 gpurun_out/r06_soak_determinism.txt
