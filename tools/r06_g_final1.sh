set -x
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python bench.py > gpurun_out/r06_bench_default.log 2>&1
tail -1 gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_bf16.json
cut -c1-400 gpurun_out/r06_bench_bf16.json
