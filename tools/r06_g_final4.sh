set -x
python -m pytest tests -m gpu -q -s > gpurun_out/r06_gpu_suite_ref.txt 2>&1; tail -3 gpurun_out/r06_gpu_suite_ref.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; tail -4 gpurun_out/r06_smoke.txt
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python bench.py > gpurun_out/r06_bench_default.log 2>&1
tail -1 gpurun_out/r06_bench_default.log > gpurun_out/r06_bench_bf16.json
cut -c1-300 gpurun_out/r06_bench_bf16.json
bash tools/collect_train_profiles.sh r06 > gpurun_out/r06_train_profiles.log 2>&1
python tools/train_step_sequence.py 1024 bf16 > gpurun_out/r06_p1_sequence.txt 2>&1; grep "^#" gpurun_out/r06_p1_sequence.txt
