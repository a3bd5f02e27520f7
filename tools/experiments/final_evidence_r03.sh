R=$PWD
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1
python bench.py > gpurun_out/r03_bench_bf16_final.json 2> gpurun_out/r03_bench_final.err
# config 4 only (the one the fold changes)
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-train-step --min-gpu-seconds 0"
python $R/bench.py --steps 10 --warmup 3 $COMMON --use-viewdirs --bend-depth 7 2> $R/gpurun_out/r03_config4_bench.err | grep '"metric"' > $R/gpurun_out/r03_config4_bench.json
rm -rf /tmp/prof_c4; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o s -- python $R/bench.py --steps 4 --warmup 1 $COMMON --no-psnr --use-viewdirs --bend-depth 7 > /tmp/prof_c4.log 2>&1
db=$(find /tmp/prof_c4 -name "*.db" | head -1)
(echo "# config4: python bench.py --steps 4 --warmup 1 $COMMON --no-psnr --use-viewdirs --bend-depth 7"; python $R/tools/rocprof_summary.py "$db") > $R/gpurun_out/r03_config4_kernel_stats.txt 2>&1
cd $R
bash tools/collect_train_profiles.sh r03 > gpurun_out/r03_collect_train.log 2>&1
python tools/train_step_scaling.py > gpurun_out/r03_train_step_modes.txt 2>&1
python tools/train_step_configs.py > gpurun_out/r03_train_step_configs.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03_gpu_suite.txt
tail -2 gpurun_out/r03_gpu_suite.txt
