#!/bin/bash
# round 4, GPU call 7: the committed evidence -- kernel trace + PMC passes of the bench command (tools/collect_profiles.sh r04), then the
# bench line itself (with cpu_baseline and train_step), smoke
set -u
mkdir -p gpurun_out
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
echo "collect rc=$?"; tail -n 5 gpurun_out/r04_collect.log
python bench.py > gpurun_out/r04_bench_bf16.json 2> gpurun_out/r04_bench_bf16.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r04_bench_bf16.json
python __graft_entry__.py smoke > gpurun_out/r04_smoke.txt 2>&1
echo "smoke rc=$?"; grep -a "smoke" gpurun_out/r04_smoke.txt
