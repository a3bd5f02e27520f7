#!/bin/bash
# round 4, GPU call 46 (the round's last GPU seconds): the x16 kernel requesting the next iteration's points behind its last LDS-DMA
# requests: parity tests that run it in all its cases, the quick bench line, the PMC passes for the traffic hash
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "(fused_into_the_network and not f32) or (split_bender and (bf16 or f16))" 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/r04_points_ahead_tests.txt
python bench.py --no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['traffic'], r['kernels_ms_per_step'])" | tee gpurun_out/r04_points_ahead_bench.txt
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1; echo "collect rc=$?"
