#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "narrow" > gpurun_out/c15_tests.log 2>&1; tail -5 gpurun_out/c15_tests.log; grep -n "^\[narrow" gpurun_out/c15_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])"; }
timeout 200 $B --netwidth 128 2>&1 | tail -1 | show "netwidth 128 bf16"
timeout 200 $B --netwidth 128 --precision f16 2>&1 | tail -1 | show "netwidth 128 f16"
timeout 200 $B --netwidth 128 --precision f32 --steps 3 --warmup 1 2>&1 | tail -1 | show "netwidth 128 f32"
timeout 200 $B 2>&1 | tail -1 | show "headline bf16"
