#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py tests/test_gpu_parity.py -q -m gpu -k "native_bender or oracle_autograd or reference_golden or split_bender_path_equals or fits" 2>&1 | tail -3
timeout 300 python tools/train_step_scaling.py 2>&1 | grep "bf16\|f32"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
timeout 300 $B --precision f32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 inference', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
