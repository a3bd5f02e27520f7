#!/bin/bash
# round 4, GPU call 8: re-collect the committed evidence after the last change to the device sources (fp32 generic fragments)
set -u
mkdir -p gpurun_out
bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
echo "collect rc=$?"
python bench.py > gpurun_out/r04_bench_bf16.json 2> gpurun_out/r04_bench_bf16.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r04_bench_bf16.json
NRNERF_FORCE_GENERIC=1 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --psnr-rays 8192 > gpurun_out/r04_generic_default_shape_bf16_bench.json 2>/dev/null
NRNERF_FORCE_GENERIC=1 python bench.py --steps 3 --warmup 1 --precision f32 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --psnr-rays 8192 > gpurun_out/r04_generic_default_shape_f32_bench.json 2>/dev/null
python tools/experiments/generic_vs_eager.py 2>&1 | grep -v amdgpu.ids | tail -n 1 > gpurun_out/r04_generic_vs_eager.txt; cat gpurun_out/r04_generic_vs_eager.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "generic or golden" 2>&1 | grep -v amdgpu.ids | tail -n 2
