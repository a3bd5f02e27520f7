#!/bin/bash
# round 4, GPU call 42 (tools/with_reference.sh): the rest of the GPU tier on the final build (call 41 ran every test that renders in
# f16 mode: 88), and the f16 1080p configuration (BASELINE config 5) + one strong-scaling shard on the final build
set -u
mkdir -p gpurun_out
{ python -m pytest tests/ -q -m gpu --ignore tests/test_gpu_parity.py --ignore tests/test_fitted_checkpoint.py 2>&1 | grep -v amdgpu.ids | tail -n 3
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not (f16 or chunk or fitted)" 2>&1 | grep -v amdgpu.ids | tail -n 3; } | tee gpurun_out/r04_gpu_suite_rest.txt
COMMON="--no-cpu-baseline --no-train-step --min-gpu-seconds 0"
run() { local name=$1 steps=$2 warm=$3; shift 3
    python bench.py --steps $steps --warmup $warm $COMMON "$@" 2> gpurun_out/r04_${name}_bench.err | grep '"metric"' > gpurun_out/r04_${name}_bench.json
    python -c "
import json,sys; j=json.loads(open('gpurun_out/r04_${name}_bench.json').read().strip().splitlines()[-1])
print('${name}', j['dtype'], round(j['value']/1e6,3), 'M rays/s', j['ms_per_step'], 'ms/step', j['roofline']['kernel'], j['roofline']['frac'], j.get('psnr_vs_oracle_db',{}).get('rgb_map'))"; }
run config5 3 1 --rays 2073600 --precision f16 --chunk 65536 --max-rays-per-launch 65536 --psnr-rays 65536
run config5_default_launch 3 1 --rays 2073600 --precision f16 --chunk 65536 --psnr-rays 65536
run strong_shard_24576 20 5 --rays 24576
