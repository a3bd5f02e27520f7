#!/bin/bash
# round 4, GPU call 41: "f16" mode on the 16x16x32 trunk-only kernel by default: every test that renders in f16 mode, smoke, and the
# f16 bench line against NRNERF_X16=0 on the same box
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_fitted_checkpoint.py -q -m gpu -k "f16 or chunk or fitted" 2>&1 | grep -v amdgpu.ids | tail -n 6 | tee gpurun_out/r04_f16_x16_tests.txt
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | grep "smoke\]" | tee gpurun_out/r04_smoke.txt
B="--precision f16 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --steps 10 --warmup 3"
run() { NRNERF_X16=$1 timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('f16 mode X16=$1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['kernels_ms_per_step'], d.get('psnr_vs_oracle_db', {}).get('rgb_map'))" || echo "X16=$1 FAILED"; }
{ run 0; run 1; run 0; run 1; } | tee gpurun_out/r04_x16_f16_ab.txt
