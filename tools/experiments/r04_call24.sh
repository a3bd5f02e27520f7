#!/bin/bash
# round 4, GPU call 24: the trunk-only fine pass on v_mfma 16x16x32 (nrnerf_net_x16.h), first light: accuracy on the fitted checkpoint
# and A/B of the default bench against the 32x32x16 kernel with fused compositing (NRNERF_X16=0)
set -u
mkdir -p gpurun_out
B="--no-cpu-baseline --no-train-step --min-gpu-seconds 0 --steps 10 --warmup 3"
for v in 0 1; do
  NRNERF_X16=$v timeout 300 python bench.py $B 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('NRNERF_X16=$v', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'], d['psnr_vs_oracle_db']['all_precisions'])" || echo "NRNERF_X16=$v FAILED"
done | tee gpurun_out/r04_x16_ab.txt
python -m pytest tests/test_fitted_checkpoint.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -n 5
