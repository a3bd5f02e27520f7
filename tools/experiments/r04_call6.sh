#!/bin/bash
# round 4, GPU call 6: products rounded behind an opaque asm: bit-identity of the fused compositing, split-vs-fused, surface; generic NSB A/B
set -u
mkdir -p gpurun_out
python tools/experiments/debug_fused_composite.py 2>&1 | grep "rgb_map\|acc_map" | head -n 6
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused or split_bender or surface or boundary_contract" > gpurun_out/r04_c6_bits.log 2>&1
echo "bits rc=$?"; grep -a "passed\|failed" gpurun_out/r04_c6_bits.log | tail -n 3; grep -a "^FAILED\|^ERROR" gpurun_out/r04_c6_bits.log | head -n 20
for NSB in 2 4; do
NRNERF_GENERIC_NSB=$NSB NRNERF_FORCE_GENERIC=1 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --no-psnr | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generic bf16 NSB=$NSB', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], d['roofline']['frac'])"
done
