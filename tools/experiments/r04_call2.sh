#!/bin/bash
# round 4, GPU call 2: fused compositing (bit-identity + A/B), > 256 samples, the generic-architecture kernel, pins
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused_into or large_sample or refused" > gpurun_out/r04_c2_fused.log 2>&1
echo "fused/large rc=$?"; tail -n 15 gpurun_out/r04_c2_fused.log
NRNERF_PIN_RECORD=gpurun_out/r04_pins2.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "generic or full_frame" > gpurun_out/r04_c2_generic.log 2>&1
echo "generic rc=$?"; grep -a "pinned fp32\|passed\|failed\|Error\|196 608\|generic" gpurun_out/r04_c2_generic.log | tail -n 40
for U in 0 1; do
  NRNERF_UNFUSED_COMPOSITE=$U python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 > gpurun_out/r04_c2_bench_unfused$U.json 2> gpurun_out/r04_c2_bench_unfused$U.err
  echo "bench unfused=$U rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c2_bench_unfused$U.json").read().strip().splitlines()[-1])
print("unfused=$U", d["value"], d["ms_per_step"], d.get("kernels_ms_per_step"), d["roofline"]["frac"], d.get("psnr_vs_oracle_db",{}).get("rgb_map"))
PY
done
