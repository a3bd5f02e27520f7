#!/bin/bash
# round 4, GPU call 4: the whole GPU tier on the build with explicit-rounding compositing + prefetch, then the generic kernel's speed
set -u
mkdir -p gpurun_out
rm -f gpurun_out/r04_pins4.jsonl
NRNERF_PIN_RECORD=gpurun_out/r04_pins4.jsonl timeout 1500 python -m pytest tests/ -q -m gpu -s > gpurun_out/r04_c4_suite.log 2>&1
echo "suite rc=$?"; grep -a "passed\|failed" gpurun_out/r04_c4_suite.log | tail -n 3; grep -a "^FAILED\|^ERROR" gpurun_out/r04_c4_suite.log | head -n 20
for P in bf16 f32; do
NRNERF_FORCE_GENERIC=1 python bench.py --steps 5 --warmup 2 --precision $P --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --psnr-rays 8192 > gpurun_out/r04_c4_bench_generic_$P.json 2> gpurun_out/r04_c4_bench_generic_$P.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c4_bench_generic_$P.json").read().strip().splitlines()[-1])
print("generic $P", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"], d["roofline"]["frac"], d.get("psnr_vs_oracle_db",{}).get("all_precisions"))
PY
done
