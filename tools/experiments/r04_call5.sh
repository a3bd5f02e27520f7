#!/bin/bash
# round 4, GPU call 5: contraction off in composite_ray (+ prefetch): bit-identity, the whole GPU tier, generic kernel speed after tuning
set -u
mkdir -p gpurun_out
rm -f gpurun_out/r04_pins5.jsonl
python tools/experiments/debug_fused_composite.py 2>&1 | grep "rgb_map\|acc_map" | head -n 6
NRNERF_PIN_RECORD=gpurun_out/r04_pins5.jsonl timeout 1500 python -m pytest tests/ -q -m gpu -s > gpurun_out/r04_c5_suite.log 2>&1
echo "suite rc=$?"; grep -a "passed\|failed" gpurun_out/r04_c5_suite.log | tail -n 3; grep -a "^FAILED\|^ERROR" gpurun_out/r04_c5_suite.log | head -n 20
for P in bf16 f32; do
NRNERF_FORCE_GENERIC=1 python bench.py --steps 5 --warmup 2 --precision $P --no-cpu-baseline --no-train-step --min-gpu-seconds 0 --no-psnr > gpurun_out/r04_c5_bench_generic_$P.json 2> gpurun_out/r04_c5_bench_generic_$P.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_c5_bench_generic_$P.json").read().strip().splitlines()[-1])
print("generic $P", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"], d["roofline"]["frac"])
PY
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'], d['roofline']['frac'])"
