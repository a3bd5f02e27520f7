#!/bin/bash
# Run on the GPU box (through gpurun): one bench line + one `rocprofv3 --kernel-trace --stats` summary per NON-DEFAULT
# configuration of BASELINE.json, so that DESIGN.md's table of those configurations rests on committed artefacts:
#   config4   use_viewdirs=True + 7-layer ray-bending MLP (finite-difference directions), bf16, one 512x384 frame
#   config5   one 1080p frame (2 073 600 rays), f16 weights, 65 536-ray chunks (and the default 2^20-ray launches)
#   w128      --netwidth 128 --netwidth_fine 128 (compiled architecture 5), bf16
# Outputs: gpurun_out/<tag>_<config>_bench.json, gpurun_out/<tag>_<config>_kernel_stats.txt  (copy into profiles/).
TAG=${1:-r03}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-train-step --min-gpu-seconds 0"
run() {   # name, steps, warmup, bench arguments...
    local name=$1 steps=$2 warm=$3; shift 3
    python $R/bench.py --steps $steps --warmup $warm $COMMON "$@" 2> $R/gpurun_out/${TAG}_${name}_bench.err | grep '"metric"' > $R/gpurun_out/${TAG}_${name}_bench.json
    local out=/tmp/prof_${name}
    rm -rf $out
    timeout 300 rocprofv3 --kernel-trace --stats -d $out -o s -- python $R/bench.py --steps 4 --warmup 1 $COMMON --no-psnr "$@" > $out.log 2>&1
    local db=$(find $out -name "*.db" | head -1)
    (echo "# $name: python bench.py --steps 4 --warmup 1 $COMMON --no-psnr $*"; python $R/tools/rocprof_summary.py "$db") > $R/gpurun_out/${TAG}_${name}_kernel_stats.txt 2>&1
    rm -rf $out
}
run config4 10 3 --use-viewdirs --bend-depth 7
run config5 3 1 --rays 2073600 --precision f16 --chunk 65536 --max-rays-per-launch 65536 --psnr-rays 65536
run config5_default_launch 3 1 --rays 2073600 --precision f16 --chunk 65536 --psnr-rays 65536
run w128 10 3 --netwidth 128
run strong_shard_24576 20 5 --rays 24576
# round 4: shapes outside the compiled set (the run-time-parameterised kernel, csrc/nrnerf_generic.h), synthetic weights
run generic_w192 5 2 --netwidth 192
run generic_w512 3 1 --netwidth 512
cd $R
for f in gpurun_out/${TAG}_*_bench.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j["dtype"], f'{j["value"] / 1e6:.3f} M rays/s', j["ms_per_step"], "ms/step, fine kernel", j["roofline"]["achieved"], "TFLOP/s", j["roofline"]["frac"], j.get("psnr_vs_oracle_db", {}).get("rgb_map"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
