# one box: BASELINE's other configurations with fixed (NRNERF_FIXED_SHARES=1) and dynamic shares of the work, interleaved
C="--no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
run() { name=$1; shift; for f in 1 0 1 0; do echo -n "$name fixed_shares=$f: "; NRNERF_FIXED_SHARES=$f python bench.py $C "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"]/1e6,3), "M rays/s", d["ms_per_step"], "ms", r["kernels_ms_per_step"])'; done; }
run config4 --steps 10 --warmup 3 --use-viewdirs --bend-depth 7
run config5 --steps 3 --warmup 1 --rays 2073600 --precision f16 --chunk 65536 --max-rays-per-launch 65536 --psnr-rays 65536
run w128 --steps 10 --warmup 3 --netwidth 128
run strong_shard_24576 --steps 20 --warmup 5 --rays 24576
run f16_frame --steps 10 --warmup 3 --precision f16
