# round 5, GPU session 11: the width-class trunk kernel for non-compiled architectures (nrnerf_gx16.h) -- parity, then speed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c11; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "w192_320 or generic" > gpurun_out/c11/pytest_fitted.txt 2>&1; grep -E "fitted checkpoint.*512x384|passed|failed|Error" gpurun_out/c11/pytest_fitted.txt | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generic" > gpurun_out/c11/pytest_generic.txt 2>&1; tail -3 gpurun_out/c11/pytest_generic.txt
ab() { # x16 flag, extra args, tag
  NRNERF_X16=$1 NRNERF_FORCE_GENERIC=$4 timeout 300 python bench.py $2 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'], d.get('psnr_vs_oracle_db',{}).get('rgb_map'))" || echo "variant [$1 $3] FAILED"
}
{ ab 0 "--netwidth 192" generic_w192 0; ab 2 "--netwidth 192" generic_w192 0; ab 0 "--netwidth 512" generic_w512 0; ab 2 "--netwidth 512" generic_w512 0; ab 0 "" default_forced_generic 1; ab 2 "" default_forced_generic 1; ab 2 "--netwidth 320" generic_w320 0; ab 2 "--netwidth 64" generic_w64 0; } > gpurun_out/c11/ab_gx16.txt 2>&1
grep "x16=" gpurun_out/c11/ab_gx16.txt
