# round 5, GPU session 4: the view-dependent head on the 16x16x32 kernel (BASELINE config 4) -- parity, then A/B against the 32x32x16 kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "config4" > gpurun_out/c4/pytest_fitted_config4.txt 2>&1; tail -3 gpurun_out/c4/pytest_fitted_config4.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant or viewdirs or config4" > gpurun_out/c4/pytest_parity.txt 2>&1; tail -3 gpurun_out/c4/pytest_parity.txt
ab() { # x16mode extra-args tag
  NRNERF_X16=$1 timeout 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['frac_issued_mfma'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $3] FAILED"
}
{ for rep in 1 2; do for m in 0 1 2; do ab $m "--use-viewdirs --bend-depth 7" config4; done; done; for m in 0 2; do ab $m "--use-viewdirs" viewdirs_bend5; done; ab 2 "" headline; } > gpurun_out/c4/ab_config4.txt 2>&1
grep "x16=" gpurun_out/c4/ab_config4.txt
