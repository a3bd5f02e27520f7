# round 5, GPU session 5: the stand-alone bender on 16x16x32 MFMAs (nrnerf_bend_x16.h) -- parity, then A/B against the 32x32x16 bender
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x16_bender" > gpurun_out/c5/pytest_bender.txt 2>&1; tail -6 gpurun_out/c5/pytest_bender.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant or x16" > gpurun_out/c5/pytest_parity.txt 2>&1; tail -3 gpurun_out/c5/pytest_parity.txt
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -k "default-default or config4-default or w128-default" > gpurun_out/c5/pytest_fitted.txt 2>&1; tail -3 gpurun_out/c5/pytest_fitted.txt
ab() { # bender extra-args tag
  NRNERF_X16_BENDER=$1 timeout 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16 bender=$1]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $3] FAILED"
}
{ for rep in 1 2; do ab 0 "" headline; ab 1 "" headline; done; ab 0 "--use-viewdirs --bend-depth 7" config4; ab 1 "--use-viewdirs --bend-depth 7" config4; ab 0 "--netwidth 128" w128; ab 1 "--netwidth 128" w128; } > gpurun_out/c5/ab_bender.txt 2>&1
grep "bender=" gpurun_out/c5/ab_bender.txt
