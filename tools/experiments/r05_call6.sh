# round 5, GPU session 6: x16 bender -- the bender-alone test, and an A/B of blocks per wave / waves per CU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x16_bender" > gpurun_out/c6/pytest_bender.txt 2>&1; grep -E "bender .* x 64|passed|failed|Error" gpurun_out/c6/pytest_bender.txt | cut -c1-300
ab() { # lib-suffix bender extra-args tag
  NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so NRNERF_X16_BENDER=$2 timeout 300 python bench.py $3 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_ms_per_step']; print('$4 [$1 x16 bender=$2]', d['value'], d['ms_per_step'], 'bend_fine', k.get('bend_fine'), 'bend_coarse', k.get('bend_coarse'))" || echo "variant [$1 $4] FAILED"
}
{ ab "" 0 "" headline; for v in "" _b2o3 _b4 _b4w4 _b4w4o3; do ab "$v" 1 "" headline; done; for v in "" _b4 _b4w4o3; do ab "$v" 1 "--use-viewdirs --bend-depth 7" config4; done; } > gpurun_out/c6/ab_bender.txt 2>&1
grep "bender=" gpurun_out/c6/ab_bender.txt
for v in _b4 _b4w4o3; do NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "x16_bender" 2>&1 | tail -1; done
