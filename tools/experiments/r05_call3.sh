# round 5, GPU session 3: the 128-wide trunk on the 16x16x32 kernel -- parity on its fitted checkpoint, and an A/B of blocks per wave x
# waves per workgroup (default build: 4 x 8 (spills 39 registers); _n44: 4 x 4; _n28: 2 x 8) against the 32x32x16 kernels (NRNERF_X16=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3; export TMPDIR=/tmp
ab() { # lib-suffix x16mode
  NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --netwidth 128 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('w128 [$1 x16=$2]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $2] FAILED"
}
{ for rep in 1 2; do ab "" 0; ab "" 1; ab "" 2; ab _n44 2; ab _n28 2; done; } > gpurun_out/c3/ab_w128.txt 2>&1
grep "^w128" gpurun_out/c3/ab_w128.txt
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "w128" > gpurun_out/c3/pytest_fitted_w128.txt 2>&1; tail -3 gpurun_out/c3/pytest_fitted_w128.txt
for v in _n44 _n28; do NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 600 python -m pytest tests/test_fitted_checkpoint.py -x -q -k "w128 and full_frame" 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant" 2>&1 | tail -2
