set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
ab() { # lib-suffix x16mode tag
  NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$1 x16=$2]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $2] FAILED"
}
{
ab "" 1; ab "" 2; ab _nb2 1; ab _nb2 2
ab "" 1; ab "" 2; ab _nb2 1; ab _nb2 2
} > gpurun_out/c1/ab.txt 2>&1
cat gpurun_out/c1/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "split_bender or x16 or fused_into" > gpurun_out/c1/pytest_default.txt 2>&1; tail -5 gpurun_out/c1/pytest_default.txt
timeout 600 python -m pytest tests/test_fitted_checkpoint.py -x -q -s > gpurun_out/c1/pytest_fitted.txt 2>&1; tail -5 gpurun_out/c1/pytest_fitted.txt
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_nb2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender_path_at_full" > gpurun_out/c1/pytest_nb2.txt 2>&1; tail -5 gpurun_out/c1/pytest_nb2.txt
