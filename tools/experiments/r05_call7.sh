# round 5, GPU session 7: x16 bender with its inputs requested one iteration ahead and the frame's latent code read once
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "x16_bender or split_bender or fused_into" 2>&1 | tail -2
ab() { # lib-suffix bender extra-args tag
  NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so NRNERF_X16_BENDER=$2 timeout 300 python bench.py $3 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_ms_per_step']; print('$4 [$1 x16 bender=$2]', d['value'], d['ms_per_step'], 'bend_fine', k.get('bend_fine'), 'bend_coarse', k.get('bend_coarse'))" || echo "variant [$1 $4] FAILED"
}
{ ab "" 0 "" headline; ab "" 1 "" headline; ab _w4o4 1 "" headline; ab "" 1 "" headline; ab "" 1 "--use-viewdirs --bend-depth 7" config4; ab "" 1 "--netwidth 128" w128; } > gpurun_out/c7/ab_bender.txt 2>&1
grep "bender=" gpurun_out/c7/ab_bender.txt
