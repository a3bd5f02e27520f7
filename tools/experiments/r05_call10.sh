# round 5, GPU session 10: width 128 on the x16 kernel -- fragments requested ahead: 8 (39 registers spilled), 4 (9 spilled, default now), 2 (none)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10; export TMPDIR=/tmp
ab() { NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --netwidth 128 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('w128 [$1]', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "variant [$1] FAILED"; }
{ for rep in 1 2; do ab _pf8; ab ""; ab _pf2; done; } > gpurun_out/c10/ab_w128_pf.txt 2>&1
grep "^w128" gpurun_out/c10/ab_w128_pf.txt
