#!/bin/bash
# round 4, GPU call 12: generic kernel, weight-fragment prefetch depth (NRN_GEN_PF) and samples per workgroup (NRN_GEN_NSB), A/B in one session
# baseline (one slab ahead): bf16 84.0 ms, f32 487.9 ms per 512x384 frame on the default shape (NRNERF_FORCE_GENERIC=1)
set -u
mkdir -p gpurun_out
B="--no-cpu-baseline --no-train-step --no-psnr --min-gpu-seconds 0"
for v in "" _pf1 _pf2 _pf6 _pf4n4; do
  for prec in bf16 f32; do
    if [ "$prec" = f32 ] && [ "$v" != "" ] && [ "$v" != "_pf2" ]; then continue; fi
    NRNERF_FORCE_GENERIC=1 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 300 python bench.py --steps 4 --warmup 2 --precision $prec $B 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$v] $prec', d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "variant [$v] $prec FAILED"
  done
done | tee gpurun_out/r04_generic_pf_ab.txt
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip.so python bench.py --steps 4 --warmup 2 --netwidth 512 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w512 pf4', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_pf1.so python bench.py --steps 4 --warmup 2 --netwidth 512 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w512 pf1', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
