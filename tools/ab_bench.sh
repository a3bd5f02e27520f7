#!/bin/bash
# A/B the builds under nonrigid_nerf_amd/lib (see csrc/Makefile TUNE/SUFFIX) in one GPU session.
#   tools/ab_bench.sh "" _w4 _pf6        -> one line per variant: rays/s, ms/step, fine-net TFLOP/s, frac
cd "$(dirname "$0")/.."
for v in "$@"; do
  NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$v]', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['kernels_ms_per_step'])" || echo "variant [$v] FAILED"
done
