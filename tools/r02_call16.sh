#!/bin/bash
mkdir -p gpurun_out
bash tools/ab_bench.sh "" _pf "" _pf
NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_pf.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "split_bender_path or full_size" 2>&1 | tail -3
