#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/soak_determinism.py 60 > gpurun_out/r02_soak.txt 2>&1; tail -12 gpurun_out/r02_soak.txt
