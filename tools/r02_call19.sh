#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/power_trace.py bf16 250 2>&1 | tail -4
timeout 300 python tools/power_trace.py f16 200 2>&1 | tail -3
bash tools/collect_profiles.sh r02 2>&1 | tail -12
