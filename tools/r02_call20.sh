#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/power_trace.py bf16 250 2>&1 | tail -4
timeout 300 python tools/power_trace.py f16 200 2>&1 | tail -4
timeout 300 python tools/power_trace.py f32 12 2>&1 | tail -4
