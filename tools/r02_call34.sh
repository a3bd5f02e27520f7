#!/bin/bash
mkdir -p gpurun_out
timeout 400 python tools/train_step_breakdown.py bf16 1024 > gpurun_out/c34_breakdown_1024.log 2>&1; grep -n "per step\|free-running" gpurun_out/c34_breakdown_1024.log; sed -n '/Name/,$p' gpurun_out/c34_breakdown_1024.log | cut -c1-70,150-240 | head -30
timeout 400 python tools/train_step_breakdown.py bf16 16384 > gpurun_out/c34_breakdown_16384.log 2>&1; grep -n "per step\|free-running" gpurun_out/c34_breakdown_16384.log; sed -n '/Name/,$p' gpurun_out/c34_breakdown_16384.log | cut -c1-70,150-240 | head -24
