#!/bin/bash
timeout 600 python tools/render_path_bench.py 300 384 512 config4 2>&1 | tail -2
