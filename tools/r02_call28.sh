#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "boundary_contract or split_bender_path_equals" 2>&1 | tail -4
