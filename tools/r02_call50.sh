#!/bin/bash
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_training.py -q -m gpu 2>&1 | tail -2
