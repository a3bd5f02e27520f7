#!/bin/bash
timeout 600 python -m pytest tests/test_training.py -q -m gpu -k "point_the_same_way" 2>&1 | tail -3
