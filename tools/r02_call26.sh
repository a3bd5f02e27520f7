#!/bin/bash
timeout 300 python tools/debug_split_views.py 2>&1 | tail -12
