#!/bin/bash
timeout 120 ./nonrigid_nerf_amd/lib/hbm_probe 2>&1 | tail -8
