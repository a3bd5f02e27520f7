#!/bin/bash
# The GPU sessions of round 5 as they were run (one function each; `gpurun -- 'bash tools/experiments/r05_sessions.sh <n>'`).  Lab notes:
# the A/B numbers they produced are under profiles/r05_*_ab.txt, the library suffixes (_nb2, _n44, _b4, _pf2, ...) are builds of
# csrc/Makefile with TUNE=... SUFFIX=... that are not kept in the tree.
session_1() {
    set -x
    cd $GRAFT_REPO_ROOT
    mkdir -p gpurun_out/c1
    export TMPDIR=/tmp
    ab() { # lib-suffix x16mode tag
      NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$1 x16=$2]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $2] FAILED"
    }
    {
    ab "" 1; ab "" 2; ab _nb2 1; ab _nb2 2
    ab "" 1; ab "" 2; ab _nb2 1; ab _nb2 2
    } > gpurun_out/c1/ab.txt 2>&1
    cat gpurun_out/c1/ab.txt
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "split_bender or x16 or fused_into" > gpurun_out/c1/pytest_default.txt 2>&1; tail -5 gpurun_out/c1/pytest_default.txt
    timeout 600 python -m pytest tests/test_fitted_checkpoint.py -x -q -s > gpurun_out/c1/pytest_fitted.txt 2>&1; tail -5 gpurun_out/c1/pytest_fitted.txt
    NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip_nb2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender_path_at_full" > gpurun_out/c1/pytest_nb2.txt 2>&1; tail -5 gpurun_out/c1/pytest_nb2.txt
}

session_2() {
    # round 5, GPU session 2: fit the non-compiled acceptance checkpoint (coarse 192 / fine 320 wide); the new >256-sample gradient and
    # 16-bit render tests; the fitted-checkpoint acceptance tests incl. the generic-kernel routes; a bench line with the new fields
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c2; export TMPDIR=/tmp
    python oracle/fit_checkpoint.py --arch w192_320 --minutes 4.0 --out gpurun_out/fitted_w192_320.tar > gpurun_out/c2/fit_w192_320.log 2>&1 &
    FIT=$!
    timeout 1200 python -m pytest tests/test_training.py -x -q -k "fp32_gradients or bf16_gradients or native_bender_forward" > gpurun_out/c2/pytest_training.txt 2>&1; tail -4 gpurun_out/c2/pytest_training.txt
    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "large_sample_counts" > gpurun_out/c2/pytest_large.txt 2>&1; tail -4 gpurun_out/c2/pytest_large.txt
    wait $FIT; tail -3 gpurun_out/c2/fit_w192_320.log
    cp gpurun_out/fitted_w192_320.tar tests/golden/
    timeout 1200 python -m pytest tests/test_fitted_checkpoint.py -x -q -s > gpurun_out/c2/pytest_fitted.txt 2>&1; tail -4 gpurun_out/c2/pytest_fitted.txt
    python bench.py --no-train-step > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err; tail -c 600 gpurun_out/c2/bench.json
}

session_3() {
    # round 5, GPU session 3: the 128-wide trunk on the 16x16x32 kernel -- parity on its fitted checkpoint, and an A/B of blocks per wave x
    # waves per workgroup (default build: 4 x 8 (spills 39 registers); _n44: 4 x 4; _n28: 2 x 8) against the 32x32x16 kernels (NRNERF_X16=0)
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c3; export TMPDIR=/tmp
    ab() { # lib-suffix x16mode
      NRNERF_X16=$2 NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --netwidth 128 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('w128 [$1 x16=$2]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $2] FAILED"
    }
    { for rep in 1 2; do ab "" 0; ab "" 1; ab "" 2; ab _n44 2; ab _n28 2; done; } > gpurun_out/c3/ab_w128.txt 2>&1
    grep "^w128" gpurun_out/c3/ab_w128.txt
    timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "w128" > gpurun_out/c3/pytest_fitted_w128.txt 2>&1; tail -3 gpurun_out/c3/pytest_fitted_w128.txt
    for v in _n44 _n28; do NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 600 python -m pytest tests/test_fitted_checkpoint.py -x -q -k "w128 and full_frame" 2>&1 | tail -1; done
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant" 2>&1 | tail -2
}

session_4() {
    # round 5, GPU session 4: the view-dependent head on the 16x16x32 kernel (BASELINE config 4) -- parity, then A/B against the 32x32x16 kernels
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c4; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "config4" > gpurun_out/c4/pytest_fitted_config4.txt 2>&1; tail -3 gpurun_out/c4/pytest_fitted_config4.txt
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant or viewdirs or config4" > gpurun_out/c4/pytest_parity.txt 2>&1; tail -3 gpurun_out/c4/pytest_parity.txt
    ab() { # x16mode extra-args tag
      NRNERF_X16=$1 timeout 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['frac_issued_mfma'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $3] FAILED"
    }
    { for rep in 1 2; do for m in 0 1 2; do ab $m "--use-viewdirs --bend-depth 7" config4; done; done; for m in 0 2; do ab $m "--use-viewdirs" viewdirs_bend5; done; ab 2 "" headline; } > gpurun_out/c4/ab_config4.txt 2>&1
    grep "x16=" gpurun_out/c4/ab_config4.txt
}

session_5() {
    # round 5, GPU session 5: the stand-alone bender on 16x16x32 MFMAs (nrnerf_bend_x16.h) -- parity, then A/B against the 32x32x16 bender
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c5; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x16_bender" > gpurun_out/c5/pytest_bender.txt 2>&1; tail -6 gpurun_out/c5/pytest_bender.txt
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or split_bender or every_compiled_variant or x16" > gpurun_out/c5/pytest_parity.txt 2>&1; tail -3 gpurun_out/c5/pytest_parity.txt
    timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -k "default-default or config4-default or w128-default" > gpurun_out/c5/pytest_fitted.txt 2>&1; tail -3 gpurun_out/c5/pytest_fitted.txt
    ab() { # bender extra-args tag
      NRNERF_X16_BENDER=$1 timeout 300 python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16 bender=$1]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'])" || echo "variant [$1 $3] FAILED"
    }
    { for rep in 1 2; do ab 0 "" headline; ab 1 "" headline; done; ab 0 "--use-viewdirs --bend-depth 7" config4; ab 1 "--use-viewdirs --bend-depth 7" config4; ab 0 "--netwidth 128" w128; ab 1 "--netwidth 128" w128; } > gpurun_out/c5/ab_bender.txt 2>&1
    grep "bender=" gpurun_out/c5/ab_bender.txt
}

session_6() {
    # round 5, GPU session 6: x16 bender -- the bender-alone test, and an A/B of blocks per wave / waves per CU
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c6; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "x16_bender" > gpurun_out/c6/pytest_bender.txt 2>&1; grep -E "bender .* x 64|passed|failed|Error" gpurun_out/c6/pytest_bender.txt | cut -c1-300
    ab() { # lib-suffix bender extra-args tag
      NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so NRNERF_X16_BENDER=$2 timeout 300 python bench.py $3 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_ms_per_step']; print('$4 [$1 x16 bender=$2]', d['value'], d['ms_per_step'], 'bend_fine', k.get('bend_fine'), 'bend_coarse', k.get('bend_coarse'))" || echo "variant [$1 $4] FAILED"
    }
    { ab "" 0 "" headline; for v in "" _b2o3 _b4 _b4w4 _b4w4o3; do ab "$v" 1 "" headline; done; for v in "" _b4 _b4w4o3; do ab "$v" 1 "--use-viewdirs --bend-depth 7" config4; done; } > gpurun_out/c6/ab_bender.txt 2>&1
    grep "bender=" gpurun_out/c6/ab_bender.txt
    for v in _b4 _b4w4o3; do NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "x16_bender" 2>&1 | tail -1; done
}

session_7() {
    # round 5, GPU session 7: x16 bender with its inputs requested one iteration ahead and the frame's latent code read once
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c7; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "x16_bender or split_bender or fused_into" 2>&1 | tail -2
    ab() { # lib-suffix bender extra-args tag
      NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so NRNERF_X16_BENDER=$2 timeout 300 python bench.py $3 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_ms_per_step']; print('$4 [$1 x16 bender=$2]', d['value'], d['ms_per_step'], 'bend_fine', k.get('bend_fine'), 'bend_coarse', k.get('bend_coarse'))" || echo "variant [$1 $4] FAILED"
    }
    { ab "" 0 "" headline; ab "" 1 "" headline; ab _w4o4 1 "" headline; ab "" 1 "" headline; ab "" 1 "--use-viewdirs --bend-depth 7" config4; ab "" 1 "--netwidth 128" w128; } > gpurun_out/c7/ab_bender.txt 2>&1
    grep "bender=" gpurun_out/c7/ab_bender.txt
}

session_9() {
    # round 5, GPU session 9: the fused training loss (nrnerf_loss_*) -- parity, the step's launch count and time; then the whole GPU tier
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c9; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_training.py -x -q -k "fused_loss or with_the_fused_loss" > gpurun_out/c9/pytest_loss.txt 2>&1; tail -5 gpurun_out/c9/pytest_loss.txt
    timeout 600 python tools/train_step_profile.py 1024 bf16 2>&1 | tail -3
    timeout 600 python tools/experiments/step_kernel_sequence.py > gpurun_out/c9/train_step_kernel_sequence_1024.txt 2>&1; head -3 gpurun_out/c9/train_step_kernel_sequence_1024.txt; tail -2 gpurun_out/c9/train_step_kernel_sequence_1024.txt
    timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/c9/pytest_gpu_full.txt 2>&1; tail -8 gpurun_out/c9/pytest_gpu_full.txt
}

session_10() {
    # round 5, GPU session 10: width 128 on the x16 kernel -- fragments requested ahead: 8 (39 registers spilled), 4 (9 spilled, default now), 2 (none)
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10; export TMPDIR=/tmp
    ab() { NRNERF_LIB=$PWD/nonrigid_nerf_amd/lib/libnrnerf_hip$1.so timeout 300 python bench.py --netwidth 128 --steps 10 --warmup 3 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('w128 [$1]', d['value'], d['ms_per_step'], r['frac'], r['kernels_ms_per_step'])" || echo "variant [$1] FAILED"; }
    { for rep in 1 2; do ab _pf8; ab ""; ab _pf2; done; } > gpurun_out/c10/ab_w128_pf.txt 2>&1
    grep "^w128" gpurun_out/c10/ab_w128_pf.txt
}

session_11() {
    # round 5, GPU session 11: the width-class trunk kernel for non-compiled architectures (nrnerf_gx16.h) -- parity, then speed
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c11; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "w192_320 or generic" > gpurun_out/c11/pytest_fitted.txt 2>&1; grep -E "fitted checkpoint.*512x384|passed|failed|Error" gpurun_out/c11/pytest_fitted.txt | cut -c1-330
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generic" > gpurun_out/c11/pytest_generic.txt 2>&1; tail -3 gpurun_out/c11/pytest_generic.txt
    ab() { # x16 flag, extra args, tag
      NRNERF_X16=$1 NRNERF_FORCE_GENERIC=$4 timeout 300 python bench.py $2 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['coarse_pass']['frac'], r['kernels_ms_per_step'], d.get('psnr_vs_oracle_db',{}).get('rgb_map'))" || echo "variant [$1 $3] FAILED"
    }
    { ab 0 "--netwidth 192" generic_w192 0; ab 2 "--netwidth 192" generic_w192 0; ab 0 "--netwidth 512" generic_w512 0; ab 2 "--netwidth 512" generic_w512 0; ab 0 "" default_forced_generic 1; ab 2 "" default_forced_generic 1; ab 2 "--netwidth 320" generic_w320 0; ab 2 "--netwidth 64" generic_w64 0; } > gpurun_out/c11/ab_gx16.txt 2>&1
    grep "x16=" gpurun_out/c11/ab_gx16.txt
}

session_12() {
    # round 5, GPU session 12: the width-class kernel with the view-dependent head; generic parity; config-4-like shapes; then the whole tier
    cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12; export TMPDIR=/tmp
    timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "generic or w192" > gpurun_out/c12/pytest_fitted.txt 2>&1; grep -E "fitted checkpoint.*512x384|passed|failed|Error" gpurun_out/c12/pytest_fitted.txt | cut -c1-300
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generic" 2>&1 | tail -2
    ab() { NRNERF_X16=$1 NRNERF_FORCE_GENERIC=$4 timeout 300 python bench.py $2 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['frac_issued_mfma'], r['coarse_pass']['frac'], r['kernels_ms_per_step'], d.get('psnr_vs_oracle_db',{}).get('rgb_map'))" || echo "variant [$1 $3] FAILED"; }
    { ab 0 "--netwidth 192 --use-viewdirs" generic_w192_views 0; ab 2 "--netwidth 192 --use-viewdirs" generic_w192_views 0; ab 0 "--use-viewdirs --bend-depth 7" config4_forced_generic 1; ab 2 "--use-viewdirs --bend-depth 7" config4_forced_generic 1; ab 2 "--use-viewdirs --bend-depth 7" config4_compiled 0; ab 2 "--netwidth 192" generic_w192 0; } > gpurun_out/c12/ab_gx16_views.txt 2>&1
    grep "x16=" gpurun_out/c12/ab_gx16_views.txt
    timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/c12/pytest_gpu_full.txt 2>&1; tail -4 gpurun_out/c12/pytest_gpu_full.txt
}

session_$1
