# round 5, GPU session 12: the width-class kernel with the view-dependent head; generic parity; config-4-like shapes; then the whole tier
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c12; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fitted_checkpoint.py -x -q -s -k "generic or w192" > gpurun_out/c12/pytest_fitted.txt 2>&1; grep -E "fitted checkpoint.*512x384|passed|failed|Error" gpurun_out/c12/pytest_fitted.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "generic" 2>&1 | tail -2
ab() { NRNERF_X16=$1 NRNERF_FORCE_GENERIC=$4 timeout 300 python bench.py $2 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --min-gpu-seconds 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 [x16=$1]', d['value'], d['ms_per_step'], r['frac'], r['frac_issued_mfma'], r['coarse_pass']['frac'], r['kernels_ms_per_step'], d.get('psnr_vs_oracle_db',{}).get('rgb_map'))" || echo "variant [$1 $3] FAILED"; }
{ ab 0 "--netwidth 192 --use-viewdirs" generic_w192_views 0; ab 2 "--netwidth 192 --use-viewdirs" generic_w192_views 0; ab 0 "--use-viewdirs --bend-depth 7" config4_forced_generic 1; ab 2 "--use-viewdirs --bend-depth 7" config4_forced_generic 1; ab 2 "--use-viewdirs --bend-depth 7" config4_compiled 0; ab 2 "--netwidth 192" generic_w192 0; } > gpurun_out/c12/ab_gx16_views.txt 2>&1
grep "x16=" gpurun_out/c12/ab_gx16_views.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/c12/pytest_gpu_full.txt 2>&1; tail -4 gpurun_out/c12/pytest_gpu_full.txt
