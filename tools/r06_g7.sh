export NRNERF_PIN_RECORD=$PWD/gpurun_out/pins_r06.jsonl
rm -f $NRNERF_PIN_RECORD
python -m pytest tests/test_gpu_parity.py -q -x -s -k "generic_exact_viewdirs_192 or exact_viewdirs or boundary_contract" 2>&1 | tail -15 > gpurun_out/r06_exact_generic.txt
cat $NRNERF_PIN_RECORD >> gpurun_out/r06_exact_generic.txt
