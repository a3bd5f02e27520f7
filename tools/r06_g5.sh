R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kb -o k -- python $R/tools/gen_train_kernels_bench.py > /tmp/kb.log 2>&1
tail -3 /tmp/kb.log
python $R/tools/rocprof_summary.py $(find /tmp/kb -name "*.db" | head -1) 2>&1 | head -40
