R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 1024 16384; do
rm -rf /tmp/tq && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tq -o s -- python $R/tools/train_step_profile.py $n bf16 > /tmp/tq.log 2>&1
db=$(find /tmp/tq -name "*.db" | head -1)
echo "=== $n: $(grep ms/step /tmp/tq.log)"
python $R/tools/rocprof_summary.py "$db" 2>/dev/null | grep -v "net_kernel by pass" | cut -c1-200 | head -60
done
