#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM-traffic PMC passes of the native training step with the
# shipped recipe (tools/train_step_profile.py), at 1024 and 16384 rays.  Outputs gpurun_out/<tag>_train_*; every file is
# stamped with the hash of ALL device sources (bench.kernel_source_sha16(train=True)) so that a stale profile is detectable.
TAG=${1:-r03}
R=$PWD
cd /tmp && export TMPDIR=/tmp
SHA=$(cd $R && python -c "import bench; print(bench.kernel_source_sha16(True))")
for n in 1024 16384; do
    out=/tmp/tprof_$n
    rm -rf $out
    timeout 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python $R/tools/train_step_profile.py $n bf16 > $out.stats.log 2>&1
    db=$(find $out/stats -name "*.db" | head -1)
    (echo "# device sources sha16 (train=True): $SHA"; echo "# rocprofv3 --kernel-trace --stats -- python tools/train_step_profile.py $n bf16   ($(grep 'ms/step' $out.stats.log | tail -1))"; python $R/tools/rocprof_summary.py "$db" | grep -v "net_kernel by pass") > $R/gpurun_out/${TAG}_train_kernel_stats_$n.txt 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcF -o f -- python $R/tools/train_step_profile.py $n bf16 > $out.f.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcW -o w -- python $R/tools/train_step_profile.py $n bf16 > $out.w.log 2>&1
    (echo "# device sources sha16 (train=True): $SHA"; python $R/tools/train_pmc_summary.py $out/pmcF $out/pmcW $n) > $R/gpurun_out/${TAG}_train_pmc_$n.txt 2>&1
    rm -rf $out
done
ls -la $R/gpurun_out/${TAG}_train_*
