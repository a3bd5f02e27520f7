#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + separate PMC passes of the bench command.
# Outputs under gpurun_out/prof_<tag>/ ; summarise afterwards with tools/rocprof_summary.py and tools/pmc_summary.py.
TAG=${1:-final}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- $CMD > $OUT/stats.log 2>&1
PMC="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-train-step --min-gpu-seconds 0"
timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmcA -o a -- $PMC > $OUT/pmcA.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcB -o b -- $PMC > $OUT/pmcB.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA --output-format csv -d $OUT/pmcC -o c -- $PMC > $OUT/pmcC.log 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmcD -o d -- $PMC > $OUT/pmcD.log 2>&1
grep -h '"metric"' $OUT/stats.log | cut -c1-160
ls $OUT
# summaries (the sqlite database itself is large and stays on the box)
cd $R
DB=$(find $OUT/stats -name "*.db" | head -1)
python tools/rocprof_summary.py "$DB" > gpurun_out/${TAG}_kernel_stats.txt 2>&1
NRNERF_PROFILE_TAG=$TAG python tools/pmc_summary.py $OUT/pmcA $OUT/pmcB $OUT/pmcC $OUT/pmcD > gpurun_out/${TAG}_pmc_summary.txt 2>&1
cp profiles/${TAG}_pmc_fine.json gpurun_out/${TAG}_pmc_fine.json    # pmc_summary.py writes it in place
find $OUT -name "*.db" -delete
