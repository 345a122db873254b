#!/bin/bash
# Round-N evidence for the headline workload: bench line, rocprofv3 kernel stats of the same command, PMC traffic.
#   scripts/profile_c4.sh <tag>      (run on the GPU box from the repo root; results under gpurun_out/<tag>/)
R=$PWD; TAG=${1:-r02}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c4 -- python $R/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/c4_kernel_stats.csv $OUT/bench_c4_kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_c4_kernel_stats.csv \;
cd $R
./scripts/pmc_passes.sh ${TAG}_traffic "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_${TAG}_traffic/summary.json $OUT/pmc_summary.json
cat $OUT/bench_c4.json; head -12 $OUT/bench_c4_kernel_stats.csv
