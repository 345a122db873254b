#!/bin/bash
# Round-6 evidence in one GPU call:  scripts/profile_round6.sh [tag]     (GPU box, repo root) -> gpurun_out/<tag>/
#   headline bench + kernel stats + PMC traffic (profile_c4.sh); the image-and-text DIFFormer-a workloads (bench line, rocprofv3
#   kernel stats of the same command, PMC counters of the sweep kernel); forward / backward times of the wide sigmoid kernels
#   against the paths they replace; the slot-order experiment of the sliced product; the other configs (profile_configs.sh).
T=${1:-r06a}
R=$PWD; OUT=$R/gpurun_out/$T; mkdir -p $OUT
set -x
./scripts/profile_c4.sh $T > $OUT/c4.log 2>&1
for W in cifar15k-a-h300 stl13k-a-h400; do
  python bench.py --workload $W --steps 30 --warmup 5 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o s -- python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-exact-pass > $OUT/stats_$W.log 2>&1)
  find $OUT/stats_$W -name "*kernel_stats.csv" -exec cp {} $OUT/${W}_kernel_stats.csv \;
done
./scripts/pmc_passes.sh ${T}_sigw "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" -- python $R/scripts/exp_sigmoid_wide_fwd.py 15000 300 bwd > $OUT/pmc_sigw.log 2>&1
cp gpurun_out/pmc_${T}_sigw/summary.json $OUT/pmc_sigw_summary.json
python scripts/exp_sigmoid_wide.py > $OUT/sigmoid_wide_times.txt 2>&1
python scripts/exp_sigmoid_wide.py --old >> $OUT/sigmoid_wide_times.txt 2>&1
python scripts/exp_slot_order.py > $OUT/slot_order.txt 2>&1
./scripts/profile_configs.sh ${T}_cfg > $OUT/cfg.log 2>&1
tail -3 $OUT/c4.log | cut -c1-400; cut -c1-600 $OUT/bench_cifar15k-a-h300.json; cat $OUT/sigmoid_wide_times.txt | grep "N="; grep -v "^$" $OUT/slot_order.txt | tail -4; tail -14 $OUT/cfg.log
