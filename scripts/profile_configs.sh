#!/bin/bash
# Bench lines (plain calls = the model's own choice of replay / eager, and DIFFORMER_AUTO_GRAPH=0) + rocprofv3 kernel stats for the BASELINE configs other than C4.
#   scripts/profile_configs.sh <tag>   (GPU box, repo root) -> gpurun_out/<tag>/
R=$PWD; TAG=${1:-r02cfg}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for W in cora-s cora-a cifar50k-s pokec-batch-s-bf16 pokec-batch-s pokec-batch-h128 cifar50k-h300 pokec-full-s pokec-full-h128 ogbn-proteins-blocks-s ogbn-proteins-zipf-blocks-s; do
  # the eager line carries the cpu_baseline leg (oracle port on the host cores; skipped by bench.py for bf16 storage)
  python bench.py --workload $W --steps 50 --warmup 5 > $OUT/bench_${W}_eager.json 2>> $OUT/err.log
  DIFFORMER_AUTO_GRAPH=0 python bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-exact-pass > $OUT/bench_${W}_nograph.json 2>> $OUT/err.log
  (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o s -- python $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-exact-pass > $OUT/stats_$W.log 2>&1)
  find $OUT/stats_$W -name "*kernel_stats.csv" -exec cp {} $OUT/${W}_kernel_stats.csv \;
done
for f in $OUT/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['config']['workload'], d['config']['launch'], round(d['ms_per_step'],4), 'ms', d['ms_per_step_events'])"; done
