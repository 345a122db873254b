#!/bin/bash
# End-of-round evidence on the final library:  scripts/profile_round6_final.sh [tag]  (GPU box, repo root) -> gpurun_out/<tag>/
#   headline bench + kernel stats + PMC traffic (profile_c4.sh); the image-and-text DIFFormer-a workloads (bench line + kernel stats);
#   the other configs (profile_configs.sh); the epoch scripts of the three training loops.
T=${1:-r06b}
R=$PWD; OUT=$R/gpurun_out/$T; mkdir -p $OUT
./scripts/profile_c4.sh $T > $OUT/c4.log 2>&1
for W in cifar15k-a-h300 stl13k-a-h400; do
  python bench.py --workload $W --steps 30 --warmup 5 > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  (cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o s -- python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-exact-pass > $OUT/stats_$W.log 2>&1)
  find $OUT/stats_$W -name "*kernel_stats.csv" -exec cp {} $OUT/${W}_kernel_stats.csv \;
  rm -rf $OUT/stats_$W
done
./scripts/profile_configs.sh ${T}_cfg > $OUT/cfg.log 2>&1
python scripts/st_epoch.py --epochs 5 2>&1 | grep -v "Warn\|return float\|amdgpu\|Consider" > $OUT/st_epoch.txt
python scripts/it_epoch.py 2>&1 | grep -v amdgpu > $OUT/it_epoch.txt
python scripts/nc_batch_epoch.py 2>&1 | grep -v amdgpu > $OUT/nc_batch_epoch.txt
tail -3 $OUT/c4.log | cut -c1-300; cut -c1-300 $OUT/bench_cifar15k-a-h300.json; tail -14 $OUT/cfg.log; cat $OUT/st_epoch.txt $OUT/it_epoch.txt $OUT/nc_batch_epoch.txt
