#!/bin/bash
# rocprofv3 kernel stats of the training loops at the scripts' configurations -> gpurun_out/r06b_loops/*.csv  (GPU box, repo root)
R=$PWD; OUT=$R/gpurun_out/r06b_loops; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run() { # name, command...
  local name=$1; shift
  rm -rf /tmp/lp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$name -o s -- "$@" > $OUT/$name.log 2>&1
  find /tmp/lp_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf /tmp/lp_$name
}
run nc_proteins python $R/scripts/nc_batch_epoch.py proteins
run nc_pokec python $R/scripts/nc_batch_epoch.py pokec
run it_cifar10_simple python $R/scripts/it_epoch.py --only cifar10 --kernel simple --train-only --epochs 20
run it_cifar10_sigmoid python $R/scripts/it_epoch.py --only cifar10 --kernel sigmoid --train-only --epochs 20
run st_wikimath python $R/scripts/st_epoch.py --only wikimath --epochs 2
run st_chickenpox python $R/scripts/st_epoch.py --only chickenpox --epochs 3
run pp_step python $R/scripts/pp_step.py
ls -la $OUT | head -20
