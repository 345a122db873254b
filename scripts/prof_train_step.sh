#!/bin/bash
# rocprofv3 kernel stats of the C4 training step (GPU box, repo root): top kernels per step -> gpurun_out/<tag>/
R=$PWD; OUT=$R/gpurun_out/${1:-train_prof}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- python $R/scripts/exp_train_step.py > $OUT/train_step.log 2>&1
cp $(find /tmp/ts -name "*kernel_stats.csv") $OUT/train_step_kernel_stats.csv
python - $OUT/train_step_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms  {r['Calls']:>5} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:130]}")
PY
