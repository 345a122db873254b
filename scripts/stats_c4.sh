#!/bin/bash
# rocprofv3 kernel stats of the headline bench command -> gpurun_out/<tag>_kernel_stats.csv (top rows printed)
R=$PWD; TAG=${1:-r04}; shift
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/stats_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -o s -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/stats_$TAG.log 2>&1
find /tmp/stats_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:64].ljust(64), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:9.1f} us', r["Percentage"].rjust(7))
PY
