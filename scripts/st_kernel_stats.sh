#!/bin/bash
# rocprofv3 kernel stats of spatial-temporal epochs (scripts/st_host_profile.py): <tag> then runs "name kernel n d deg T mode"
T=${1:-st}
cd /tmp; export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf /tmp/st_stats_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_stats_$name -o st -- python $GRAFT_REPO_ROOT/scripts/st_host_profile.py "$@" > /dev/null 2>&1
  find /tmp/st_stats_$name -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${T}_${name}_kernel_stats.csv \;
  echo "== $name"; head -8 $GRAFT_REPO_ROOT/gpurun_out/${T}_${name}_kernel_stats.csv | cut -c1-60,200-
}
run chickenpox_simple simple 20 4 4 104 cum
run covid_sigmoid sigmoid 129 8 12 50 cum
run wikimath_simple simple 1068 14 10 146 inc
run wikimath_sigmoid sigmoid 1068 14 10 40 inc
