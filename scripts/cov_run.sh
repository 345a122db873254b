#!/bin/bash
# Kernel coverage of a subset of the GPU tests:  scripts/cov_run.sh <tag> <pytest args...>   (on the GPU box, from the repo root)
# -> gpurun_out/cov_<tag>/ (rocprofv3 kernel stats per process), gpurun_out/cov_<tag>_pytest.log
R=$PWD; T=$1; shift
mkdir -p $R/gpurun_out/cov_$T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cov_$T -- python -m pytest "$@" -q -p no:cacheprovider > $R/gpurun_out/cov_${T}_pytest.log 2>&1
echo rc=$? >> $R/gpurun_out/cov_${T}_pytest.log
cd $R
find gpurun_out/cov_$T -name "*kernel_trace.csv" -delete
grep -E "passed|failed|rc=" gpurun_out/cov_${T}_pytest.log | tail -3
