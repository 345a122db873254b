#!/bin/bash
# The part of scripts/profile_round.sh that a late change of the sigmoid / row-GEMM / Linear kernels touches (no PMC passes, no
# node-order / scaling experiments):  scripts/profile_round_delta.sh <tag>  -> gpurun_out/<tag>/, gpurun_out/<tag>_cfg/
T=${1:-r05e}; R=$PWD; mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/$T/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.txt 2>&1
python bench.py > gpurun_out/$T/bench_c4.json 2> gpurun_out/$T/bench_c4.err
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats -o c4 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/$T/stats.log 2>&1)
find gpurun_out/$T/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/$T/bench_c4_kernel_stats.csv \;
python scripts/exp_train_step.py > gpurun_out/$T/train_step.log 2>&1
python scripts/exp_train_step.py h128 >> gpurun_out/$T/train_step.log 2>&1
python scripts/exp_sigmoid_bwd.py > gpurun_out/$T/sigmoid_bwd.log 2>&1
DIFFORMER_SIGMOID_BWD_SPLIT=1 python scripts/exp_sigmoid_bwd.py 2>&1 | grep default >> gpurun_out/$T/sigmoid_bwd.log
python scripts/exp_c5_bf16.py > gpurun_out/$T/c5_bf16.log 2>&1
./scripts/profile_configs.sh ${T}_cfg > gpurun_out/${T}_cfg.log 2>&1
tail -3 gpurun_out/$T/pytest_gpu.txt; tail -2 gpurun_out/$T/smoke.txt; cut -c1-260 gpurun_out/$T/bench_c4.json; tail -14 gpurun_out/${T}_cfg.log
