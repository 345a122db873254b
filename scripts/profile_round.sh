#!/bin/bash
# Everything the round's tracked evidence comes from, in one GPU call:  scripts/profile_round.sh <tag>
#   bench line + rocprofv3 kernel stats + PMC passes of the headline workload, the other configs (eager / hipGraph), the
#   Zipf-degree graph, the Pokec mini-batch pass, the training step, the sigmoid backward.  Results: gpurun_out/<tag>*/
T=${1:-r05e}
set -x
./scripts/profile_c4.sh $T > gpurun_out/${T}_c4.log 2>&1
./scripts/profile_configs.sh ${T}_cfg > gpurun_out/${T}_cfg.log 2>&1
python bench.py --workload ogbn-proteins-zipf-s --no-cpu-baseline > gpurun_out/$T/bench_zipf.json 2> gpurun_out/$T/bench_zipf.err
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zipf_stats -o z -- python $OLDPWD/bench.py --workload ogbn-proteins-zipf-s --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1; find /tmp/zipf_stats -name "*kernel_stats.csv" -exec cp {} $OLDPWD/gpurun_out/$T/zipf_kernel_stats.csv \;)
python scripts/exp_pokec_epoch.py > gpurun_out/$T/pokec_epoch.log 2>&1
python scripts/exp_train_step.py > gpurun_out/$T/train_step.log 2>&1
python scripts/exp_sigmoid_bwd.py > gpurun_out/$T/sigmoid_bwd.log 2>&1
python scripts/exp_sliced_shard.py > gpurun_out/$T/sliced_shard.log 2>&1
# round 5: the spatial-temporal epochs (whole-model tiny kernels / layer path), bfloat16 scaling, hub rows and node order
python scripts/st_epoch.py --epochs 3 > gpurun_out/$T/st_epoch_tiny.log 2>&1
DIFFORMER_TINY=0 python scripts/st_epoch.py --epochs 3 > gpurun_out/$T/st_epoch_layers.log 2>&1
python scripts/exp_c5_bf16.py > gpurun_out/$T/c5_bf16.log 2>&1
python scripts/exp_bf16_scaling.py > gpurun_out/$T/bf16_scaling.log 2>&1
python scripts/exp_regional_order.py > gpurun_out/$T/regional_order.log 2>&1
tail -5 gpurun_out/${T}_c4.log; tail -14 gpurun_out/${T}_cfg.log; cut -c1-300 gpurun_out/$T/bench_zipf.json; tail -8 gpurun_out/$T/pokec_epoch.log; tail -4 gpurun_out/$T/train_step.log; tail -4 gpurun_out/$T/sigmoid_bwd.log; tail -9 gpurun_out/$T/sliced_shard.log
