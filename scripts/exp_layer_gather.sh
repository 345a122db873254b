#!/bin/bash
# In-kernel aggregation of the closed-form layer (dif_simple_layer_gather_*) against SpMM launch + layer kernel:
# parity tests, then the sparse configs with the switch on and off.   GPU box, repo root -> gpurun_out/gather/
R=$PWD; OUT=$R/gpurun_out/gather; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_closed_form.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
for W in pokec-batch-s pokec-batch-s-bf16 cora-s; do
  for G in 1 0; do
    DIFFORMER_LAYER_GATHER=$G timeout 300 python bench.py --workload $W --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_${W}_g$G.json 2>> $OUT/err.log
    DIFFORMER_LAYER_GATHER=$G DIFFORMER_AUTO_GRAPH=0 timeout 300 python bench.py --workload $W --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_${W}_g${G}_eager.json 2>> $OUT/err.log
  done
done
for W in pokec-batch-s pokec-batch-s-bf16; do
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o s -- python $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1)
find $OUT/stats_$W -name "*kernel_stats.csv" -exec cp {} $OUT/${W}_kernel_stats.csv \;
done
python scripts/exp_layer_gather.py > $OUT/kernel_times.txt 2>&1; cat $OUT/kernel_times.txt
tail -3 $OUT/pytest.log
for f in $OUT/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],4), 'ms', d['ms_per_step_events'])"; done
for W in pokec-batch-s pokec-batch-s-bf16; do head -8 $OUT/${W}_kernel_stats.csv | cut -c1-130,200-330; done
