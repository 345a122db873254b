#!/bin/bash
# the layer kernel leaves the Gram partials of its OUTPUT (DIFFORMER_CHAIN_GRAM=1) on the configs whose
# forward is a chain of short launches; same box, plain model calls (auto-captured hipGraph replay)
for W in cifar50k-s pokec-batch-s pokec-batch-s-bf16 cora-s; do
  for F in 0 1; do
    DIFFORMER_CHAIN_GRAM=$F python bench.py --workload $W --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$W', 'CHAIN_GRAM=$F', round(d['ms_per_step'],4), 'ms', d['config']['launch'])"
  done
done
