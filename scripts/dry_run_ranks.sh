#!/bin/bash
# The driver's N-GPU command line with every rank on this box's ONE GPU and the collectives over gloo (DIFFORMER_BENCH_ONE_GPU=1):
# exercises the N-rank code path end to end at the full C4 size; the timings mean nothing.   scripts/dry_run_ranks.sh 8 [row|slice]
N=${1:-8}; PRODUCT=${2:-row}
DIFFORMER_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 3 --warmup 1 --shard-product $PRODUCT 2> gpurun_out/dry_${N}_${PRODUCT}.err | grep "^{" > gpurun_out/dry_${N}_${PRODUCT}.json
python - <<PY
import json
d = json.load(open("gpurun_out/dry_${N}_${PRODUCT}.json"))
print("n_gpus", d["n_gpus"], d["config"]["parallelism"], "ms_per_step", round(d["ms_per_step"], 3), "(one GPU, gloo: not a measurement)")
for p in d.get("per_rank_phases", []):
    print("  rank", p["rank"], "rows", p["rows"], {k: round(v, 3) for k, v in p["kernels_ms"].items()}, {k: round(v, 3) for k, v in p["collectives_ms"].items()})
PY
