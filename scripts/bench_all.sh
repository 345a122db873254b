#!/bin/bash
# One bench line per BASELINE config (and the extra workloads given as arguments) -> gpurun_out/<tag>_bench_<workload>.json + a summary
TAG=${1:-r04}; shift
mkdir -p gpurun_out
for W in ogbn-proteins-s cora-s cora-a cifar50k-s pokec-batch-s-bf16 pokec-batch-s "$@"; do
  python bench.py --workload $W --no-cpu-baseline > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err || tail -3 gpurun_out/${TAG}_bench_$W.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f'{d["config"]["workload"]:24s} {d["ms_per_step"]:.4f} ms  {d["value"] / 1e6:8.1f} M nodes/s  {r["entry_point"]} [{r["bound"]}] '
          f'{(r.get("achieved") or 0):.3g} {r.get("unit")} frac {(r.get("frac") or 0):.3f} share {(r.get("share_of_forward") or 0):.2f} launch {(r.get("avg_launch_ms") or 0) * 1e3:.1f} us')
PY
