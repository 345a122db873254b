#!/bin/bash
# HBM traffic of the dominant kernel from rocprofv3 PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE,
# kernel-trace only), plus the calibration of both counters on known byte counts.  Run on the GPU box from the repo root.
set -e
R=$PWD; OUT=$R/gpurun_out/pmc_traffic; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 $R/scripts/pmc_calibrate.hip -o /tmp/pmc_cal
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cal_$C -o cal -- /tmp/pmc_cal > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/bench_$C -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
def load(d):
    f = glob.glob(f"$OUT/{d}/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for d in (f"cal_{C}", f"bench_{C}"):
        for (k, c), v in load(d).items():
            if any(s in k for s in ("read_dword", "write_dwordx4", "spmm", "project_reduce", "simple_apply", "skinny_linear")):
                res.setdefault(k, {})[c] = v
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/pmc_traffic_raw.json", "w"), indent=1)
PY
