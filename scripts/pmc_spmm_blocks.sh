#!/bin/bash
# FETCH_SIZE of the blocked SpMM vs number of source blocks (C4 graph).  Run on the GPU box from the repo root.
R=$PWD; OUT=$R/gpurun_out/pmc_blocks; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o blocks -- python $R/scripts/exp_spmm_blocks.py "$@" 2>&1 | grep n_blocks
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "spmm_blocked" in r["Kernel_Name"] or "spmm_wave_row" in r["Kernel_Name"]]
# dispatches come in groups of 12 per n_blocks value (2 warm-up + 10 timed)
vals = [float(r["Counter_Value"]) for r in rows]
for i in range(0, len(vals), 12):
    g = vals[i:i + 12]
    print(f"group {i // 12}: FETCH_SIZE mean {sum(g) / len(g) / 1e6:.3f} GB raw (x2 corrected {2 * sum(g) / len(g) / 1e6:.3f} GB)")
PY
