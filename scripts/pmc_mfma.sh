#!/bin/bash
# MFMA-pipe utilisation of the sigmoid attention kernel (a2, and the segmented mode of DIFFormer_v2) from rocprofv3 PMC
# counters, one counter per pass (kernel-trace only):  util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
# (MFMA_BUSY counts 32 cycles per v_mfma_f32_16x16x4_f32 summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs).
# Run on the GPU box from the repo root; prints one line per (kernel, grid).
set -e
R=$PWD; OUT=$R/gpurun_out/pmc_mfma; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o p -- python $R/scripts/exp_sigmoid_scaling.py > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
def load(c):
    f = glob.glob(f"$OUT/{c}/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sigmoid_attn_kernel" in r["Kernel_Name"]:
            agg[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
m, g, b = load("SQ_VALU_MFMA_BUSY_CYCLES"), load("GRBM_GUI_ACTIVE"), load("SQ_BUSY_CYCLES")
res = {}
for k in sorted(m):
    res[k] = {"mfma_busy_cycles": m[k], "gui_active": g.get(k), "sq_busy_cycles": b.get(k),
              "mfma_util": m[k] / (g[k] / 8 * 1024) if g.get(k) else None}
    print(k, res[k])
json.dump(res, open("$OUT/mfma_util.json", "w"), indent=1)
PY
