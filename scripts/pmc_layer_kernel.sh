#!/bin/bash
# LDS bank-conflict counters of the closed-form layer kernel for two builds (GPU box, repo root):
#   scripts/pmc_layer_kernel.sh <out-dir> [lib ...]      ("" = the in-tree build)
OUT=$PWD/gpurun_out/${1:-pmc_layer}; shift; mkdir -p $OUT; R=$PWD
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "${lib:-default}" .so)
  DIFFORMER_HIP_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$tag -o p -- python $R/scripts/exp_layer_kernel.py > $OUT/$tag.log 2>&1
  python - "$OUT/$tag" "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "simple_layer_kernel" not in k: continue
    short = k[k.find("simple_layer_kernel"):][:60]
    agg[short][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(short, r["Counter_Name"])] += 1
for k, v in agg.items():
    c, a = v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 1)
    print(f"{sys.argv[2]}: {k}: SQ_LDS_BANK_CONFLICT {c:.3g} / SQ_LDS_IDX_ACTIVE {a:.3g} = {100 * c / a:.1f} %")
PY
done
