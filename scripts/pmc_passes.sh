#!/bin/bash
# Generic rocprofv3 PMC collection: one pass per counter group (kernel-trace only, never combined with sys/runtime
# traces), then per-kernel averages as JSON.
#   scripts/pmc_passes.sh <tag> "<C1> <C2> ..." ["<C3> ..." ...] -- <command ...>
# Output: gpurun_out/pmc_<tag>/summary.json  (kernel -> counter -> mean over dispatches, plus dispatch count)
R=$PWD; TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for G in "${GROUPS_[@]}"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pass$i -o p -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i ($G) failed" >> $OUT/errors.log
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, dispatches=max(len(v) for v in cs.values())) for k, cs in agg.items()}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
