mkdir -p gpurun_out/fill; timeout 500 python -m pytest tests/test_gpu_sliced.py -x -q > gpurun_out/fill/pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/fill/pytest.log | tail -5; timeout 200 python scripts/fuzz_sliced.py 30 2>&1 | tail -1; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | grep -o "cold_csr_build_ms[^,]*"; python - <<P
import csv,glob
f=glob.glob("/tmp/st/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("sliced_sort","sliced_color","sliced_table","sliced_fill","csr_","radix_","scan_")):
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:28]
        print(short, r["Calls"], r["AverageNs"])
P
