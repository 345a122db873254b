"""Start / end (us) and queue of the last kernels of a rocprofv3 --kernel-trace CSV:  python scripts/kernel_timeline.py trace.csv [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -70:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%9.1f %9.1f  q=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), name))
