"""rocprofv3 kernel trace -> average duration per (kernel, workgroups): python scripts/trace_by_grid.py DIR [name filter]."""
import csv
import glob
import os
import re
import sys

rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if len(sys.argv) > 2 and sys.argv[2] not in name:
                continue
            short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
            wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r.get("Grid_Size_Y", 1) or 1)
            key = (short, wgs, int(r["Workgroup_Size_X"]))
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n, s, mn = rows.get(key, (0, 0.0, 1e30))
            rows[key] = (n + 1, s + d, min(mn, d))
print(f"{'kernel':48s} {'wgs':>5s} {'threads':>7s} {'calls':>7s} {'avg us':>9s} {'min us':>9s}")
for (k, w, t), (n, s, mn) in sorted(rows.items()):
    print(f"{k[:48]:48s} {w:5d} {t:7d} {n:7d} {s / n:9.2f} {mn:9.2f}")
