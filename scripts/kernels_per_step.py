"""Kernel list of a training step from a rocprofv3 kernel_stats.csv:  python scripts/kernels_per_step.py <csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
for r in rows[:48]:
    c = int(r["Calls"])
    print("%6.1f/step %7.1f us %8.1f us/step  %s" % (c / steps, float(r["AverageNs"]) / 1000, float(r["TotalDurationNs"]) / steps / 1000, r["Name"][:110]))
print(sum(int(r["Calls"]) for r in rows) / steps, "kernels/step", sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1000, "us/step")
