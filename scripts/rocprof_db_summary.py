"""Dump the kernel-stats summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    w.writerow([name, calls, f"{total:.2f}", f"{avg:.2f}", f"{pct:.2f}"])
