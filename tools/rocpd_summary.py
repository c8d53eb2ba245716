"""Summarise a rocprofv3 rocpd sqlite DB (--kernel-trace) into a per-kernel stats CSV (name, calls, total/avg/min/max us, %)."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
for r in rows:
    w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
print("kernels", len(rows), "total ms", tot / 1e6)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{100.0*r[2]/tot:6.2f}%  {r[1]:6d} calls  avg {r[3]/1e3:9.1f} us  total {r[2]/1e6:9.2f} ms  {r[0][:110]}")
