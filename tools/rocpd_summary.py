"""Summarise a rocprofv3 rocpd sqlite DB (--kernel-trace) into a per-kernel stats CSV (name, calls, total/avg/min/max, %).

usage: rocpd_summary.py <results.db> <out.csv> [top_n] [steady_steps]
`steady_steps` > 0 keeps only the last N occurrences of the step delimiter kernel (adamw_kernel, launched once per
parameter group per step, two groups) -- i.e. the steady state, without warm-up and the GEMM autotuner's candidate runs."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
steady = int(sys.argv[4]) if len(sys.argv) > 4 else 0
where = ""
if steady > 0:
    marks = [r[0] for r in cur.execute("select end from kernels where name like '%adamw_kernel%' order by start").fetchall()]
    per_step = 2
    if len(marks) >= per_step * (steady + 1):
        t0 = marks[-per_step * steady - 1]
        t1 = marks[-1]
        where = f" where start > {t0} and end <= {t1}"
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels"
                   + where + " group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
for r in rows:
    w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 3)])
print("kernels", len(rows), "total ms", tot / 1e6, "(steady: last %d steps)" % steady if where else "")
for r in rows[:top]:
    print(f"{100.0*r[2]/tot:6.2f}%  {r[1]:6d} calls  avg {r[3]/1e3:9.1f} us  total {r[2]/1e6:9.2f} ms  {r[0][:110]}")
