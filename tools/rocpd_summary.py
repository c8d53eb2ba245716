"""Summarise a rocprofv3 rocpd sqlite DB (--kernel-trace) into a per-kernel stats CSV (name, calls, total/avg/min/max, %).

usage: rocpd_summary.py <results.db> <out.csv> [top_n] [steady_steps]
`steady_steps` > 0 keeps only the last N steps (delimited by the optimiser's launches, see step_marks) -- i.e. the steady
state, without warm-up and the GEMM autotuner's candidate runs."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
steady = int(sys.argv[4]) if len(sys.argv) > 4 else 0
where = ""
def step_marks(cur):
    """end time of the LAST optimiser launch of every step: the optimiser's launches (adamw_kernel eager, adamw_dev_kernel replayed;
    2-4 per step, all within a fraction of a millisecond at the step's end) are clustered by the gaps between them"""
    ends = [r[0] for r in cur.execute("select end from kernels where name like '%adamw%kernel%' order by start").fetchall()]
    marks = []
    for i, e in enumerate(ends):
        if i + 1 == len(ends) or ends[i + 1] - e > 3_000_000:
            marks.append(e)
    return marks
if steady > 0:
    marks = step_marks(cur)
    if len(marks) >= steady + 1:
        t0 = marks[-steady - 1]
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
