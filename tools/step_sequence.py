"""Time-ordered kernel sequence of ONE steady-state step in a rocprofv3 --kernel-trace rocpd DB (the step before the last AdamW).
usage: step_sequence.py <results.db> [steps_back=2]  -> per stream: offset from the step's start (us), duration, idle gap in front, name."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
scol = "stream_id" if "stream_id" in cols else "queue_id"
marks = [r[0] for r in cur.execute("select end from kernels where name like '%adamw%kernel%' order by start").fetchall()]
t0, t1 = marks[-2 * back - 1], marks[-2 * back + 1]
rows = cur.execute(f"select {scol}, start, end, name from kernels where start > {t0} and end <= {t1} order by start").fetchall()
print(f"step of {(t1 - t0) / 1e6:.2f} ms, {len(rows)} kernels")
last = {}
for s, a, b, n in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    n = n.split("(")[0][:100]
    gap = (a - last[s]) / 1e3 if s in last else 0.0
    last[s] = b
    print(f"s{s} {(a - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f} {gap:8.1f}  {n}")
