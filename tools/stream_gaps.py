"""Per-stream busy time and idle gaps of the steady-state steps in a rocprofv3 --kernel-trace rocpd DB.
usage: stream_gaps.py <results.db> [steady_steps]   -> per stream: busy ms/step, span, number of kernels, and for the busiest
stream the distribution of idle gaps between consecutive kernels (where a captured graph or fewer launches would pay)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steady = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
def step_marks(cur):
    """end time of the LAST optimiser launch of every step: the optimiser's launches (adamw_kernel eager, adamw_dev_kernel replayed;
    2-4 per step, all within a fraction of a millisecond at the step's end) are clustered by the gaps between them"""
    ends = [r[0] for r in cur.execute("select end from kernels where name like '%adamw%kernel%' order by start").fetchall()]
    marks = []
    for i, e in enumerate(ends):
        if i + 1 == len(ends) or ends[i + 1] - e > 3_000_000:
            marks.append(e)
    return marks
marks = step_marks(cur)
steady = min(steady, len(marks) - 1)
t0, t1 = marks[-steady - 1], marks[-1]
rows = cur.execute(f"select {scol}, start, end, name from kernels where start > {t0} and end <= {t1} order by start").fetchall()
wall = (t1 - t0) / 1e6 / steady
by = {}
for s, a, b, n in rows:
    by.setdefault(s, []).append((a, b, n))
print(f"wall {wall:.2f} ms/step over {steady} steps; streams by {scol}")
# union busy time over all streams
ev = sorted((a, b) for _, a, b, _ in rows)
busy = 0; ce = None; cs = None
for a, b in ev:
    if ce is None or a > ce:
        if ce is not None: busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
print(f"GPU busy (union of all streams) {busy/1e6/steady:.2f} ms/step = {100*busy/(t1-t0):.1f}% of wall")
main = max(by, key=lambda s: sum(b - a for a, b, _ in by[s]))
for s, ks in sorted(by.items(), key=lambda kv: -sum(b - a for a, b, _ in kv[1])):
    bt = sum(b - a for a, b, _ in ks)
    print(f"  stream {s}: {len(ks)/steady:7.1f} kernels/step  busy {bt/1e6/steady:7.2f} ms/step")
ks = by[main]
gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
gaps = [g for g in gaps if g > 0]
import statistics
print(f"main stream {main}: {len(gaps)/steady:.0f} gaps/step, total idle {sum(gaps)/1e6/steady:.2f} ms/step, median {statistics.median(gaps)/1e3:.2f} us")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    sel = [g for g in gaps if lo <= g < hi]
    print(f"   gaps {lo/1e3:6.0f}-{hi/1e3:<8.0f} us: {len(sel)/steady:7.1f}/step  {sum(sel)/1e6/steady:6.2f} ms/step")
big = sorted(((ks[i + 1][0] - ks[i][1], ks[i][2][:60], ks[i + 1][2][:60]) for i in range(len(ks) - 1)), reverse=True)[:12]
for g, a, b in big:
    print(f"   {g/1e3:8.1f} us between {a} -> {b}")
# per-stream time by kernel (what the critical stream is made of)
import re
for s, ks in sorted(by.items(), key=lambda kv: -sum(b - a for a, b, _ in kv[1])):
    agg = {}
    for a, b, n in ks:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = n.split("(")[0][:70] if "gemm_fast_kernel" not in n else n.split("(")[0][:90]
        e = agg.setdefault(n, [0, 0])
        e[0] += 1
        e[1] += b - a
    print(f"stream {s}: kernel time by name (ms/step, launches/step)")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {t/1e6/steady:7.3f} ms  {c/steady:6.1f}  {n}")
