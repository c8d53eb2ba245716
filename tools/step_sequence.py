"""Time-ordered kernel sequence of ONE steady-state step in a rocprofv3 --kernel-trace rocpd DB.
usage: step_sequence.py <results.db> [steps_back=2] [from_us to_us]
  -> per kernel: stream, offset from the step's start (us), duration, idle gap in front of it on its stream, name; then, for the
     window, the GPU's busy time (union over streams) and the kernel time by name.  The step is delimited by the end of the last
     optimiser launch of the step before (launches of one step are clustered by the gaps between them, as in stream_gaps.py)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo = float(sys.argv[3]) if len(sys.argv) > 4 else None
hi = float(sys.argv[4]) if len(sys.argv) > 4 else None
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
scol = "stream_id" if "stream_id" in cols else "queue_id"
ends = [r[0] for r in cur.execute("select end from kernels where name like '%adamw%kernel%' order by start").fetchall()]
marks = [e for i, e in enumerate(ends) if i + 1 == len(ends) or ends[i + 1] - e > 3_000_000]
t0, t1 = marks[-back - 1], marks[-back]
rows = cur.execute(f"select {scol}, start, end, name from kernels where start > {t0} and end <= {t1} order by start").fetchall()
print(f"step of {(t1 - t0) / 1e6:.2f} ms, {len(rows)} kernels")
last = {}
sel = []
for s, a, b, n in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    n = n.split("(")[0][:100]
    gap = (a - last[s]) / 1e3 if s in last else 0.0
    last[s] = b
    off = (a - t0) / 1e3
    if lo is not None and not (lo <= off < hi):
        continue
    sel.append((s, a, b, n))
    print(f"s{s} {off:9.1f} {(b - a) / 1e3:8.1f} {gap:8.1f}  {n}")
if sel:
    ev = sorted((a, b) for _, a, b, _ in sel)
    busy, cs, ce = 0, ev[0][0], ev[0][1]
    for a, b in ev[1:]:
        if a > ce:
            busy += ce - cs; cs, ce = a, b
        else:
            ce = max(ce, b)
    busy += ce - cs
    span = max(b for _, _, b, _ in sel) - min(a for _, a, _, _ in sel)
    print(f"# window: span {span / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, kernel time summed {sum(b - a for _, a, b, _ in sel) / 1e3:.1f} us")
    by = {}
    for s, a, b, n in sel:
        e = by.setdefault((s, n), [0, 0]); e[0] += 1; e[1] += b - a
    for (s, n), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"#   s{s} {t / 1e3:9.1f} us {c:4d}  {n}")
