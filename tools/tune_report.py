"""Per-shape view of the GEMM core from an autotuner log (run anything with TRIS_TUNE_LOG=<file>: one line per product shape
the first-encounter autotuner tuned, with the winner's idle-device time).  Prints the shapes ranked by the time they spend
above a reference rate (default 160 TFLOP/s fp32-equivalent, what the long-K products reach in x3 mode), i.e. where a
better tile / split-K / kernel would pay, next to an HBM floor for the operand and output bytes.
usage: python tools/tune_report.py <tune_log.txt> [reference_tflops] [top_n]
kind codes: ak 0 row-major A, 1 A^T (k-major), 2 im2col gather; bkind 0 B^T (n,k), 1 B (k,n), 2 mirrored-tap weights (dgrad),
3 im2col gather (wgrad), 4 pre-split planes; +16 = forward product with fused BatchNorm statistics."""
import re
import sys

ref = float(sys.argv[2]) if len(sys.argv) > 2 else 160.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
pat = re.compile(r"ak=(\d+) bkind=(\d+) M=(\d+) N=(\d+) K=(\d+) batch=(\d+) mode=(\d+) -> (\d+)x(\d+) sk=(\d+) nw=(\d+)(?: pipe=\d+)?\s+"
                 r"([\d.]+) us\s+([\d.]+) TFLOP")
rows, seen = [], set()
for line in open(sys.argv[1]):
    m = pat.match(line)
    if not m:
        continue
    ak, bk, M, N, K, batch, mode, bm, bn, sk, nw = (int(x) for x in m.groups()[:11])
    us, tf = float(m.group(12)), float(m.group(13))
    key = (ak, bk, M, N, K, batch, mode)
    if key in seen:
        continue
    seen.add(key)
    flop = 2.0 * M * N * K * batch
    ideal = flop / (ref * 1e12) * 1e6
    a_el = M * K / (9.0 if ak == 2 else 1.0)                     # implicit GEMM: K counts every tap, the gathered tensor is read once
    b_el = N * K / (9.0 if bk % 16 == 3 else 1.0)
    byt = 4.0 * batch * (M * N + a_el + b_el)
    rows.append((us - ideal, ak, bk, M, N, K, batch, f"{bm}x{bn}", sk, us, tf, byt / 5e6))
rows.sort(reverse=True)
print(f"{len(rows)} shapes, one call each: {sum(r[9] for r in rows) / 1e3:.2f} ms; above {ref:.0f} TFLOP/s: "
      f"{sum(max(r[0], 0) for r in rows) / 1e3:.2f} ms")
print(f"{'ak':>2} {'bk':>2} {'M':>8} {'N':>6} {'K':>8} {'b':>3} {'tile':>8} {'sk':>3} {'us':>8} {'TFLOP/s':>8} {'over ref us':>11} {'~HBM floor us':>13}")
for r in rows[:top]:
    print(f"{r[1]:>2} {r[2]:>2} {r[3]:>8} {r[4]:>6} {r[5]:>8} {r[6]:>3} {r[7]:>8} {r[8]:>3} {r[9]:>8.1f} {r[10]:>8.1f} {r[0]:>11.1f} {r[11]:>13.1f}")
