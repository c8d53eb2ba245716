"""Per-shape view of the GEMM core from an autotuner log (run anything with TRIS_TUNE_LOG=<file>, or ops.set_option("TUNE_LOG",
file): one line per product shape the first-encounter autotuner tuned, with the winner's idle-device time).  Prints the shapes
ranked by the time they spend above a reference rate (default 160 TFLOP/s fp32-equivalent), i.e. where a better tile / split-K /
kernel would pay, next to an HBM floor for the operand and output bytes.
usage: python tools/tune_report.py <tune_log.txt> [reference_tflops] [top_n]
kind codes: ak 0 row-major A, 1 A^T (k-major), 2 im2col gather; bkind 0 B^T (n,k), 1 B (k,n), 2 mirrored-tap weights (dgrad),
3 im2col gather (wgrad); +16 = forward product with fused BatchNorm statistics, +32 = fused BatchNorm-backward reduction.
`idle_rates(path)` (bench.py): {kind: TFLOP/s} -- total FLOPs / total idle-device time of the tuned shapes per kind."""
import re
import sys

PAT = re.compile(r"ak=(\d+) bkind=(\d+) M=(\d+) N=(\d+) K=(\d+) batch=(\d+) mode=(\d+) -> (\d+)x(\d+) sk=(\d+)(?: nw=\d+)?(?: pipe=\d+)?\s+"
                 r"([\d.]+) us\s+([\d.]+) TFLOP")
PAT_CONV = re.compile(r"conv3x3 bkind=(\d+) M=(\d+) N=(\d+) K=(\d+) HxW=\d+x\d+ -> \w+ \d+\s+([\d.]+) us")
PAT_WG = re.compile(r"wgrad3x3 Cout=(\d+) Cin=(\d+) pixels=(\d+) HxW=\d+x\d+ -> \w+ \d+\s+([\d.]+) us")


def parse(path):
    rows, seen = [], set()
    for line in open(path):
        m = PAT.match(line)
        if not m:
            continue
        ak, bk, M, N, K, batch, mode, bm, bn, sk = (int(x) for x in m.groups()[:10])
        us, tf = float(m.group(11)), float(m.group(12))
        key = (ak, bk, M, N, K, batch, mode)
        if key in seen:
            continue
        seen.add(key)
        rows.append(dict(ak=ak, bk=bk, M=M, N=N, K=K, batch=batch, mode=mode, tile=f"{bm}x{bn}", sk=sk, us=us, tflops=tf))
    return rows


def idle_rates(path):
    """{kind: idle-device TFLOP/s of the tuner's winners}; kinds as in bench.py's by_kind where they can be told apart"""
    acc = {}

    def add(kind, flop, us):
        e = acc.setdefault(kind, [0.0, 0.0])
        e[0] += flop
        e[1] += us
    seen = set()
    for line in open(path):
        m = PAT_CONV.match(line)
        if m:
            bk, M, N, K, us = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
            if ("c", bk, M, N, K) in seen:
                continue
            seen.add(("c", bk, M, N, K))
            kind = "conv3x3_dgrad_bnbwd" if bk % 16 == 2 and bk >= 64 else "conv3x3_dgrad" if bk % 16 == 2 else "conv3x3_fwd"
            add(kind, 2.0 * M * N * K, us)
            continue
        m = PAT_WG.match(line)
        if m:
            co, ci, px, us = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
            if ("w", co, ci, px) in seen:
                continue
            seen.add(("w", co, ci, px))
            add("conv3x3_wgrad", 2.0 * co * 9 * ci * px, us)
    for r in parse(path):
        if r["ak"] == 2 or r["bk"] % 16 == 3:
            continue                      # (3x3 products are counted from their own lines: direct or implicit, whichever won)
        add("gemm_bnbwd" if r["bk"] >= 32 else "gemm", 2.0 * r["M"] * r["N"] * r["K"] * r["batch"], r["us"])
    return {k: round(v[0] / (v[1] * 1e-6) / 1e12, 1) for k, v in acc.items() if v[1] > 0}


def main():
    ref = float(sys.argv[2]) if len(sys.argv) > 2 else 160.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    rows = []
    for r in parse(sys.argv[1]):
        flop = 2.0 * r["M"] * r["N"] * r["K"] * r["batch"]
        ideal = flop / (ref * 1e12) * 1e6
        a_el = r["M"] * r["K"] / (9.0 if r["ak"] == 2 else 1.0)      # implicit GEMM: K counts every tap, the gathered tensor is read once
        b_el = r["N"] * r["K"] / (9.0 if r["bk"] % 16 == 3 else 1.0)
        byt = 4.0 * r["batch"] * (r["M"] * r["N"] + a_el + b_el)
        rows.append((r["us"] - ideal, r, byt / 5e6))
    rows.sort(key=lambda t: -t[0])
    print(f"{len(rows)} shapes, one call each: {sum(t[1]['us'] for t in rows) / 1e3:.2f} ms; above {ref:.0f} TFLOP/s: "
          f"{sum(max(t[0], 0) for t in rows) / 1e3:.2f} ms")
    print(f"{'ak':>2} {'bk':>2} {'M':>8} {'N':>6} {'K':>8} {'b':>3} {'tile':>8} {'sk':>3} {'us':>8} {'TFLOP/s':>8} {'over ref us':>11} {'~HBM floor us':>13}")
    for over, r, fl in rows[:top]:
        print(f"{r['ak']:>2} {r['bk']:>2} {r['M']:>8} {r['N']:>6} {r['K']:>8} {r['batch']:>3} {r['tile']:>8} {r['sk']:>3} {r['us']:>8.1f} "
              f"{r['tflops']:>8.1f} {over:>11.1f} {fl:>13.1f}")
    print("idle-device TFLOP/s per kind:", idle_rates(sys.argv[1]))


if __name__ == "__main__":
    main()
