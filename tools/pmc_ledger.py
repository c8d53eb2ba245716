"""Per-(kind, shape) HBM traffic ledger of the GEMM / conv family: PMC bytes next to algorithmic bytes.

Joins   (a) the launches of ONE instrumented step in issue order -- tools/step_ledger.py with LEDGER_ORDER=<file>, run under rocprofv3 --
with    (b) the per-dispatch FETCH_SIZE / WRITE_SIZE of the same two runs (rocprofv3 --kernel-trace --pmc FETCH_SIZE, and again
            with WRITE_SIZE: separate passes; FETCH_SIZE doubled for gfx950, MI355X_MICROARCH.md HBM section; units KB).
Every ledger record launches exactly one MAIN kernel (gemm_fast / gemm / wgrad3x3_direct / stem_wgrad / mha / xattn_px) possibly
followed by helper launches (split-K / slab reduces, the cross-attention preparation): the last R main dispatches of the process are
the R records of the profiled step (it is the last step the tool runs).  The join is checked (counts must agree) and refused otherwise.
usage: pmc_ledger.py order.txt fetch.csv write.csv [out.txt]"""
import collections
import csv
import re
import sys

MAIN = re.compile(r"gemm_fast_kernel|gemm_kernel<|wgrad3x3_direct_kernel|stem_wgrad_kernel|stem_conv1_kernel|mha_mfma_fwd|mha_h2_fwd|xattn_px_kernel|xattn_px_bwd_kernel|"
                  r"xattn_fused|xattn_pair")
HELPER = re.compile(r"splitk_reduce|slab_reduce|stem_wgrad_reduce|xattn_text_planes|xattn_bwd_prep|xattn_prep")


def dispatches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    return [(r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0) for r in rows]


def attach(disp):
    """-> list of [main name, bytes incl. helpers] in dispatch order"""
    out = []
    for name, v in disp:
        if MAIN.search(name):
            out.append([name, v])
        elif HELPER.search(name) and out:
            out[-1][1] += v
    return out


def main():
    order = [ln.rstrip("\n").split("\t") for ln in open(sys.argv[1])]
    R = len(order)
    f = attach(dispatches(sys.argv[2], "FETCH_SIZE"))
    w = attach(dispatches(sys.argv[3], "WRITE_SIZE"))
    out = open(sys.argv[4], "w") if len(sys.argv) > 4 else None
    if len(f) < R or len(w) < R:
        sys.exit(f"join refused: {R} ledger records, {len(f)} / {len(w)} main dispatches in the PMC passes")
    f, w = f[-R:], w[-R:]
    agg = collections.OrderedDict()
    klass = {"gemm": "gemm_fast_kernel|gemm_kernel<", "conv3x3_fwd": "gemm_fast_kernel|gemm_kernel<|stem_conv1", "conv3x3_dgrad": "gemm_fast_kernel|gemm_kernel<",
             "conv3x3_wgrad": "gemm_fast_kernel|gemm_kernel<|wgrad3x3_direct|stem_wgrad", "mha_fwd": "mha_", "xattn_fwd": "xattn", "xattn_bwd": "xattn_px_bwd"}
    for (k, fl, nb, ms), (fn, fb), (wn, wb) in zip(order, f, w):
        want = next(v for kk, v in klass.items() if k.split(":")[0].startswith(kk))
        if not re.search(want, fn):
            sys.exit(f"join refused: record {k} landed on kernel {fn[:70]}")
        if fn != wn:
            sys.exit(f"join refused: the two PMC passes disagree on the kernel of a record ({fn[:60]} vs {wn[:60]})")
        e = agg.setdefault(k, [0, 0.0, 0.0, 0.0, fn])
        e[0] += 1
        e[1] += float(nb)
        e[2] += 2.0 * fb       # gfx950: FETCH_SIZE tallies 128-byte requests at 64 B
        e[3] += wb
    rows = sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3] - kv[1][1]))
    tot_a = sum(v[1] for v in agg.values())
    tot_p = sum(v[2] + v[3] for v in agg.values())
    lines = [f"# tools/pmc_ledger.py: {R} launches of one step; algorithmic {tot_a / 1e9:.2f} GB, PMC fetch(x2) + write {tot_p / 1e9:.2f} GB "
             f"= {tot_p / max(tot_a, 1):.2f} x; excess {max(tot_p - tot_a, 0) / 1e9:.2f} GB",
             f"# {'n':>3} {'alg MB':>9} {'fetch MB':>9} {'write MB':>9} {'ratio':>6} {'excess MB':>10}  kind:shape   [kernel]"]
    for k, (n, a, fb, wb, kn) in rows:
        kn = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", kn)[:60]
        lines.append(f"  {n:3d} {a / 1e6:9.1f} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {(fb + wb) / max(a, 1):6.2f} {(fb + wb - a) / 1e6:10.1f}  {k}   [{kn}]")
    for ln in lines:
        print(ln)
        if out:
            out.write(ln + "\n")


if __name__ == "__main__":
    main()
