"""Per-shape ledger of the GEMM / conv family inside one Stage-1 step at the headline batch.

Every product launch of one instrumented step (kernels serialised: side streams and graph replay off, as in bench.py's roofline
pass) is bracketed by HIP events and logged with its shape, its algorithmic FLOPs and its algorithmic HBM bytes (every operand
read once, every output written once).  Rows are grouped by (kind, shape) and ranked by the time they spend ABOVE their own floor
    floor = max(bytes / 6.3 TB/s (achievable HBM), FLOPs / 600 TFLOP/s (the h2 core's MFMA-only rate under load))
so the top of the list is where kernel time can actually be taken back, and the column says from which side.
usage: python tools/step_ledger.py [batch=48] [out.txt]
LEDGER_ORDER=<file>: also write the launches in issue order (kind:shape, FLOPs, algorithmic bytes, ms) -- run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and again with WRITE_SIZE) and join with tools/pmc_ledger.py."""
import os
import sys
import warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tris_amd import ops
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.config import cfg
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch

HBM_ACH, MFMA_ACH = 6.3e12, 600e12


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else None
    if cfg.own_stream:
        torch.cuda.set_stream(ops.compute_stream())
    if not os.environ.get("TRIS_GEMM_MODE"):
        ops.set_gemm_mode("h2")
    args = get_parser().parse_args(["--backbone", os.environ.get("LEDGER_BACKBONE", "clip-RN50"), "--size", "320", "--max_query_len", "20",
                                    "--negative_samples", "3", "--batch_size", str(B), "--epoch", "15"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
    seed_fill(model.state_dict(), 1234)
    seed_fill(aux.state_dict(), 4321)
    freeze_aux(aux)
    bb, new = model.trainable_parameters()
    opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: (1 - x / 15000) ** 0.9)
    b = synthetic_batch(B, 320, 20, 3, seed=7)
    img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()

    def step():
        return train_step(model, aux, opt, img, ids, neg, args, sched, None)
    ops._PROF_SHAPES = True
    with cfg.override(step_graph="0", text_stream=False, wgrad_stream=False):
        step()
        step()
        ops.profile_begin()
        step()
        rec = ops.profile_end()
    if os.environ.get("LEDGER_ORDER"):      # the launches in issue order: tools/pmc_ledger.py joins them with a rocprofv3 --pmc pass of this run
        with open(os.environ["LEDGER_ORDER"], "w") as fh:
            for k, fl, ms, nb in rec:
                fh.write(f"{k}\t{fl:.0f}\t{nb:.0f}\t{ms:.6f}\n")
    rows = {}
    for k, fl, ms, nb in rec:
        e = rows.setdefault(k, [0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += fl
        e[2] += ms
        e[3] += nb
    tot = sum(v[2] for v in rows.values())
    lines = [f"# tools/step_ledger.py B={B} mode={ops.get_gemm_mode()} planes={cfg.h2_planes}: {len(rec)} launches, {tot:.3f} ms serialised, "
             f"{sum(v[1] for v in rows.values()) / 1e9:.1f} GFLOP, {sum(v[3] for v in rows.values()) / 1e9:.2f} GB algorithmic",
             f"# {'ms':>8} {'%':>5} {'n':>4} {'us/call':>8} {'TF/s':>7} {'GB/s':>7} {'floor us':>8} {'side':>4} {'over ms':>8}  kind:shape"]
    table = []
    for k, (n, fl, ms, nb) in rows.items():
        f_h, f_m = nb / HBM_ACH * 1e3, fl / MFMA_ACH * 1e3
        floor = max(f_h, f_m)
        table.append((ms - floor, k, n, fl, ms, nb, floor, "hbm" if f_h >= f_m else "mfma"))
    table.sort(key=lambda t: -t[0])
    over_tot = sum(max(t[0], 0) for t in table)
    for over, k, n, fl, ms, nb, floor, side in table:
        lines.append(f"  {ms:8.3f} {100 * ms / tot:5.1f} {n:4d} {1e3 * ms / n:8.1f} {fl / (ms * 1e-3) / 1e12:7.1f} {nb / (ms * 1e-3) / 1e9:7.0f} "
                     f"{1e3 * floor / n:8.1f} {side:>4} {over:8.3f}  {k}")
    lines.append(f"# time above the floors: {over_tot:.3f} ms of {tot:.3f}")
    for ln in lines:
        print(ln, flush=True)
        if out:
            out.write(ln + "\n")


if __name__ == "__main__":
    main()
