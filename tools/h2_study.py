"""Study for a cheaper exact-class product (DESIGN.md section 6, round-3 addendum): what would a TWO-piece fp16 split (three f16 MFMAs
per product instead of six bf16 ones) do to the accuracy of the step's dense products, given fp16's exponent range?

One training step runs on the GPU with `ops.gemm` wrapped: for a sample of the products it launches (data gradients, weight
gradients, forward Linear layers -- whatever goes through ops.gemm with TRIS_BN_BWD_FUSE=0) the operands are taken as they are
and the product is re-evaluated with torch in several arithmetics, each against an fp64 reference:
  f32      plain fp32 product                                  x3   three bf16 pieces, six products (the product path)
  x2       two bf16 pieces, three products                     h2   two fp16 pieces, three products, no scaling
  h2t      h2 with one power-of-two scale per tensor           h2r  h2 with one power-of-two scale per operand row (A rows, B cols)
  h2r_ftz  h2r with fp16 subnormals flushed to zero
Error = ||C - C64||_F / ||C64||_F (and the worst row: max_i ||C_i - C64_i|| / ||C64_i||).  Developer tool: uses torch.matmul."""
import os, sys, warnings
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); os.environ["TRIS_BN_BWD_FUSE"] = "0"; os.environ["TRIS_STEP_GRAPH"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tris_amd import ops
from tris_amd.args import get_parser
from tris_amd.CLIP import clip
from tris_amd.model.model_stage1 import TRIS
from tris_amd.optim import FusedAdamW
from tris_amd.train_stage1 import freeze_aux, train_step
from tris_amd.utils.synth import seed_fill, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 9
F16_MIN_NORMAL = 2.0 ** -14


def pieces_bf16(a, n):
    out, r = [], a
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def pieces_f16(a, ftz=False):
    hi = a.half().float()
    lo = (a - hi).half().float()
    if ftz:
        hi = torch.where(hi.abs() < F16_MIN_NORMAL, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < F16_MIN_NORMAL, torch.zeros_like(lo), lo)
    return hi, lo


def pow2(x):   # power of two that brings x to [2^13, 2^14): products of two such operands stay far below fp16's 65504
    e = torch.floor(torch.log2(x.clamp_min(1e-30)))   # (all-zero rows: any finite scale)
    return torch.exp2(13.0 - e)


def study(tag, A, Bm):
    """A [M,K], Bm [K,N] fp32"""
    ref = A.double() @ Bm.double()
    nr = ref.norm()
    rown = ref.norm(dim=1).clamp_min(1e-300)

    def err(C):
        d = C.double() - ref
        return float(d.norm() / nr), float((d.norm(dim=1) / rown).max())
    res = {"f32": err(A @ Bm)}
    a3, b3 = pieces_bf16(A, 3), pieces_bf16(Bm, 3)
    res["x3"] = err(a3[2] @ b3[0] + a3[0] @ b3[2] + a3[1] @ b3[1] + a3[1] @ b3[0] + a3[0] @ b3[1] + a3[0] @ b3[0])
    res["x2"] = err(a3[1] @ b3[0] + a3[0] @ b3[1] + a3[0] @ b3[0])

    def h2(sa, sb, ftz=False):
        ah, al = pieces_f16(A * sa, ftz)
        bh, bl = pieces_f16(Bm * sb, ftz)
        return (al @ bh + ah @ bl + ah @ bh) / (sa * sb)
    one = torch.ones((), device=A.device)
    res["h2"] = err(h2(one, one))
    res["h2t"] = err(h2(pow2(A.abs().max()), pow2(Bm.abs().max())))
    sa, sb = pow2(A.abs().amax(dim=1, keepdim=True)), pow2(Bm.abs().amax(dim=0, keepdim=True))
    res["h2r"] = err(h2(sa, sb))
    res["h2r_ftz"] = err(h2(sa, sb, True))
    tiny = float((A.abs() < F16_MIN_NORMAL).float().mean()), float((Bm.abs() < F16_MIN_NORMAL).float().mean())
    print(f"{tag:34s} amax {float(A.abs().max()):8.2e} {float(Bm.abs().max()):8.2e}  <2^-14: {tiny[0]:5.3f} {tiny[1]:5.3f} | " +
          " ".join(f"{k} {v[0]:.1e}/{v[1]:.1e}" for k, v in res.items()), flush=True)
    return res


def main():
    args = get_parser().parse_args(["--size", "320", "--negative_samples", "3", "--max_query_len", "20"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
    seed_fill(model.state_dict(), 1234); seed_fill(aux.state_dict(), 4321); freeze_aux(aux)
    bb, new = model.trainable_parameters()
    opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr, weight_decay=args.weight_decay)
    b = synthetic_batch(B, 320, 20, 3, seed=7)
    bt = (b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda())
    os.environ["TRIS_TEXT_STREAM"] = os.environ["TRIS_WGRAD_STREAM"] = "0"   # one stream: the wrapper reads operands in place
    for _ in range(2):
        train_step(model, aux, opt, *bt, args, None)
    torch.cuda.synchronize()
    real, count, seen, rows = ops.gemm, [0], set(), []

    def wrapped(A, Bt, C, M, N, K, lda, ldb, ldc, tA, tB, batch=1, **kw):
        count[0] += 1
        key = (M, N, K, tA, tB)
        ok = batch == 1 and M * N <= (1 << 27) and lda == (M if tA else K) and ldb == (K if tB else N) and min(M, N, K) >= 16
        if ok and key not in seen and count[0] % EVERY == 0:
            seen.add(key)
            with torch.no_grad():
                Am = A.reshape(-1)[:M * K].view(K, M).t() if tA else A.reshape(-1)[:M * K].view(M, K)
                Bm = Bt.reshape(-1)[:N * K].view(N, K).t() if tB else Bt.reshape(-1)[:N * K].view(K, N)
                rows.append(study(f"{'T' if tA else 'N'}{'T' if tB else 'N'} M{M} N{N} K{K}", Am.contiguous(), Bm.contiguous()))
        return real(A, Bt, C, M, N, K, lda, ldb, ldc, tA, tB, batch=batch, **kw)
    ops.gemm = wrapped
    print(f"# one training step at B = {B}; every {EVERY}th ops.gemm launch with a new shape; error = relative Frobenius / worst row")
    train_step(model, aux, opt, *bt, args, None)
    torch.cuda.synchronize()
    ops.gemm = real
    print(f"# {len(rows)} products studied of {count[0]} ops.gemm launches; worst over them (Frobenius / worst row):")
    for k in rows[0]:
        print(f"#   {k:8s} {max(r[k][0] for r in rows):.2e} / {max(r[k][1] for r in rows):.2e}")


main()
