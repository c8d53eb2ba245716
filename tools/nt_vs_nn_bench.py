"""Data gradient of a Linear layer dX = dY . W as the NN product it is (B = W [N_out, K_in] row-major: k-major rows, transposed while
staged) against the NT product on a transposed copy of the weight (B^T = W^T rows contiguous along the reduction): the fast kernel's two
B kinds at the transformer shapes of the step (GPU box):  python tools/nt_vs_nn_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
ops.set_gemm_mode("h2")
ops.set_autotune(True)


def tag(t):
    slot = ops._h2_slot()
    ops.call("tris_amax_bits_f32", ops.P(t), t.numel(), slot, ops._stream())
    t._h2 = (ops._H2["step"], slot, t._version)


for (M, Nout, Kin, what) in ((19248, 3072, 768, "ViT-B/16 c_fc dgrad"), (19248, 768, 3072, "ViT-B/16 c_proj dgrad"),
                             (19248, 2304, 768, "ViT-B/16 in_proj dgrad"), (19248, 768, 768, "ViT-B/16 out_proj dgrad"),
                             (3840, 2048, 512, "text c_fc dgrad"), (3840, 512, 2048, "text c_proj dgrad"), (3840, 1536, 512, "text in_proj dgrad")):
    g = torch.Generator().manual_seed(M + Nout)
    dY = torch.randn(M, Nout, generator=g).cuda()
    W = (torch.randn(Nout, Kin, generator=g) * 0.05).cuda()
    Wt = W.t().contiguous()
    tag(dY); tag(W); Wt._h2 = W._h2
    dX = torch.empty(M, Kin, device="cuda")
    res = {}
    for form in ("NN", "NT"):
        def run():
            if form == "NN":
                ops.gemm(dY, W, dX, M, Kin, Nout, Nout, Kin, Kin, False, False)
            else:
                ops.gemm(dY, Wt, dX, M, Kin, Nout, Nout, Nout, Kin, False, True)
        for _ in range(3):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            run()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        res[form] = (us, dX.clone())
        print(f"{what:26s} M {M:6d} N {Kin:5d} K {Nout:5d}  {form}  {us:8.1f} us  {2.0 * M * Kin * Nout / us / 1e6:7.1f} TF/s")
    print(f"{'':26s} NT / NN time {res['NT'][0] / res['NN'][0]:.2f}   max |diff| / max {float((res['NT'][1] - res['NN'][1]).abs().max() / res['NN'][1].abs().max()):.1e}")
