"""Flash-style attention, f32-MFMA kernels (csrc/attn_mfma.hip) against the h2 kernels on the 16-bit MFMA (csrc/attn_h2.hip):
forward and backward at the shapes of the step (GPU box):  python tools/mha_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
from tris_amd.config import cfg
ops.set_gemm_mode("h2")
for (N, L, heads, causal, what) in ((48, 401, 12, False, "ViT-B/16 trunk, 320 px"), (192, 50, 12, False, "aux ViT-B/32"),
                                    (3840, 20, 8, True, "text encoder (flash form forced)")):
    W = heads * 64
    g = torch.Generator().manual_seed(1)
    qkv0 = torch.randn(N, L, 3 * W, generator=g).cuda()
    wt = torch.randn(N, L, W, generator=g).cuda()
    fl = 4.0 * N * heads * L * L * 64 * (0.5 if causal else 1.0)
    res = {}

    def tag(t):   # what the producing product does in the step: leave the tensor's amax word behind
        slot = ops._h2_slot()
        ops.call("tris_amax_bits_f32", ops.P(t), t.numel(), slot, ops._stream())
        t._h2 = (ops._H2["step"], slot, t._version)
    for form in ("f32", "h2"):
        cfg.mha_h2, cfg.mha = form == "h2", "mfma"
        qkv = qkv0.clone().requires_grad_(True)
        tag(qkv); tag(wt)
        ts_f, ts_b = [], []
        for it in range(8):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            qkv.grad = None
            a.record()
            o = ops.mha(qkv, heads, causal)
            b.record()
            o.backward(wt)
            c.record()
            torch.cuda.synchronize()
            ts_f.append(a.elapsed_time(b)); ts_b.append(b.elapsed_time(c))
        f, bw = sorted(ts_f)[3], sorted(ts_b)[3]
        res[form] = (o.detach().clone(), qkv.grad.clone())
        print(f"{what:34s} N {N:5d} L {L:4d} {form:4s} ({ops.MHA_STATS['last']})  fwd {f * 1e3:8.1f} us {fl / f / 1e9:7.1f} TF/s   "
              f"bwd {bw * 1e3:8.1f} us {2.5 * fl / bw / 1e9:7.1f} TF/s")
    eo = float((res["h2"][0] - res["f32"][0]).abs().max() / res["f32"][0].abs().max())
    eg = float((res["h2"][1] - res["f32"][1]).abs().max() / res["f32"][1].abs().max())
    print(f"{'':34s} h2 vs f32: out {eo:.2e}  dqkv {eg:.2e} of the largest element")
