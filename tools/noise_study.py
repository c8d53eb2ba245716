"""Dev tool (GPU box): gradient / activation error of the HIP path vs an fp64 CPU oracle, next to the error of the
fp32 CPU oracle vs the same fp64 truth.  Tells apart 'fp32 round-off amplified by train-mode BN at tiny batch' from
real bugs.  Usage: python tools/noise_study.py [B] [seed]"""
import os; os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill)
import sys
import warnings

import torch

sys.path.insert(0, ".")
from oracle import tris_oracle as O  # noqa: E402
from tris_amd.args import get_parser  # noqa: E402
from tris_amd.CLIP import clip  # noqa: E402
from tris_amd.model.model_stage1 import TRIS  # noqa: E402
from tris_amd.optim import FusedAdamW  # noqa: E402
from tris_amd.train_stage1 import freeze_aux, stage1_forward_losses  # noqa: E402
from tris_amd.utils.synth import seed_fill, synthetic_batch  # noqa: E402



def study(B=2, seed=1234, verbose=True):
    args = get_parser().parse_args(["--size", "320", "--negative_samples", "3"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=20)
    seed_fill(m.state_dict(), seed)
    seed_fill(aux.state_dict(), 4321)
    freeze_aux(aux)
    b = synthetic_batch(B, 320, 20, 3, seed=7)


    def oracle(dt):
        sd = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()).contiguous().clone()
              for k, v in m.state_dict().items()}
        ax = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()).clone()
              for k, v in aux.state_dict().items()}
        bb = dict(b)
        bb["img"] = b["img"].to(dt)
        bbk, newk = O.trainable_split(sd)
        for k in bbk + newk + ["logit_scale"]:
            sd[k].requires_grad_(True)
        out = O.stage1_losses(sd, ax, bb, faithful=False)
        out["loss"].backward()
        return sd, out


    sd64, o64 = oracle(torch.float64)
    sd32, o32 = oracle(torch.float32)
    bbp, newp = m.trainable_parameters()
    FusedAdamW([{"params": bbp}, {"params": newp}], lr=1e-5)
    losses, cls, sig = stage1_forward_losses(m, aux, b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda(), args)
    losses[0].backward()

    named = dict(m.named_parameters())
    rows = []
    for k, p in named.items():
        if p.grad is None or sd64[k].grad is None:
            continue
        g64 = sd64[k].grad
        n64 = float(g64.norm())
        if n64 < 1e-9 * max(1.0, float(sd32[k].grad.abs().max()) * g64.numel() ** 0.5) or n64 == 0.0:
            continue  # analytically-zero gradients (conv biases in front of InstanceNorm)
        gh = p.grad.detach().cpu().double()
        g32 = sd32[k].grad.double()
        s = float(g64.abs().max())
        rows.append(dict(key=k, hip_maxrel=float((gh - g64).abs().max()) / s, f32_maxrel=float((g32 - g64).abs().max()) / s,
                         hip_normrel=abs(float(gh.norm()) - n64) / n64, f32_normrel=abs(float(g32.norm()) - n64) / n64,
                         cos=float((gh * g64).sum() / (gh.norm() * g64.norm()))))
    res = dict(hip=losses.tolist(), f32=[float(o32[k]) for k in ("loss", "l1", "l4", "l5")],
               f64=[float(o64[k]) for k in ("loss", "l1", "l4", "l5")], rows=rows)
    for name, t, k in (("cls", cls, "cls"), ("sig", sig, "sig")):
        res[name + "_hip"] = float((t.detach().cpu().double() - o64[k]).abs().max())
        res[name + "_f32"] = float((o32[k].double() - o64[k]).abs().max())
    return res


if __name__ == "__main__":
    import statistics
    r = study(int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
    print("losses hip", r["hip"])
    print("losses f32", r["f32"])
    print("losses f64", r["f64"])
    print("cls err hip/f32", r["cls_hip"], r["cls_f32"], " sig err hip/f32", r["sig_hip"], r["sig_f32"])
    rows = sorted(r["rows"], key=lambda x: -x["hip_normrel"])
    print("worst 20 by hip norm-rel:  hip_maxrel  f32_maxrel  hip_normrel  f32_normrel  cos  key")
    for x in rows[:20]:
        print(f"  {x['hip_maxrel']:.2e}  {x['f32_maxrel']:.2e}  {x['hip_normrel']:.2e}  {x['f32_normrel']:.2e}  {x['cos']:.6f}  {x['key']}")
    print("median hip/f32 maxrel ratio:", statistics.median([x["hip_maxrel"] / (x["f32_maxrel"] + 1e-12) for x in rows]))
    print("median hip/f32 normrel ratio:", statistics.median([x["hip_normrel"] / (x["f32_normrel"] + 1e-12) for x in rows]))
    print("lowest cos:")
    for x in sorted(rows, key=lambda x: x["cos"])[:12]:
        print(f"  {x['hip_maxrel']:.2e}  {x['f32_maxrel']:.2e}  {x['hip_normrel']:.2e}  {x['f32_normrel']:.2e}  {x['cos']:.6f}  {x['key']}")
