"""TEST INFRASTRUCTURE ONLY -- the fp32 round-off floor of the G6 gradient probes (tests/golden/g6_noise_floor.npz).

    python oracle/gen_noise_floor.py

Runs oracle/tris_oracle.py on the G5/G6 inputs (seed-filled weights 1234 / 4321, synthetic batch of 2, seed 7) twice -- in fp32
and in fp64 -- and records, for every probed parameter of tests/golden/g5_g6_step.npz: the fp64 gradient norm and first 16
gradient values (the truth up to fp64 round-off) and how far the fp32 run is from them.  tests/test_gpu_parity.py::
test_g5_g6_train_step bounds the HIP path's deviation from the fp64 values by a small multiple of the fp32 run's own deviation:
a bound calibrated on the problem's conditioning (about 50 train-mode BatchNorms at batch 2) instead of a flat percentage.
No reference code is involved: the oracle is pinned against the reference by oracle/gen_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
os.environ.setdefault("TRIS_RANDOM_INIT", "1")

from oracle import tris_oracle as O  # noqa: E402
from tris_amd.utils.shapes import aux_state_dict_spec, empty_state_dict, tris_state_dict_spec  # noqa: E402
from tris_amd.utils.synth import seed_fill, synthetic_batch  # noqa: E402


def run(dt):
    sd = seed_fill(empty_state_dict(tris_state_dict_spec()), 1234)
    aux = seed_fill(empty_state_dict(aux_state_dict_spec()), 4321)
    sd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
    aux = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in aux.items()}
    b = dict(synthetic_batch(2, 320, 20, 3, seed=7))
    b["img"] = b["img"].to(dt)
    bb, new = O.trainable_split(sd)
    for k in bb + new + ["logit_scale"]:
        sd[k].requires_grad_(True)
    out = O.stage1_losses(sd, aux, b, faithful=False)
    out["loss"].backward()
    return sd


def main():
    torch.set_num_threads(8)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g5_g6_step.npz"), allow_pickle=False)
    probes = [n[len("grad_norm."):] for n in g.files if n.startswith("grad_norm.")]
    s64, s32 = run(torch.float64), run(torch.float32)
    rec = {}
    for k in probes:
        a, c = s64[k].grad.double().reshape(-1), s32[k].grad.double().reshape(-1)
        n64 = float(a.norm())
        rec["norm64." + k] = np.float64(n64)
        rec["head64." + k] = a[:16].numpy().copy()
        rec["f32_normdev." + k] = np.float64(abs(float(c.norm()) - n64))
        rec["f32_headdev." + k] = np.float64(float((c[:16] - a[:16]).abs().max()))
        rec["f32_maxdev." + k] = np.float64(float((c - a).abs().max()))
        rec["absmax64." + k] = np.float64(float(a.abs().max()))
        print(f"{k:62s} |g| {n64:.4e}  f32 norm dev {rec['f32_normdev.' + k] / max(n64, 1e-300):.2e}  head dev / max {rec['f32_headdev.' + k] / max(rec['absmax64.' + k], 1e-300):.2e}"
              f"  vs golden norm {abs(float(g['grad_norm.' + k]) - n64) / max(n64, 1e-300):.2e}")
    np.savez(os.path.join(ROOT, "tests", "golden", "g6_noise_floor.npz"), **rec)


if __name__ == "__main__":
    main()
