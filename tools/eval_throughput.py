"""Evaluation throughput of BASELINE configs[1] (validate.py's loop): refs per second of the one-ref-at-a-time loop (trunk and
sentence halves replayed from hipGraphs) against the batched evaluation (cfg.eval_group refs per pass), on a synthetic loader of
RefCOCOg-shaped refs (320 px images, 2 sentences each, 427 x 640 masks) -- same (oIoU, mIoU, hit) from both, asserted here.
Used by bench.py (`eval` object) and runnable on its own:  python tools/eval_throughput.py"""
import json
import os
import sys
import time
import warnings
from types import SimpleNamespace

os.environ.setdefault("TRIS_RANDOM_INIT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def measure(n_refs=64, sentences=2, groups=(1, 16, 32), model=None):
    from tris_amd.args import get_parser
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.utils.synth import seed_fill, synthetic_batch, synthetic_ids
    from tris_amd.validate import validate
    if model is None:
        a = get_parser().parse_args(["--size", "320", "--max_query_len", "20"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = TRIS(a).cuda()
        seed_fill(model.state_dict(), 1234)
    was_training = model.training
    model.eval()
    args = SimpleNamespace(print_freq=10 ** 9, cam_save_dir=None, name_save_dir=None, dataset="refcocog", save_cam=False,
                           max_query_len=20)
    rng = np.random.RandomState(5)
    loader = []
    for r in range(n_refs):
        img = synthetic_batch(1, 320, 20, 0, seed=500 + r)["img"].cuda()
        ids = torch.from_numpy(synthetic_ids(sentences, 20, rng)).cuda()
        tgt = torch.zeros(1, 427, 640, dtype=torch.int64, device="cuda")
        y0, x0 = int(rng.randint(0, 200)), int(rng.randint(0, 300))
        tgt[0, y0:y0 + 180, x0:x0 + 260] = 1
        loader.append(({"img": img, "word_ids": ids.t().reshape(1, 1, 20, sentences)},
                       {"target": tgt, "boxes": torch.tensor([[x0, y0, x0 + 260, y0 + 180]]), "img_path": torch.tensor([r])}))
    out, first = {}, None
    quiet = SimpleNamespace(info=lambda *a, **k: None)
    for g in groups:
        from tris_amd.config import cfg
        with cfg.override(eval_group=int(g)):
            validate(args, loader[:max(g, 4)], model, 0, logger=quiet)          # graphs / autotune / allocator warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = validate(args, loader, model, 0, logger=quiet)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res = (res[0], float(res[1]), res[2])
        first = res if first is None else first
        assert res == first, ("batched evaluation changed the metrics", g, res, first)
        out[f"group{g}"] = {"refs_per_s": round(n_refs / dt, 1), "ms_per_image_sentence": round(dt / (n_refs * sentences) * 1e3, 3)}
    if was_training:
        model.train()
    base = out[f"group{groups[0]}"]["refs_per_s"]
    best = max(v["refs_per_s"] for v in out.values())
    return {"refs": n_refs, "sentences_per_ref": sentences, "metrics_identical_across_groups": True,
            "oIoU_mIoU_hit": [round(float(v), 6) for v in first], "eval_refs_per_s": best, "speedup_vs_one_at_a_time": round(best / base, 2),
            "by_group": out}


if __name__ == "__main__":
    print(json.dumps(measure()))
