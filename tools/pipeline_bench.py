"""Input-pipeline measurement: HBM cache build rate, per-batch assembly time, the gather+normalise
kernel against the HBM roofline, and the reference-style CPU path (ReferDataset.__getitem__ with PIL transforms) beside
it.  Used by bench.py (`input_pipeline` object) and runnable on its own:  python tools/pipeline_bench.py"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(batch=48, size=320, n_images=48, iters=20):
    from tris_amd import ops
    from tris_amd.dataset.hbm import HbmLoader, HbmReferCache
    from tris_amd.dataset.ReferDataset import ReferDataset
    from tris_amd.dataset.transform import get_transform
    from tris_amd.utils.synth import make_mini_refer, word_hash_tokenize
    root = make_mini_refer(tempfile.mkdtemp(prefix="tris_refer_"), n_images=n_images, seed=0,
                           sizes=[(480, 640), (640, 480), (375, 500), (427, 640)])
    ds = ReferDataset(root, "refcocog", "umd", image_transforms=get_transform(size, True), split="train", eval_mode=False,
                      size=size, max_tokens=20, negative_samples=3, tokenizer=word_hash_tokenize)
    # reference-style CPU path: one process, PIL decode + resize + normalise per sample
    np.random.seed(0)
    t0 = time.perf_counter()
    n_cpu = min(len(ds), 64)
    for i in range(n_cpu):
        ds[i]
    cpu_rate = n_cpu / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    cache = HbmReferCache(ds, size)
    build_s = time.perf_counter() - t0
    loader = HbmLoader(cache, batch_size=batch)
    idx = list(np.random.RandomState(0).randint(0, len(ds), batch))
    for _ in range(3):
        loader.train_batch(idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loader.train_batch(idx)
    torch.cuda.synchronize()
    batch_ms = (time.perf_counter() - t0) / iters * 1e3
    # the dominant kernel alone, HIP events on the launch stream
    slot = torch.from_numpy(cache.slot_of[idx]).cuda()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.gather_normalize(cache.images, slot, cache.lut)
    e0.record()
    for _ in range(iters):
        ops.gather_normalize(cache.images, slot, cache.lut)
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / iters
    bytes_alg = batch * size * size * (3 + 12)          # uint8 pixels in, fp32 channels-last out
    return {
        "workload": f"batch {batch} x {size}px from an HBM cache of {cache.images.shape[0]} images / {len(ds)} refs",
        "batch_assembly_ms": round(batch_ms, 3), "images_per_s": round(batch / batch_ms * 1e3, 1),
        "cache_build_images_per_s": round(cache.images.shape[0] / build_s, 1),
        "roofline": {"kernel": "u8_gather_normalize_kernel", "bound": "hbm", "achieved": round(bytes_alg / k_ms / 1e6, 1),
                     "peak": 8000.0, "unit": "GB/s", "frac": round(bytes_alg / k_ms / 1e6 / 8000.0, 4), "traffic": None},
        "cpu_baseline": {"value": round(cpu_rate, 1), "unit": "img/s", "cores": 1, "kind": "port",
                         "sample": f"{n_cpu} x ReferDataset.__getitem__ (PIL decode + Pillow resize + normalise), one process"},
    }


if __name__ == "__main__":
    print(json.dumps(measure()))
