"""HBM-resident input pipeline for Stage-1 (replaces DataLoader(ReferDataset, num_workers=2) +
dataset/transform.py for the training / evaluation loops).

At >800 img/s per GPU two CPU workers decoding and resizing JPEGs cannot feed the step, and an MI355X has 288 GB of
HBM: the whole dataset, decoded and resized ONCE to the network resolution, is a few GB of uint8 (RefCOCOg-umd train:
~21.9k images x 320*320*3 B = 6.7 GB).  So:

  build  (once):  PIL decode on the host -> uint8 to the GPU -> Pillow-exact antialiased resize (tris_resample_u8)
                  into images[n_img, S, S, 3];  masks -> Pillow-exact NEAREST (tris_gather2d_u8) into targets[n_ref, S, S]
                  (training) or kept at their original sizes in one flat buffer (evaluation);  token rows of every
                  sentence into tokens[n_sent, L].
  batch  (per step):  the host draws the sentence / negatives of each sample with the reference's own
                  np.random.choice sequence (ReferDataset.sample_text), ships ONE small int64 index vector, and three
                  gather kernels assemble the batch: images are normalised through a 3x256 table straight into the
                  channels-last fp32 layout the stem convolution reads (no NCHW->NHWC pass).

Batches carry the reference's keys, shapes and dtypes (what default_collate makes of ReferDataset samples), so
train_one_epoch / validate take either loader.
"""
import os

import numpy as np
import torch

from .. import ops
from .transform import normalize_table


class HbmReferCache:
    def __init__(self, dataset, size, device="cuda"):
        if not torch.cuda.is_available():
            raise ops.NoGpuError("HbmReferCache needs the GPU (the CPU path is DataLoader(ReferDataset))")
        self.dataset, self.size, self.device = dataset, size, torch.device(device)
        ds, S, dev = dataset, size, self.device
        n = len(ds)
        slot = {}
        self.slot_of = np.empty(n, np.int64)
        recs = [ds.image_record(i) for i in range(n)]
        for i, (_, img_id, _) in enumerate(recs):
            self.slot_of[i] = slot.setdefault(img_id, len(slot))
        self.images = torch.empty(len(slot), S, S, 3, dtype=torch.uint8, device=dev)
        self.lut = normalize_table().to(dev)
        train = not ds.eval_mode
        self.targets = torch.empty(n, S, S, dtype=torch.uint8, device=dev) if train else None
        self.eval_targets = []            # evaluation: masks at their original sizes
        boxes, paths, sizes, self.files = [], [], [], []
        # decode (once per image) and mask rasterisation (once per ref) on a few host threads -- Pillow releases the
        # GIL while decoding --, uploads and the resize kernels in order on this thread
        from concurrent.futures import ThreadPoolExecutor
        first_ref = {}
        for i in range(n):
            first_ref.setdefault(int(self.slot_of[i]), i)
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            for s, img in zip(first_ref, pool.map(ds.load_image, first_ref.values())):
                ops.resample_u8(torch.from_numpy(np.asarray(img).copy()).to(dev), S, S, out=self.images[s])
            for i, (annot, bbox) in enumerate(pool.map(ds.load_mask, range(n))):
                m = torch.from_numpy(np.asarray(annot).copy()).to(dev)
                if train:
                    ops.resize_nearest_u8(m, S, S, out=self.targets[i])
                else:
                    self.eval_targets.append(m)
                w, h = annot.size
                fname = recs[i][2]["file_name"]
                boxes.append(bbox)
                sizes.append((h, w))
                paths.append(int(fname.split(".")[0].split("_")[-1]))
                self.files.append(fname)
        self.boxes = torch.from_numpy(np.stack(boxes).astype(np.int64)).to(dev)
        self.orig_size = torch.from_numpy(np.array(sizes, np.int64)).to(dev)
        self.img_path = torch.tensor(paths, dtype=torch.int64, device=dev).view(-1, 1)
        self._index_tokens(n)
        torch.cuda.synchronize(dev)

    def _index_tokens(self, n):
        # token rows of every sentence, ref-major
        ds, dev = self.dataset, self.device
        self.sent_base = np.zeros(n + 1, np.int64)
        for i in range(n):
            self.sent_base[i + 1] = self.sent_base[i] + len(ds.input_ids[i])
        rows = torch.cat([r for ref in ds.input_ids for r in ref], dim=0)
        mrows = torch.cat([r for ref in ds.word_masks for r in ref], dim=0)
        assert rows.shape[1] * rows.element_size() % 4 == 0
        self.tokens = rows.contiguous().to(dev)
        self.token_masks = mrows.contiguous().to(dev)

    def nbytes(self):
        t = self.images.numel() + self.tokens.numel() * self.tokens.element_size()
        if self.targets is not None:
            t += self.targets.numel()
        return t + sum(m.numel() for m in self.eval_targets)


class HbmLoader:
    """Iterates (samples, targets) batches assembled on the GPU from an HbmReferCache.

    sampler: any iterable of dataset indices (e.g. torch DistributedSampler); otherwise sequential or shuffled with
    torch's global RNG like DataLoader(shuffle=True).  Evaluation datasets yield one ref per batch (all its sentences
    stacked on the last axis, the mask at its original size), as the reference's batch_size=1 loaders do."""

    def __init__(self, cache, batch_size=1, sampler=None, shuffle=False, drop_last=False, planar=False):
        self.cache, self.batch_size, self.sampler = cache, batch_size, sampler
        self.shuffle, self.drop_last, self.planar = shuffle, drop_last, planar
        self.dataset = cache.dataset
        if self.dataset.eval_mode and batch_size != 1:
            raise ValueError("evaluation refs have ragged sentence counts and mask sizes: batch_size must be 1")

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        if self.sampler is not None:
            return [int(i) for i in self.sampler]
        n = len(self.dataset)
        return torch.randperm(n).tolist() if self.shuffle else list(range(n))

    def __iter__(self):
        order = self._order()
        bs = self.batch_size
        for a in range(0, len(order), bs):
            chunk = order[a:a + bs]
            if len(chunk) < bs and self.drop_last:
                break
            yield self.eval_batch(chunk[0]) if self.dataset.eval_mode else self.train_batch(chunk)

    # ---- training: [B] refs, one drawn sentence + negatives each ---------------------------------------------------
    def train_batch(self, indices):
        c, ds = self.cache, self.dataset
        B, K = len(indices), ds.negative_samples
        picks = [ds.sample_text(i) for i in indices]          # host RNG, reference call order
        idx = np.empty(B * (3 + K), np.int64)
        idx[0:B] = indices
        idx[B:2 * B] = c.slot_of[indices]
        idx[2 * B:3 * B] = [c.sent_base[i] + p["choice"] for i, p in zip(indices, picks)]
        idx[3 * B:] = [c.sent_base[j] + s for p in picks for j, s in p["neg"]]
        dev = torch.from_numpy(idx).to(c.device, non_blocking=True)
        ref, slot, pos, neg = dev[0:B], dev[B:2 * B], dev[2 * B:3 * B], dev[3 * B:]
        S = c.size
        samples = {
            "img": ops.gather_normalize(c.images, slot, c.lut, planar=self.planar),
            "word_ids": ops.gather_rows(c.tokens, pos).unsqueeze(1),
            "word_masks": ops.gather_rows(c.token_masks, pos).unsqueeze(1),
        }
        if K > 0:
            samples["neg_sents"] = [tuple(ds.all_sentences[p["neg"][k][0]][p["neg"][k][1]] for p in picks) for k in range(K)]
            samples["neg_word_ids"] = ops.gather_rows(c.tokens, neg).view(B, K, -1)
        targets = {
            "target": ops.gather_rows(c.targets.view(-1, S * S), ref).view(B, 1, S, S).to(torch.int64),
            "img_path": ops.gather_rows(c.img_path, ref).view(B),
            "sentences": [ds.all_sentences[i][p["choice"]] for i, p in zip(indices, picks)],
            "boxes": ops.gather_rows(c.boxes, ref),
            "orig_size": ops.gather_rows(c.orig_size, ref),
            "img_path_full": [c.files[i] for i in indices],
        }
        return samples, targets

    # ---- evaluation: one ref, all its sentences ---------------------------------------------------------------------
    def eval_batch(self, index):
        c, ds = self.cache, self.dataset
        lo, hi = int(c.sent_base[index]), int(c.sent_base[index + 1])
        slot = torch.tensor([c.slot_of[index]], dtype=torch.int64, device=c.device)
        samples = {
            "img": ops.gather_normalize(c.images, slot, c.lut, planar=self.planar),
            "word_ids": c.tokens[lo:hi].t().unsqueeze(0).unsqueeze(0),        # [1, 1, L, S_ref]
            "word_masks": c.token_masks[lo:hi].t().unsqueeze(0).unsqueeze(0),
        }
        m = c.eval_targets[index]
        targets = {
            "target": m.view(1, 1, *m.shape).to(torch.int64),
            "img_path": c.img_path[index].view(1),
            "sentences": [(s,) for s in ds.all_sentences[index]],
            "boxes": c.boxes[index:index + 1],
            "orig_size": c.orig_size[index:index + 1],
            "img_path_full": [c.files[index]],
        }
        return samples, targets
