"""ReferDataset with the reference's constructor, attributes and sample format (dataset/ReferDataset.py:36-252).

Differences in structure, not behaviour: the text side of a sample (`sample_text`) is separated from the pixel side
(`load_image_and_mask`) so that the HBM pipeline (tris_amd.dataset.hbm) can draw the sentences on the host -- with the
reference's exact `np.random.choice` call sequence -- while the pixels come from the uint8 cache on the GPU.
`__getitem__` composes the two and returns the reference's `(samples, targets)` dicts.
"""
import os

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from ..CLIP import clip
from .refer import REFER


class ReferDataset(data.Dataset):
    def __init__(self, refer_data_root="./data", dataset="refcoco", splitBy="unc", bert_tokenizer="clip",
                 image_transforms=None, max_tokens=20, split="train", eval_mode=True, size=448, scales=False,
                 negative_samples=0, positive_samples=1, pseudo_path=None, tokenizer=None):
        self.clip = "clip" in bert_tokenizer
        self.negative_samples = negative_samples
        self.positive_samples = positive_samples
        self.classes = []
        self.image_transforms = image_transforms
        self.split = split
        self.refer = REFER(refer_data_root, dataset, splitBy)
        self.scales = scales
        self.size = size
        self.pseudo_path = pseudo_path
        if pseudo_path is not None:
            raise NotImplementedError("pseudo_path feeds Stage-2 training, outside the Stage-1 path (SURVEY.md 8f)")
        self.max_tokens = max_tokens
        self.ref_ids = self.refer.getRefIds(split=split)
        self.imgs = [self.refer.Imgs[i] for i in self.refer.getImgIds(self.ref_ids)]
        self.tokenizer = tokenizer or clip.tokenize   # `tokenizer`: extension (same call signature as clip.tokenize)
        self.eval_mode = eval_mode

        # per ref: token rows ([1, L] each), their >0 masks, raw sentences  (ReferDataset.py:94-119)
        self.input_ids, self.word_masks, self.all_sentences = [], [], []
        self.refid2index = {r: i for i, r in enumerate(self.ref_ids)}
        for r in self.ref_ids:
            ids, masks, raws = [], [], []
            for sent in self.refer.Refs[r]["sentences"]:
                row = self.tokenizer(sent["sent"]).squeeze(0)[: self.max_tokens].numpy()
                ids.append(torch.tensor(row).unsqueeze(0))
                masks.append(torch.tensor(np.array(row > 0, dtype=int)).unsqueeze(0))
                raws.append(sent["sent"])
            self.input_ids.append(ids)
            self.word_masks.append(masks)
            self.all_sentences.append(raws)

    def __len__(self):
        return len(self.ref_ids)

    # ---- pixel side ----------------------------------------------------------------------------------------------
    def image_record(self, index):
        ref_id = self.ref_ids[index]
        img_id = self.refer.getImgIds(ref_id)
        return ref_id, img_id[0], self.refer.Imgs[img_id[0]]

    def load_image(self, index):
        _, _, rec = self.image_record(index)
        return Image.open(os.path.join(self.refer.IMAGE_DIR, rec["file_name"])).convert("RGB")

    def load_mask(self, index):
        """-> ('P' PIL mask of the ref's pixels covered by exactly one polygon, bbox x1y1x2y2)"""
        ref = self.refer.loadRefs(self.ref_ids[index])[0]
        bbox = np.array(self.refer.Anns[ref["ann_id"]]["bbox"], dtype=int)
        bbox[2], bbox[3] = bbox[0] + bbox[2], bbox[1] + bbox[3]
        ref_mask = np.array(self.refer.getMask(ref)["mask"])
        annot = np.zeros(ref_mask.shape)
        annot[ref_mask == 1] = 1
        return Image.fromarray(annot.astype(np.uint8), mode="P"), bbox

    def load_pil(self, index):
        """-> (RGB PIL image, mask, bbox)"""
        return (self.load_image(index),) + self.load_mask(index)

    # ---- text side (all the randomness of a sample, in the reference's call order) ---------------------------------
    def negative_pool(self, index):
        """refs of the same image, other than this one, that belong to this split -- scanning stops at the first ref of
        the image that is not in the split (the reference's try/except-break, ReferDataset.py:198-206)"""
        ref_id, img_id, _ = self.image_record(index)
        pool = []
        for item in self.refer.imgToRefs[img_id]:
            other = item["ref_id"]
            if other == ref_id:
                continue
            if other not in self.refid2index:
                break
            pool.append(self.refid2index[other])
        return pool

    def sample_text(self, index):
        """-> dict(choice, neg=[(ref index, sentence index)...]) drawn with np.random.choice exactly as the reference's
        training branch does (ReferDataset.py:172-229): one draw for the sentence, then two per negative."""
        n = len(self.input_ids[index])
        choice = np.random.choice(n)
        neg = []
        if self.negative_samples > 0:
            pool = self.negative_pool(index)
            sentence = self.all_sentences[index][choice]
            while len(neg) < self.negative_samples:
                if pool:
                    j = pool[np.random.choice(len(pool))]
                    s = np.random.choice(len(self.input_ids[j]))
                    neg.append((j, s))
                else:   # no other ref on this image: any sentence of the dataset that differs from the positive one
                    j = np.random.choice(len(self.input_ids))
                    s = np.random.choice(len(self.input_ids[j]))
                    if self.all_sentences[j][s] != sentence:
                        neg.append((j, s))
        return {"choice": choice, "neg": neg}

    def __getitem__(self, index):
        img, annot, bbox = self.load_pil(index)
        _, _, rec = self.image_record(index)
        w, h = annot.size
        if self.image_transforms is not None:
            img, target = self.image_transforms(img, annot)
        else:
            target = annot
        samples = {"img": img}
        if self.eval_mode:   # every sentence of the ref, stacked on a trailing axis
            samples["word_ids"] = torch.cat([e.unsqueeze(-1) for e in self.input_ids[index]], dim=-1)
            samples["word_masks"] = torch.cat([a.unsqueeze(-1) for a in self.word_masks[index]], dim=-1)
            sentences = list(self.all_sentences[index])
            neg = None
        else:
            pick = self.sample_text(index)
            c = pick["choice"]
            samples["word_ids"] = self.input_ids[index][c]
            samples["word_masks"] = self.word_masks[index][c]
            sentences = self.all_sentences[index][c]
            neg = pick["neg"]
        if self.negative_samples > 0:
            # (in eval mode the reference would hit an unbound name here; evaluation always runs with 0 negatives)
            samples["neg_sents"] = [self.all_sentences[j][s] for j, s in neg]
            samples["neg_word_ids"] = torch.cat([self.input_ids[j][s] for j, s in neg], dim=0)
        targets = {
            "target": target.unsqueeze(0),
            "img_path": int(rec["file_name"].split(".")[0].split("_")[-1]),
            "sentences": sentences,
            "boxes": bbox,
            "orig_size": np.array([h, w]),
            "img_path_full": rec["file_name"],
        }
        return samples, targets
