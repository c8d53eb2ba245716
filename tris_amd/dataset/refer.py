"""REFER index over refs(<splitBy>).p + instances.json -- the query surface of the reference's dataset/refer.py:44-291
that ReferDataset and the evaluation scripts use (display helpers :235-277, 329+ are out of scope).

Same attribute / method names and return conventions as the reference so dataset code written against it runs
unchanged; masks come from tris_amd.dataset.cocomask (pycocotools is not a dependency here).
"""
import json
import os.path as osp
import pickle
from collections import defaultdict

import numpy as np

from . import cocomask

_COCO_SETS = ("refcoco", "refcoco+", "refcocog")


def _as_list(v):
    return v if isinstance(v, list) else [v]


class REFER:
    def __init__(self, data_root="./datasets", dataset="refcoco", splitBy="unc"):
        self.ROOT_DIR = osp.abspath(osp.dirname(__file__))
        self.DATA_DIR = osp.join(data_root, "refer", dataset)
        if dataset in _COCO_SETS:
            self.IMAGE_DIR = osp.join(data_root, "train2014")
        elif dataset == "refclef":
            self.IMAGE_DIR = osp.join(data_root, "images/saiapr_tc-12")
        else:
            raise SystemExit("No refer dataset is called [%s]" % dataset)   # the reference prints and sys.exit()s
        with open(osp.join(self.DATA_DIR, "refs(" + splitBy + ").p"), "rb") as f:
            refs = pickle.load(f)
        with open(osp.join(self.DATA_DIR, "instances.json"), "r") as f:
            inst = json.load(f)
        self.data = {"dataset": dataset, "refs": refs, "images": inst["images"],
                     "annotations": inst["annotations"], "categories": inst["categories"]}
        self.createIndex()

    def createIndex(self):
        d = self.data
        self.Anns = {a["id"]: a for a in d["annotations"]}
        self.Imgs = {i["id"]: i for i in d["images"]}
        self.Cats = {c["id"]: c["name"] for c in d["categories"]}
        img_to_anns, img_to_refs, cat_to_refs = defaultdict(list), defaultdict(list), defaultdict(list)
        for a in d["annotations"]:
            img_to_anns[a["image_id"]].append(a)
        self.Refs, self.refToAnn, self.annToRef = {}, {}, {}
        self.Sents, self.sentToRef, self.sentToTokens = {}, {}, {}
        for ref in d["refs"]:
            rid = ref["ref_id"]
            self.Refs[rid] = ref
            img_to_refs[ref["image_id"]].append(ref)
            cat_to_refs[ref["category_id"]].append(ref)
            self.refToAnn[rid] = self.Anns[ref["ann_id"]]
            self.annToRef[ref["ann_id"]] = ref
            for sent in ref["sentences"]:
                self.Sents[sent["sent_id"]] = sent
                self.sentToRef[sent["sent_id"]] = ref
                self.sentToTokens[sent["sent_id"]] = sent["tokens"]
        self.imgToAnns, self.imgToRefs, self.catToRefs = dict(img_to_anns), dict(img_to_refs), dict(cat_to_refs)

    # ---- id queries ------------------------------------------------------------------------------------------------
    def getRefIds(self, image_ids=[], cat_ids=[], ref_ids=[], split=""):
        image_ids, cat_ids, ref_ids = _as_list(image_ids), _as_list(cat_ids), _as_list(ref_ids)
        refs = self.data["refs"]
        if image_ids:
            refs = [self.imgToRefs[i] for i in image_ids]   # (a list of lists, as in the reference: :153)
        if cat_ids:
            refs = [r for r in refs if r["category_id"] in cat_ids]
        if ref_ids:
            refs = [r for r in refs if r["ref_id"] in ref_ids]
        if split:
            if split in ("testA", "testB", "testC"):
                refs = [r for r in refs if split[-1] in r["split"]]
            elif split in ("testAB", "testBC", "testAC"):
                refs = [r for r in refs if r["split"] == split]
            elif split == "test":
                refs = [r for r in refs if "test" in r["split"]]
            elif split in ("train", "val"):
                refs = [r for r in refs if r["split"] == split]
            else:
                raise SystemExit("No such split [%s]" % split)
        return [r["ref_id"] for r in refs]

    def getAnnIds(self, image_ids=[], cat_ids=[], ref_ids=[]):
        image_ids, cat_ids, ref_ids = _as_list(image_ids), _as_list(cat_ids), _as_list(ref_ids)
        if image_ids:
            anns = [a for i in image_ids if i in self.imgToAnns for a in self.imgToAnns[i]]
        else:
            anns = self.data["annotations"]
        if cat_ids:
            anns = [a for a in anns if a["category_id"] in cat_ids]
        return [a["id"] for a in anns]   # ref_ids never narrows the result in the reference either (:190-192)

    def getImgIds(self, ref_ids=[]):
        ref_ids = _as_list(ref_ids)
        if ref_ids:
            return list(set(self.Refs[r]["image_id"] for r in ref_ids))
        return self.Imgs.keys()

    def getCatIds(self):
        return self.Cats.keys()

    # ---- record loaders ------------------------------------------------------------------------------------------
    def _load(self, table, ids, scalar_types):
        if isinstance(ids, list):
            return [table[i] for i in ids]
        if isinstance(ids, scalar_types):
            return [table[ids]]
        return None

    def loadRefs(self, ref_ids=[]):
        return self._load(self.Refs, ref_ids, (int,))

    def loadAnns(self, ann_ids=[]):
        return self._load(self.Anns, ann_ids, (int, str))

    def loadImgs(self, image_ids=[]):
        return self._load(self.Imgs, image_ids, (int,))

    def loadCats(self, cat_ids=[]):
        return self._load(self.Cats, cat_ids, (int,))

    def getRefBox(self, ref_id):
        return self.refToAnn[ref_id]["bbox"]   # [x, y, w, h]

    def getMask(self, ref):
        """-> {'mask': uint8 [h,w] (sum over the ref's polygons), 'area': int}"""
        ann = self.refToAnn[ref["ref_id"]]
        image = self.Imgs[ref["image_id"]]
        seg = ann["segmentation"]
        if isinstance(seg[0], list):
            rle = cocomask.frPyObjects(seg, image["height"], image["width"])
        else:
            rle = seg
        m = cocomask.decode(rle)
        m = np.sum(m, axis=2).astype(np.uint8)
        return {"mask": m, "area": sum(cocomask.area(rle))}
