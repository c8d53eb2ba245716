"""Stage-1 evaluation on MI355X -- `validate` with the reference's signature and return value
(validate.py:130-249): per (image, sentence) response map -> bilinear to the annotation size (align_corners=True)
-> max-normalise -> threshold 1e-9 -> I/U, IoU, pointing-game hit.

Differences that do not change the returned numbers:
  * the RN50 trunk + vis_project run once per image and are reused for all of its sentences (the reference
    recomputes them per sentence, validate.py:173-179);
  * resize / normalise / threshold / counting run in one device kernel chain (ops.eval_post) -- one host sync per
    sentence instead of several;
  * the box metrics that the reference only prints (cv2 contours + torchvision NMS, validate.py:195-201) are not
    produced (cv2 / torchvision are not part of this stack);
  * under torch.distributed the five accumulators are all-reduced so every rank returns the global numbers
    (the reference returns rank-local meters).
"""
import json
import os
import time

import numpy as np
import torch

from . import ops
from .config import cfg
from .utils.util import AverageMeter


def _host(x):
    """batch-dict field -> numpy (DataLoader gives CPU tensors / arrays, the HBM loader device tensors)"""
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def _graphed(net, img, L, cache):
    """hipGraph replay of the two halves of the eval forward, one capture per input shape (TRIS_HIPGRAPH=0 disables)."""
    if not cfg.hipgraph:
        return None
    key = (tuple(img.shape), int(L))
    if key not in cache:
        from .graphs import GraphedStage1Eval
        cache[key] = GraphedStage1Eval(net, tuple(img.shape), int(L))
    return cache[key]


def isCorrectHit(bbox_annot, heatmap, gt_mask=None):
    """validate.py:106-117: arg-max point inside any GT box (hit) / on the GT mask (hitm)."""
    max_loc = np.unravel_index(np.argmax(heatmap, axis=None), heatmap.shape)
    hitm = 1 if (gt_mask is not None and bool(gt_mask[max_loc[0], max_loc[1]])) else 0
    for bbox in bbox_annot:
        if bbox[0] <= max_loc[1] <= bbox[2] and bbox[1] <= max_loc[0] <= bbox[3]:
            return 1, max_loc, hitm
    return 0, max_loc, hitm


def get_scores(clip_model, fg_224_eval, word_id):
    """validate.py:120-127: cosine logits of aux-CLIP image features vs text features [N1, N2]."""
    f_i = ops.l2norm(clip_model.encode_image(fg_224_eval))
    f_t = ops.l2norm(clip_model.encode_text(word_id)[1])
    return ops.matmul(f_i, f_t, tB=True)


@torch.no_grad()
def validate(args, data_loader, model, local_rank=0, visualize=False, logger=None, save_cam=False):
    """TRIS_EVAL_GROUP (default 16) refs are evaluated per pass: ONE trunk call on the stacked images, ONE text-encoder call on
    all of their sentences, the heads per (image, sentence) pair, one host sync per group -- under ops.batch_invariant(), which
    makes every row of every dense product independent of the rows it shares a launch with, so the returned (oIoU, mIoU, hit)
    are bit-for-bit those of the one-ref-at-a-time loop (TRIS_EVAL_GROUP=1: the reference's loop structure, trunk and sentence
    halves replayed from hipGraphs)."""
    num_steps = len(data_loader)
    model.eval()
    net = model.module if hasattr(model, "module") else model
    say = logger.info if logger is not None else print
    say("Starting validation without PRMS")
    if save_cam and args.name_save_dir:
        os.makedirs(args.name_save_dir, exist_ok=True)
    if save_cam and args.cam_save_dir:
        os.makedirs(args.cam_save_dir, exist_ok=True)
    batch_time, mIOU_meter = AverageMeter(), AverageMeter()
    st = {"I": 0, "U": 0, "n_sent": 0, "hit": 0}
    cam_out_name = []
    graphs = {}
    group = max(1, int(cfg.eval_group))
    end = time.time()

    def account(idx, j, img_id, bbox, I, U, am, cam, n_img):
        """one (image, sentence) result -> the meters, in the reference's order (validate.py:180-236)"""
        st["n_sent"] += 1
        st["I"] += I
        st["U"] += U
        mIOU_meter.update(I / U if U > 0 else 0.0, n_img)
        y, x = divmod(am, cam.shape[1])
        for b in bbox:
            if b[0] <= x <= b[2] and b[1] <= y <= b[3]:
                st["hit"] += 1
                break
        if args.cam_save_dir is not None and save_cam:
            np.save(os.path.join(args.cam_save_dir, f"{idx}_{j}_{img_id}.npy"), cam.cpu().numpy())
        if args.name_save_dir is not None and save_cam:
            cam_out_name.append(f"{idx}_{j}_{img_id}")

    def report(idx):
        say(f"Test: [{idx:4d}/{num_steps}] | mIOU {100 * mIOU_meter.avg:.3f} | Overall IOU "
            f"{100 * float(st['I']) / max(float(st['U']), 1.0):.3f} | Hit {st['hit'] / max(st['n_sent'], 1) * 100:.3f} | "
            f"Time {batch_time.val:.3f} ({batch_time.avg:.3f})")

    def flush(refs):
        """refs: [(idx, img_id, img [1,3,H,W], word_ids [1,L,S], tgt u8 [oH,oW], bbox)] with one image shape"""
        nonlocal end
        if not refs:
            return
        imgs = torch.cat([r[2] for r in refs], 0)
        vis = net.encode_visual(imgs)
        ids = torch.cat([r[3][0].t() for r in refs], 0).contiguous()                       # [sum S, L]
        owner = [i for i, r in enumerate(refs) for _ in range(r[3].size(-1))]
        maps = net.forward_pairs(vis, ids, owner, imgs.shape[2])                           # [sum S, 1, H, W]
        ius, cams, k = [], [], 0
        for r in refs:
            for _ in range(r[3].size(-1)):
                iu, cam = ops.eval_post(maps[k:k + 1], r[4])
                ius.append(iu)
                cams.append(cam)
                k += 1
        vals = torch.stack(ius).tolist()                                                   # the one host sync of the group
        k = 0
        for idx, img_id, img, word_ids, tgt, bbox in refs:
            for j in range(word_ids.size(-1)):
                I, U, am = vals[k]
                account(idx, j, img_id, bbox, I, U, am, cams[k], img.size(0))
                k += 1
            batch_time.update((time.time() - end) / len(refs))
            if idx % args.print_freq == 0:
                report(idx)
        end = time.time()

    with ops.batch_invariant():
        pending = []
        for idx, (samples, targets) in enumerate(data_loader):
            img_id = int(_host(targets["img_path"]).reshape(-1)[0]) if "img_path" in targets else idx
            word_ids = samples["word_ids"].squeeze(1).cuda(local_rank, non_blocking=True)      # [1, L, S]
            img = samples["img"].cuda(local_rank, non_blocking=True)                           # [1, 3, H, W]
            target = targets["target"].cuda(local_rank, non_blocking=True)
            tgt = (target.reshape(target.shape[-2:]) != 0).to(torch.uint8)
            bbox = _host(targets["boxes"]).reshape(-1, 4) if "boxes" in targets else np.zeros((0, 4))
            if group > 1 and img.shape[0] == 1:
                if pending and (pending[0][2].shape != img.shape or pending[0][3].shape[1] != word_ids.shape[1]):
                    flush(pending)
                    pending = []
                pending.append((idx, img_id, img, word_ids, tgt, bbox))
                if len(pending) == group:
                    flush(pending)
                    pending = []
                continue
            gr = _graphed(net, img, word_ids.shape[1], graphs)
            if gr is not None:
                gr.visual(img)                                                                 # once per image (hipGraph)
            else:
                vis = net.encode_visual(img)
            for j in range(word_ids.size(-1)):
                wid = word_ids[:, :, j].contiguous()
                out = gr.sentence(wid) if gr is not None else net.forward_cached(vis, wid, img.shape[2])  # [1,1,H,W]
                iu, cam = ops.eval_post(out, tgt)
                I, U, am = iu.tolist()                                                         # the one host sync
                account(idx, j, img_id, bbox, I, U, am, cam, img.size(0))
            batch_time.update(time.time() - end)
            end = time.time()
            if idx % args.print_freq == 0:
                report(idx)
        flush(pending)
    I_sum, U_sum, n_sent, hit_acc = st["I"], st["U"], st["n_sent"], st["hit"]
    if args.name_save_dir is not None and save_cam:
        with open(os.path.join(args.name_save_dir, f"{args.dataset}_train_cam_name.json"), "w") as f:
            f.write(json.dumps(cam_out_name))
    acc = torch.tensor([float(I_sum), float(U_sum), float(mIOU_meter.sum), float(mIOU_meter.count), float(hit_acc),
                        float(n_sent)], dtype=torch.float64, device="cuda")
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc)
    I_t, U_t, iou_sum, iou_cnt, hits, nsent = acc.tolist()
    overall_IoU = 100 * I_t / max(U_t, 1.0)
    mIOU = torch.tensor(100 * iou_sum / max(iou_cnt, 1.0))
    hit = 100 * hits / max(nsent, 1.0)
    say(f"Test: mIOU {float(mIOU):.5f}  Overall IOU {overall_IoU:.5f}  HiT {hit:.3f}")
    return overall_IoU, mIOU, hit


@torch.no_grad()
def validate_same_sentence(args, data_loader, model, local_rank=0, visualize=False, logger=None, save_cam=False,
                           clip_model=None):
    """PRMS evaluation (validate.py:252-387): for every ref, each sentence's response map is scored by the aux CLIP
    against ALL sentences of the ref (sum of cosines of the CLIP-masked image with each sentence); the best-scoring
    map is used for every sentence of the ref (so I, U, hit are weighted by the sentence count exactly as the reference
    does, :336-344).  Here the S foreground images go through the aux ViT as ONE batch and the S sentences through the
    text tower once (the reference runs S*S single forwards)."""
    from .CLIP import clip as _clip
    num_steps = len(data_loader)
    model.eval()
    net = model.module if hasattr(model, "module") else model
    say = logger.info if logger is not None else print
    say("Starting validation with PRMS")
    save_cam = bool(getattr(args, "save_cam", save_cam))
    if save_cam and args.name_save_dir:
        os.makedirs(args.name_save_dir, exist_ok=True)
    if save_cam and args.cam_save_dir:
        os.makedirs(args.cam_save_dir, exist_ok=True)
    if clip_model is None:
        clip_model, _ = _clip.load("ViT-B/32", device="cuda", jit=False, txt_length=args.max_query_len)
    clip_model.eval()
    R = 224
    mIOU_meter = AverageMeter()
    I_sum = U_sum = 0.0
    n_sent = hit_acc = 0
    cam_out_name = []
    for idx, (samples, targets) in enumerate(data_loader):
        img_id = int(_host(targets["img_path"]).reshape(-1)[0]) if "img_path" in targets else idx
        word_ids = samples["word_ids"].squeeze(1).cuda(local_rank, non_blocking=True)      # [1, L, S]
        img = samples["img"].cuda(local_rank, non_blocking=True)
        target = targets["target"].cuda(local_rank, non_blocking=True)
        tgt = (target.reshape(target.shape[-2:]) != 0).to(torch.uint8)
        bbox = np.asarray(targets["boxes"].cpu() if torch.is_tensor(targets.get("boxes")) else targets.get("boxes", [])).reshape(-1, 4)
        S = word_ids.size(-1)
        n_sent += S
        ids = word_ids[0].t().contiguous()                                                 # [S, L]
        vis = net.encode_visual(img)
        maps = [net.forward_cached(vis, ids[j:j + 1], img.shape[2]) for j in range(S)]     # S x [1,1,H,W]
        outs = torch.cat(maps, 0)                                                          # [S,1,H,W]
        img224 = ops.resize_bilinear(img, (R, R), True) if img.shape[2] != R else img
        cam224 = ops.resize_bilinear(outs, (R, R), True) if img.shape[2] != R else outs
        patches = ops.fg_patches(cam224, img224.expand(S, -1, -1, -1).contiguous(), clip_model.visual.patch_size)
        f_i = ops.l2norm(clip_model.visual.forward_patches(patches))                       # [S,E]
        f_t = ops.l2norm(clip_model.encode_text(ids)[1])                                   # [S,E]
        score = ops.matmul(f_i, f_t, tB=True).sum(dim=1)                                   # [S]  (validate.py:327-329)
        best = int(torch.argmax(score).item())                                             # first maximum, as `>` does
        iu, cam = ops.eval_post(maps[best], tgt)
        I, U, am = iu.tolist()
        I_sum += float(I) * S * S
        U_sum += float(U) * S * S
        mIOU_meter.update(I / U if U > 0 else 0.0, img.size(0) * S)
        y, x = divmod(am, cam.shape[1])
        hit = 0
        for b in bbox:
            if b[0] <= x <= b[2] and b[1] <= y <= b[3]:
                hit = 1
                break
        hit_acc += hit * S
        if args.cam_save_dir is not None and save_cam:
            np.save(os.path.join(args.cam_save_dir, f"{idx}_{img_id}.npy"), cam.cpu().numpy())
        if args.name_save_dir is not None and save_cam:
            cam_out_name.append(f"{idx}_{img_id}")
        if idx % args.print_freq == 0:
            say(f"Test: [{idx:4d}/{num_steps}] | mIOU {100 * mIOU_meter.avg:.3f} | Overall IOU "
                f"{100 * I_sum / max(U_sum, 1.0):.3f} | Hit {hit_acc / max(n_sent, 1) * 100:.3f}")
    if args.name_save_dir is not None and save_cam:
        with open(os.path.join(args.name_save_dir, f"{args.dataset}_train_names.json"), "w") as f:
            f.write(json.dumps(cam_out_name))
    overall_IoU = 100 * I_sum / max(U_sum, 1.0)
    mIOU = torch.tensor(100 * mIOU_meter.avg)
    hit = 100 * hit_acc / max(n_sent, 1)
    say(f"Test: mIOU {float(mIOU):.5f}  Overall IOU {overall_IoU:.5f}  HiT {hit:.3f}")
    return overall_IoU, mIOU, hit
