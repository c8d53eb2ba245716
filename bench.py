"""Benchmark of the TRIS Stage-1 training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full Stage-1 training iteration (TRIS forward, CLIP-guided fg / negative-sample / cls losses
through the frozen aux ViT-B/32, backward, [gradient all-reduce], AdamW, LR schedule) on 48 synthetic 320x320
images + 20-token sentences + 3 negatives per image PER GPU (weak scaling), fp32 storage, seed-filled weights.

Arithmetic of the dense products: `value` is measured in "h2" (two fp16 pieces per operand, fp32 accumulate, fp32-class accuracy
-- the whole parity suite runs in it: tests/test_gpu_parity.py) unless TRIS_GEMM_MODE selects another; `value_x3` -- the same K
steps in the split-bf16 x3 arithmetic of the earlier rounds, same process, same model -- is printed beside it.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
os.environ.setdefault("TRIS_RANDOM_INIT", "1")  # synthetic weights (seed-fill): no CLIP checkpoint needed
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PER_GPU_BATCH = 48
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA; x3 spends 6 bf16 MFMAs per fp32-accurate product, h2 3 f16 MFMAs
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (metric config: 48)")
    ap.add_argument("--backbone", default="clip-RN50", choices=["clip-RN50", "clip-ViT-B/16"],
                    help="clip-RN50 = the metric configuration (BASELINE configs[2]/[3]); clip-ViT-B/16 = configs[4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batches", default="48,2", help="batch sizes of the CPU-baseline leg (first = reported value)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the side measurements (input pipeline, evaluation, "
                    "other issue form, ViT-B/16, one-rank distributed path)")
    ap.add_argument("--headline-only", action="store_true", help="profiling runs: the timed steps of `value` and nothing else "
                    "(no x3 leg, no instrumented roofline step)")
    return ap.parse_args()


def cpu_baseline(batches=(48, 2), timed=3):
    """The oracle's reference-faithful Stage-1 step (incl. the reference's redundant work: attnpool, second aux image
    forward, per-image negative-text loop, aux weight gradients) on the host cores, as BASELINE.md section 3 / SURVEY.md 8d
    prescribe: the metric's batch (48) and config[0]'s batch (2), 1 warm-up + >= 3 timed steps each, median reported."""
    import statistics
    from oracle import tris_oracle as O
    from tris_amd.utils.shapes import aux_state_dict_spec, empty_state_dict, tris_state_dict_spec
    from tris_amd.utils.synth import seed_fill, synthetic_batch
    # the GPU node's host is 2 x 64-core EPYC (256 hw threads); PyTorch-CPU throughput on this workload PEAKS at 32
    # threads there (64 threads: 0.55x, 128 threads: 0.2x -- measured, tools/cpu_probe.py), so 32 is the fair setting
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    res = {}
    for B in batches:
        sd = seed_fill(empty_state_dict(tris_state_dict_spec()), 1234)
        aux = seed_fill(empty_state_dict(aux_state_dict_spec()), 4321)
        b = synthetic_batch(B, 320, 20, 3, seed=7)
        state = {}
        O.train_step(sd, aux, b, state=state, faithful=True)  # warm-up
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            O.train_step(sd, aux, b, state=state, faithful=True)
            ts.append(time.perf_counter() - t0)
        res[B] = {"median_s_per_step": round(statistics.median(ts), 3), "img_per_s": round(B / statistics.median(ts), 4),
                  "steps_s": [round(t, 3) for t in ts]}
    main_b = batches[0]
    return {"value": res[main_b]["img_per_s"], "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{timed} timed + 1 warm-up reference-faithful fp32 train steps of oracle/tris_oracle.py at batch "
                      f"{main_b} (the metric's batch; same synthetic 320px / 20-token / 3-negative workload; median "
                      f"{res[main_b]['median_s_per_step']} s/step; host has {os.cpu_count()} hw threads, {cores} used "
                      f"because more threads run slower)",
            "by_batch": {str(k): v for k, v in res.items()}}


def _mfma_peak(mode):
    return F32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS / (6.0 if mode == "x3" else 3.0)


def _arith_text(mode):
    return {"f32": "v_mfma_f32_32x32x2_f32 (f32 in)",
            "x3": "split-bf16 x3: 6 x v_mfma_f32_32x32x16_bf16 per fp32-accurate product; achieved / peak are in fp32-equivalent "
                  "FLOPs (peak = 2500 TFLOP/s bf16 dense / 6; the f32-input MFMA peak is 157.3)",
            "h2": "h2: two fp16 pieces per operand (residual pre-scaled by 2^11, one power-of-two scale per tensor from a device-side "
                  "amax), 3 x v_mfma_f32_32x32x16_f16 per fp32-accurate product, two fp32 accumulators; achieved / peak are in "
                  "fp32-equivalent FLOPs (peak = 2500 TFLOP/s f16 dense / 3 = 833.3; the f32-input MFMA peak is 157.3)"}[mode]


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    rccl = None
    if world > 1:
        # the collective backend really spans N ranks on N distinct devices: one all-reduce of ones and an all-gather of device ids
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        devs = [torch.zeros(1, device="cuda", dtype=torch.int64) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([local], device="cuda", dtype=torch.int64))
        rccl = {"backend": dist.get_backend(), "ranks_seen": int(ones.item()), "devices": [int(d.item()) for d in devs]}
        assert rccl["backend"] == "nccl" and rccl["ranks_seen"] == a.gpus and len(set(rccl["devices"])) == a.gpus, \
            f"RCCL saw {rccl} but --gpus {a.gpus}"
    from tris_amd import ops
    from tris_amd.config import cfg
    if world > 1:
        # which stream runs on which hardware queue, with the collective backend's own stream in the picture (ops.place_streams)
        cs = ops.place_streams()
        if cs is not None:
            torch.cuda.set_stream(cs)
    elif cfg.own_stream:
        # never compute on the process's DEFAULT stream: whatever is launched there -- eager kernels and hipGraphs alike -- runs
        # EXCLUSIVELY with respect to hipGraphs launched on other streams (tools/graph_step_overlap_probe2.py: every graph of the step,
        # and plain eager kernels, next to a graph of sleeping one-thread kernels on another stream: the sum of the two times, not the
        # maximum), i.e. the text towers replayed from graphs stop the trunk for as long as they run.  ops.compute_stream() hands out
        # one non-default stream per device.
        torch.cuda.set_stream(ops.compute_stream())
    torch.cuda.synchronize()

    from tris_amd.args import get_parser
    from tris_amd.CLIP import clip
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.optim import FusedAdamW
    from tris_amd.parallel import attach_reducer, convert_sync_batchnorm
    from tris_amd.train_stage1 import freeze_aux, train_step
    from tris_amd.utils.synth import seed_fill, synthetic_batch

    if not os.environ.get("TRIS_GEMM_MODE"):
        ops.set_gemm_mode("h2")                  # the package default (ops._ARITH); value_x3 is measured beside it
    if "TRIS_STEP_GRAPH" not in os.environ:
        cfg.step_graph = "seg"                   # `value` = the step as the trainer issues it (train_stage1.main): segmented hipGraph replay
    mode = ops.get_gemm_mode()
    QL = int(os.environ.get("TRIS_BENCH_QL", "20"))   # (20 = the metric's query length; the override exists for what-if experiments only)

    def build(backbone, distributed=False, force=False):
        args = get_parser().parse_args(["--backbone", backbone, "--size", "320", "--max_query_len", str(QL),
                                        "--negative_samples", "3", "--batch_size", str(a.batch), "--epoch", "15"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = TRIS(args).cuda().train()
        seed_fill(model.state_dict(), 1234)
        bb, new = model.trainable_parameters()
        opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr,
                         weight_decay=args.weight_decay)
        max_iter = 1000 * args.epoch
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: (1 - x / max_iter) ** 0.9)
        reducer = None
        if distributed:
            convert_sync_batchnorm(model)
            reducer = attach_reducer(model, opt, force=force)   # all-reduce segments launched from inside backward
        return args, model, opt, sched, reducer

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=QL)
    seed_fill(aux.state_dict(), 4321)
    freeze_aux(aux)
    args, model, opt, sched, reducer = build(a.backbone, distributed=world > 1)
    b = synthetic_batch(a.batch, 320, QL, 3, seed=7, rank=rank)
    img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()

    def step():
        return train_step(model, aux, opt, img, ids, neg, args, sched, reducer)

    def timed(fn, steps, warmup):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks"""
        for _ in range(warmup):
            fn()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        host = time.perf_counter() - t0      # the host's share: all K steps issued (it may have waited on a full queue)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, host, out

    step()   # one untimed priming step, always: first-encounter GEMM autotuning, allocator growth, RCCL channel set-up
    dt, host_issue, losses = timed(step, a.steps, a.warmup)
    loss_vals = losses.tolist()
    # h2 operand planes: the range tell-tale of the last step (plane tensors with > 1 % of their elements below the 2^-27 floor of their
    # scale: ops.h2_range_report) and how the products / gradients of a step travelled (ops.PL_STATS)
    h2_range = ops.h2_range_report() if (mode == "h2" and cfg.h2_planes) else None
    comm_exposed = reducer.exposed_ms() if reducer is not None else None   # compute-stream wait for collectives, last timed step
    transport = None
    if world > 1:
        from tris_amd import comm
        comm.check_errors()    # a SyncBatchNorm mailbox exchange that timed out must fail the run, not skew it
        transport = "mailbox" if any(m is not None for m in comm.Mailbox._by_group.values()) else "torch.distributed"

    # ---- the same K steps in the x3 arithmetic of the earlier rounds: same process, same model, same timing rule ----------------
    value_x3 = None
    if a.headline_only:
        if rank == 0:
            print(json.dumps({"value": round(world * a.batch * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 3),
                              "gemm_mode": mode, "headline_only": True}), flush=True)
        if world > 1:
            dist.barrier()
            __import__('tris_amd.comm', fromlist=['x']).shutdown()
            dist.destroy_process_group()
        return
    if mode == "h2":
        ops.set_gemm_mode("x3")
        try:
            step()                                   # (first-encounter autotuning of the x3 kernels)
            dt3, host3, l3 = timed(step, a.steps, a.warmup)
            value_x3 = {"value": round(world * a.batch * a.steps / dt3, 2), "ms_per_step": round(dt3 / a.steps * 1e3, 3),
                        "host_issue_ms_per_step": round(host3 / a.steps * 1e3, 3),
                        "losses_last_step": [round(v, 5) for v in l3.tolist()]}
        finally:
            ops.set_gemm_mode(mode)

    # ---- live roofline measurement of the dominant kernel family (one extra, untimed, instrumented step) ----
    # (kernels are timed one at a time: the stream overlap of the production step is switched off for this pass so that a
    # launch's HIP-event bracket measures that kernel alone, not whatever else shares the GPU with it)
    from tris_amd import _lib as _tl
    with cfg.override(step_graph="0", text_stream=False, wgrad_stream=False):
        step()
        fused0 = _tl.query("tris_splitk_fused_launches")
        ops.profile_begin()
        step()
        rec_all = ops.profile_end()
        fused_per_step = int(_tl.query("tris_splitk_fused_launches") - fused0)   # split-K products finished inside their own launch
    xa = [r for r in rec_all if r[0].startswith("xattn_fwd")]
    xb = [r for r in rec_all if r[0] == "xattn_bwd_px"]
    rec = [r for r in rec_all if not r[0].startswith("xattn")]
    fl = sum(r[1] for r in rec)
    ms = sum(r[2] for r in rec)
    nb = sum(r[3] for r in rec)
    kinds = {}
    for k, f, m, by in rec:
        e = kinds.setdefault(k, [0, 0.0, 0.0, 0.0])
        e[0] += 1
        e[1] += f
        e[2] += m
        e[3] += by
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = (time.perf_counter() - t1) * 1e3
    ach = fl / (ms * 1e-3) / 1e12
    peak = _mfma_peak(mode)
    hbm_alg = nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof = {"bound": "mfma" if ach / peak >= hbm_alg else "hbm", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None,
            # the family against its OTHER roof: bytes (every operand once, every output once) / family time / 8 TB/s
            "hbm_frac_algorithmic": round(hbm_alg, 4), "algorithmic_gbytes_per_step": round(nb / 1e9, 2),
            "bound_note": ("neither roof is near: the family is a mix of MFMA-heavy 3x3 / long-K products and short-K 1x1 products "
                           "that stream their outputs; `bound` names the roof it is closer to, `frac` is against the MFMA peak of the "
                           "arithmetic, hbm_frac_* against 8 TB/s; by_kind has both per kind (DESIGN.md section 5)"),
            # the same achieved rate against the f32-INPUT MFMA peak (what an fp32 product costs without the split)
            "peak_f32_mfma": F32_MFMA_PEAK_TFLOPS, "frac_vs_f32_mfma": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
            "arithmetic": _arith_text(mode),
            "kernel": "gemm_fast_kernel<BM,BN,A,B,EPI,PREC> + wgrad3x3_direct_kernel (MFMA GEMM / implicit-GEMM / direct 3x3 family)",
            "launches_per_step": len(rec), "kernel_ms_per_step": round(ms, 3), "fused_splitk_products_per_step": fused_per_step,
            "fused_epilogue_note": ("kinds *_bnbwd are data-gradient products whose epilogue also does the reduction pass of the "
                                    "BatchNorm backward that consumes them (their extra activation-sized streams are in the byte "
                                    "counts, no FLOPs are counted for them)"),
            "algorithmic_gflop_per_step": round(fl / 1e9, 1),
            "by_kind": {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2], 3),
                            "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 else None,
                            "hbm_frac_algorithmic": round(v[3] / (v[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[2] > 0 else None}
                        for k, v in kinds.items()}}

    # PMC-derived HBM traffic / matrix-pipe utilisation of the same kernel family: separate rocprofv3 --pmc passes of THIS command
    # (tools/closing_profiles.sh), committed under profiles/ -- read, not measured here; the files are named in the notes
    def _prof(name):   # this round's committed profile if present, else the last round's (the note names the file that was read)
        for rd in ("r6", "r5", "r4"):
            p = os.path.join(ROOT, "profiles", f"{rd}_{name}")
            if os.path.exists(p):
                return p, f"{rd}_{name}"
        return os.path.join(ROOT, "profiles", f"r6_{name}"), f"r6_{name}"
    for tag in ((mode,) if mode in ("h2", "x3") else ()):
        try:
            pfile, pname = _prof(f"{tag}_pmc_hbm_traffic.json")
            pmc = json.load(open(pfile))["gemm_family"]
            roof["traffic"] = int((pmc["fetch_bytes_per_step"] + pmc["write_bytes_per_step"]) / pmc["launches_per_step"])
            roof["hbm_frac_pmc"] = round((pmc["fetch_bytes_per_step"] + pmc["write_bytes_per_step"]) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["traffic_note"] = ("bytes per launch, averaged over the family: (FETCH_SIZE x2 + WRITE_SIZE) per step / launches per "
                                    f"step from profiles/{pname} (rocprofv3 --pmc pass of this command, "
                                    "tools/closing_profiles.sh; not re-measured in this run)")
        except Exception:
            pass
        try:
            sfile, sname = _prof(f"{tag}_pmc_mfma_util.json")
            sq = json.load(open(sfile))
            roof["mfma_utilisation_pmc"] = sq["families"]["gemm_family"]["mfma_utilisation"]
            roof["mfma_utilisation_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) over the family, "
                                             f"profiles/{sname} (rocprofv3 --pmc, kernels serialised, static tile choice)")
        except Exception:
            pass
    try:   # per-kind rate of the tuner's winners on an IDLE device (profiles/r5_autotune_log_<mode>.txt): the in-step loss to
        # co-scheduling with the other streams is the gap between these and by_kind above
        from tools.tune_report import idle_rates
        roof["idle_device_tflops"] = idle_rates(_prof(f"autotune_log_{mode}.txt")[0])
    except Exception:
        pass
    roof_xb = None
    if xb:   # the backward of the pair as one persistent launch (csrc/xattn_px.hip xattn_px_bwd_kernel + its preparation launch)
        P_, N_, C_ = 100, a.batch, 1024
        bb = a.batch * (5 * P_ * C_ + N_ * C_ + 5 * P_ * N_) * 4 + 3 * N_ * C_ * 4
        bms = sum(r[2] for r in xb) / len(xb)
        roof_xb = {"bound": "hbm", "achieved": round(bb / (bms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(bb / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "us": round(bms * 1e3, 1),
                   "algorithmic_bytes_per_call": bb,
                   "kernel": "xattn_bwd_planes_kernel + xattn_px_bwd_kernel (dQv, dKv, dVv and both soft-max backwards of the bilateral "
                             "cross attention in ONE persistent launch cut by pixel rows; the three [N, C] sums over images and pixels "
                             "follow as split-K products and are in the GEMM family)"}
    roof_x = None
    if xa:
        P_, N_, C_ = 100, a.batch, 1024
        xbytes = a.batch * (4 * P_ * C_ + N_ * C_) * 4 + 3 * N_ * C_ * 4
        fused = [r for r in xa if r[0] in ("xattn_fwd_fused", "xattn_fwd_px")]
        if fused and len(fused) == len(xa) and fused[0][0] == "xattn_fwd_px":
            xname = ("xattn_px_kernel (ONE persistent launch cut by pixel rows, csrc/xattn_px.hip; " +
                     ("h2: two fp16 pieces per operand, three MFMAs per product" if (mode == "h2" and cfg.xattn_h2) else
                      "x3: three bf16 pieces per operand, six MFMAs per product") +
                     ") + xattn_text_planes_kernel (the sentence operands as piece planes)")
        elif fused and len(fused) == len(xa):
            xname = "xattn_fused_kernel (ONE persistent launch, csrc/xattn_fused.hip) + xattn_text_planes_kernel (sentence bf16 planes)"
        elif mode in ("x3", "h2"):
            xname = "xattn_scores_x3_kernel + xattn_out_x3_kernel"
        else:
            xname = "xattn_scores_kernel + xattn_colsoftmax_kernel + xattn_out_kernel"
        xms = sum(r[2] for r in xa)
        roof_x = {"bound": "hbm", "achieved": round(xbytes / (xms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(xbytes / (xms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                  "kernel": xname + " (fused bilateral cross attention, forward, all launches, HIP events around the call)",
                  "single_launch": bool(fused and len(fused) == len(xa)),
                  "algorithmic_bytes_per_launch_pair": xbytes, "us": round(xms * 1e3, 1),
                  "bound_note": ("pixel-row form: one hand-off, 240 workgroups on 256 CUs, traffic ~1.2 x algorithmic; every workgroup reads "
                                 "the full sentence operands from L2 (h2: 0.57 MB of fp16 piece planes, x3: 0.86 MB, against 0.32 MB of its "
                                 "own HBM stream); the workgroup lifetime is 22.6 us in h2 (28.5 in x3), the rest of the call is the "
                                 "preparation launch (3-6 us) and the ramp / drain of 240 x 512-thread workgroups with 120 KB of LDS each "
                                 "(7 us): profiles/r5_xattn_phase_table.txt; 33-35 us on an idle device (0.31-0.33)")
                  if fused and len(fused) == len(xa) and fused[0][0] == "xattn_fwd_px" else None,
                  "mfma_tflops": round(sum(r[1] for r in xa) / (xms * 1e-3) / 1e12, 2)}
        try:   # HBM bytes of the cross-attention launches from the same --pmc passes
            import csv as _csv
            tr = 0
            xfile, xpname = _prof(f"{mode}_pmc_hbm_traffic.csv")
            trb = 0
            for r in _csv.DictReader(open(xfile)):
                if "xattn_" in r["Kernel"]:
                    by = int(r["FetchBytesPerStep(x2 corrected)"]) + int(r["WriteBytesPerStep"])
                    if "bwd" in r["Kernel"]:      # (xattn_px_bwd_kernel, xattn_bwd_planes_kernel: the backward's launches)
                        trb += by
                    else:
                        tr += by
            if trb and roof_xb is not None:
                roof_xb["traffic"] = trb
                roof_xb["traffic_note"] = (f"FETCH_SIZE x2 + WRITE_SIZE of the two launches (profiles/{xpname}): "
                                           f"{trb / roof_xb['algorithmic_bytes_per_call']:.2f}x the algorithmic bytes")
            if tr:
                roof_x["traffic"] = tr
                roof_x["traffic_note"] = (f"FETCH_SIZE x2 + WRITE_SIZE of the cross-attention launches of one forward (profiles/{xpname}, "
                                          f"rocprofv3 --pmc, B=48): {tr / xbytes:.2f}x the algorithmic bytes")
        except Exception:
            pass
    out = None
    if rank == 0:
        dtype = {"h2": "f32 storage/accumulate; products h2 = 2 x f16 pieces", "x3": "f32 storage/accumulate; products x3 = 3 x bf16 pieces",
                 "f32": "f32"}[mode]
        out = {"metric": f"Stage-1 training images/sec @320px bs48 (TRIS {a.backbone}, 3 negatives)",
               "value": round(world * a.batch * a.steps / dt, 2), "unit": "img/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": dtype, "gemm_mode": mode, "data": "synthetic",
               "config": {"workload": "Stage-1 train step, RefCOCOg-shaped synthetic batch: 48 img/GPU 320x320, "
                                      f"20-token query + 3 negative queries per image, {a.backbone} trunk + frozen aux "
                                      "CLIP ViT-B/32, AdamW (BASELINE.json " +
                                      ("configs[2]/[3])" if a.backbone == "clip-RN50" else
                                       "configs[4]; the reference defines no such model: parity unpinned, DESIGN.md)"),
                          "per_gpu_batch": a.batch, "global_batch": world * a.batch, "size": 320, "query_len": QL,
                          "negative_samples": 3, "parallelism": f"dp{world}", "sync_bn": world > 1, "sync_bn_transport": transport},
               "untimed_priming_steps": 1, "losses_last_step": [round(v, 5) for v in loss_vals], "ms_single_step_synced": round(one, 3),
               "host_issue_ms_per_step": round(host_issue / a.steps * 1e3, 3),
               "comm_exposed_ms": None if comm_exposed is None else round(comm_exposed, 3),
               "sparse_embed_exchange": bool(reducer is not None and reducer.sparse_embed),
               "streams": {"text_encoders_on_side_stream": cfg.text_stream, "weight_gradients_on_side_stream": cfg.wgrad_stream,
                           "compute_on_own_stream": bool(cfg.own_stream and world == 1)},
               "h2_operand_planes": bool(cfg.h2_planes and mode == "h2"),
               "h2_out_of_range_operands": None if h2_range is None else h2_range["out_of_range_operands"], "h2_range": h2_range,
               "step_issue": {"0": "eager launches", "1": "one hipGraph", "seg": "chain of single-stream hipGraphs"}[cfg.step_graph],
               "rccl": rccl,    # (N > 1: what the collective backend saw -- N ranks on N distinct devices, asserted at start-up)
               "roofline": roof, "roofline_xattn": roof_x, "roofline_xattn_bwd": roof_xb}
        if value_x3 is not None:
            out["value_x3"] = value_x3["value"]
            out["x3"] = dict(value_x3, note="the same K steps in the split-bf16 x3 arithmetic (the arithmetic of `value` in rounds 1-3), "
                                            "same process, same model and optimiser state continuing, same barrier / synchronize rule")
    if rank == 0 and world == 1:
        if not a.no_pipeline:
            # input pipeline ahead of the step (HBM-resident uint8 dataset -> batch), measured apart from `value`
            try:
                from tools.pipeline_bench import measure as pipeline_measure
                out["input_pipeline"] = pipeline_measure(batch=a.batch)
            except Exception as e:  # reported, never hidden
                out["input_pipeline"] = {"error": repr(e)}
        if not a.no_pipeline and a.backbone == "clip-RN50":
            # evaluation throughput of configs[1] (validate.py's loop): one ref at a time vs batched, identical metrics
            try:
                from tools.eval_throughput import measure as eval_measure
                out["eval"] = eval_measure()
                out["eval_refs_per_s"] = out["eval"]["eval_refs_per_s"]
            except Exception as e:  # reported, never hidden
                out["eval"] = {"error": repr(e)}
        if not a.no_pipeline:
            # the other way of ISSUING the same step (bit-identical results, tests/test_gpu_step_graph.py), timed like `value`
            other = "seg" if cfg.step_graph == "0" else "0"
            try:
                del losses
                with cfg.override(step_graph=other):
                    dtg, hi, _ = timed(step, a.steps, 3)
                out["step_graph_segmented" if other == "seg" else "step_eager"] = {
                    "ms_per_step": round(dtg / a.steps * 1e3, 3), "host_issue_ms_per_step": round(hi / a.steps * 1e3, 3),
                    "img_per_s": round(a.batch * a.steps / dtg, 2), "steps": a.steps,
                    "note": ("the step replayed from a chain of single-stream hipGraphs (cfg.step_graph = 'seg')" if other == "seg" else
                             "the step issued as eager launches (cfg.step_graph = '0')") + "; `value` above is the other form"}
            except Exception as e:  # reported, never hidden
                out["step_graph_segmented" if other == "seg" else "step_eager"] = {"error": repr(e)}
        if not a.no_pipeline and a.backbone == "clip-RN50":
            # BASELINE configs[4]: the ViT-B/16 trunk at B = 48, a few steps in both arithmetics (no reference definition of this
            # model exists: parity unpinned)
            try:
                del model, opt, sched
                torch.cuda.empty_cache()
                vargs, vmodel, vopt, vsched, _ = build("clip-ViT-B/16")
                rec_v = {}
                for m_ in (("h2", "x3") if mode == "h2" else (mode,)):
                    ops.set_gemm_mode(m_)
                    vstep = lambda: train_step(vmodel, aux, vopt, img, ids, neg, vargs, vsched, None)   # noqa: E731
                    vstep()
                    dtv, hv, lv = timed(vstep, 5, 2)
                    rec_v[m_] = {"img_per_s": round(a.batch * 5 / dtv, 2), "ms_per_step": round(dtv / 5 * 1e3, 3), "steps": 5,
                                 "losses_last_step": [round(v, 5) for v in lv.tolist()]}
                ops.set_gemm_mode(mode)
                out["vit_b16"] = dict(rec_v, config="BASELINE configs[4]: clip-ViT-B/16 trunk, 48 img 320x320 (401 tokens), 20-token "
                                                    "queries + 3 negatives, one GPU; 5 timed steps after 1 priming + 2 warm-up steps")
                del vmodel, vopt, vsched
                torch.cuda.empty_cache()
            except Exception as e:  # reported, never hidden
                ops.set_gemm_mode(mode)
                out["vit_b16"] = {"error": repr(e)}
        if not a.no_pipeline and a.backbone == "clip-RN50":
            # the data-parallel code path at ONE rank: real RCCL calls (process group of size 1), SyncBatchNorm over the mailbox
            # transport, the gradient reducer with its sparse embedding exchange -- what the distributed machinery costs per step
            # before any wire is involved.  No scaling is measured here.
            try:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
                cs = ops.place_streams()
                if cs is not None:
                    torch.cuda.set_stream(cs)
                dargs, dmodel, dopt, dsched, dred = build("clip-RN50", distributed=True, force=True)
                dstep = lambda: train_step(dmodel, aux, dopt, img, ids, neg, dargs, dsched, dred)   # noqa: E731
                dstep()
                dtd, hd, ld = timed(dstep, max(5, a.steps // 2), 2)
                from tris_amd import comm
                comm.check_errors()
                n_ = max(5, a.steps // 2)
                out["dist_path_one_rank"] = {
                    "ms_per_step": round(dtd / n_ * 1e3, 3), "img_per_s": round(a.batch * n_ / dtd, 2), "steps": n_,
                    "comm_exposed_ms": round(dred.exposed_ms(), 3), "sparse_embed_exchange": bool(dred.sparse_embed),
                    "sync_bn_transport": "mailbox" if any(m is not None for m in comm.Mailbox._by_group.values()) else "torch.distributed",
                    "losses_last_step": [round(v, 5) for v in ld.tolist()],
                    "note": "world size 1 with the distributed code path forced: its per-step overhead next to `value`; no scaling "
                            "curve has been measured on hardware"}
                comm.shutdown()
                dist.destroy_process_group()
            except Exception as e:  # reported, never hidden
                out["dist_path_one_rank"] = {"error": repr(e)}
        if not a.no_cpu_baseline and a.backbone == "clip-RN50":
            out["cpu_baseline"] = cpu_baseline(tuple(int(x) for x in a.cpu_batches.split(",")))
    if world > 1:
        dist.barrier()
        __import__('tris_amd.comm', fromlist=['x']).shutdown()
        dist.destroy_process_group()
    if rank == 0:
        line = json.dumps(out)
        sys.stderr.flush()
        try:   # RCCL prints a version banner through C stdio (buffered when stdout is a pipe): push it out BEFORE the JSON
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)  # the ONE JSON line, last thing this process writes


if __name__ == "__main__":
    main()
