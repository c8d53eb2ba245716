"""Benchmark of the TRIS Stage-1 training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full Stage-1 training iteration (TRIS forward, CLIP-guided fg / negative-sample / cls losses
through the frozen aux ViT-B/32, backward, [gradient all-reduce], AdamW, LR schedule) on 48 synthetic 320x320
images + 20-token sentences + 3 negatives per image PER GPU (weak scaling), fp32, seed-filled weights.
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA; the x3 mode spends 6 bf16 MFMAs per fp32-accurate product
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
    ap.add_argument("--no-pipeline", action="store_true", help="skip the input-pipeline measurement (SURVEY.md 8f-1)")
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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    import torch.distributed as dist
    # TRIS_FORCE_DIST=1 exercises the RCCL code path (SyncBN collectives + gradient all-reduce) with a single rank
    force = os.environ.get("TRIS_FORCE_DIST") == "1"
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if world > 1 or force:
        # which stream runs on which hardware queue, with the collective backend's own stream in the picture (ops.place_streams)
        from tris_amd import ops as _ops
        cs = _ops.place_streams()
        if cs is not None:
            torch.cuda.set_stream(cs)
    _dummies = []
    for _ in range(int(os.environ.get("TRIS_DBG_DUMMY_STREAMS", "0"))):   # (developer knob: shift the stream -> hardware-queue mapping)
        st_ = torch.cuda.Stream()
        with torch.cuda.stream(st_):
            torch.zeros(1, device="cuda").add_(1)
        _dummies.append(st_)
    torch.cuda.synchronize()

    from tris_amd import ops
    from tris_amd.args import get_parser
    from tris_amd.CLIP import clip
    from tris_amd.model.model_stage1 import TRIS
    from tris_amd.optim import FusedAdamW
    from tris_amd.parallel import attach_reducer, convert_sync_batchnorm
    from tris_amd.train_stage1 import freeze_aux, train_step
    from tris_amd.utils.synth import seed_fill, synthetic_batch

    QL = int(os.environ.get("TRIS_BENCH_QUERY_LEN", "20"))   # developer knob (what-if runs); the metric configuration is 20
    args = get_parser().parse_args(["--backbone", a.backbone, "--size", "320", "--max_query_len", str(QL),
                                    "--negative_samples", "3", "--batch_size", str(a.batch), "--epoch", "15"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = TRIS(args).cuda().train()
        aux, _ = clip.load("ViT-B-32", device="cuda", txt_length=args.max_query_len)
    seed_fill(model.state_dict(), 1234)
    seed_fill(aux.state_dict(), 4321)
    freeze_aux(aux)
    bb, new = model.trainable_parameters()
    opt = FusedAdamW([{"params": bb, "lr": args.lr * args.lr_multi}, {"params": new}], lr=args.lr,
                     weight_decay=args.weight_decay)
    max_iter = 1000 * args.epoch
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda x: (1 - x / max_iter) ** 0.9)
    reducer = None
    if world > 1 or force:
        if os.environ.get("TRIS_DBG_NO_SYNCBN") != "1":       # (developer A/B knobs: which half of the exchange costs what)
            convert_sync_batchnorm(model)
        if os.environ.get("TRIS_DBG_NO_REDUCER") != "1":
            reducer = attach_reducer(model, opt, force=force)   # all-reduce segments launched from inside backward
    b = synthetic_batch(a.batch, 320, QL, 3, seed=7, rank=rank)
    img, ids, neg = b["img"].cuda(), b["word_ids"].cuda(), b["neg_word_ids"].cuda()

    def step():
        return train_step(model, aux, opt, img, ids, neg, args, sched, reducer)

    step()   # one untimed priming step, always: first-encounter GEMM autotuning, allocator growth, RCCL channel set-up
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    host_issue = time.perf_counter() - t0    # the host's share: all K steps issued (it may have waited on a full queue)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_vals = losses.tolist()
    comm_exposed = reducer.exposed_ms() if reducer is not None else None   # compute-stream wait for collectives, last timed step
    transport = None
    if world > 1 or force:
        from tris_amd import comm
        comm.check_errors()    # a SyncBatchNorm mailbox exchange that timed out must fail the run, not skew it
        transport = "mailbox" if any(m is not None for m in comm.Mailbox._by_group.values()) else "torch.distributed"

    # ---- live roofline measurement of the dominant kernel family (one extra, untimed, instrumented step) ----
    # (kernels are timed one at a time: the stream overlap of the production step is switched off for this pass so that a
    # launch's HIP-event bracket measures that kernel alone, not whatever else shares the GPU with it)
    saved_env = {k: os.environ.get(k) for k in ("TRIS_TEXT_STREAM", "TRIS_WGRAD_STREAM")}
    os.environ["TRIS_TEXT_STREAM"] = os.environ["TRIS_WGRAD_STREAM"] = "0"
    step()
    ops.profile_begin()
    step()
    rec_all = ops.profile_end()
    for k, v in saved_env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    xa = [r for r in rec_all if r[0].startswith("xattn")]
    rec = [r for r in rec_all if not r[0].startswith("xattn")]
    fl = sum(r[1] for r in rec)
    ms = sum(r[2] for r in rec)
    kinds = {}
    for k, f, m in rec:
        e = kinds.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += f
        e[2] += m
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = (time.perf_counter() - t1) * 1e3
    ach = fl / (ms * 1e-3) / 1e12
    mode = ops.get_gemm_mode()
    peak = F32_MFMA_PEAK_TFLOPS if mode == "f32" else BF16_MFMA_PEAK_TFLOPS / (6.0 if mode == "x3" else 3.0)
    roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": None,
            # the same achieved rate against the f32-INPUT MFMA peak (what an fp32 product costs without the bf16 split)
            "peak_f32_mfma": F32_MFMA_PEAK_TFLOPS, "frac_vs_f32_mfma": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
            "arithmetic": ("v_mfma_f32_32x32x2_f32 (f32 in)" if mode == "f32" else
                           "split-bf16 x3: 6 x v_mfma_f32_32x32x16_bf16 per fp32-accurate product; achieved/peak are in "
                           "fp32-equivalent FLOPs (peak = 2500 TFLOP/s bf16 dense / 6; the f32-input MFMA peak is 157.3)"
                           if mode == "x3" else
                           "split-bf16 x2 (opt-in throughput mode): 3 x v_mfma_f32_32x32x16_bf16 per product, 16-bit significands; "
                           "peak = 2500 / 3"),
            "kernel": "gemm_fast_kernel<BM,BN,A,B,EPI,PREC> (MFMA GEMM / implicit-GEMM conv family)",
            "launches_per_step": len(rec), "kernel_ms_per_step": round(ms, 3),
            "fused_epilogue_note": ("kinds *_bnbwd are data-gradient products whose epilogue also does the reduction pass of the "
                                    "BatchNorm backward that consumes them (two more activation-sized streams per launch, no FLOPs "
                                    "counted for it): they lower this family's rate by ~3 % and remove 29 reduction launches "
                                    "(-1.0 ms per step, DESIGN.md section 3 'Round 3')"),
            "algorithmic_gflop_per_step": round(fl / 1e9, 1),
            "by_kind": {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2], 3),
                            "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 else None}
                        for k, v in kinds.items()}}

    # PMC-derived HBM traffic of the same kernel family (separate rocprofv3 --pmc passes, committed under profiles/)
    try:
        pmc_file = "r3_pmc_hbm_traffic.json" if mode == "x3" else "r1c_pmc_hbm_traffic.json"
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["gemm_family"]
        roof["traffic"] = int((pmc["fetch_bytes_per_step"] + pmc["write_bytes_per_step"]) / pmc["launches_per_step"])
        roof["traffic_note"] = ("bytes per launch, averaged over the family: (FETCH_SIZE x2 + WRITE_SIZE) per step / launches per "
                                f"step from profiles/{pmc_file} (rocprofv3 --pmc, B=48 step)")
    except Exception:
        pass
    try:   # matrix-pipe utilisation of the family from the SQ counters (separate --pmc pass, committed under profiles/)
        sq = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_mfma_util.json")))
        if mode == "x3":
            roof["mfma_utilisation_pmc"] = sq["families"]["gemm_family"]["mfma_utilisation"]
            roof["mfma_utilisation_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) over the family, "
                                             "profiles/r3_pmc_mfma_util.json (rocprofv3 --pmc, kernels serialised, static tile choice)")
    except Exception:
        pass
    roof_x = None
    if xa:
        P_, N_, C_ = 100, a.batch, 1024
        xbytes = a.batch * (4 * P_ * C_ + N_ * C_) * 4 + 3 * N_ * C_ * 4
        fused = [r for r in xa if r[0] == "xattn_fwd_fused"]
        if fused and len(fused) == len(xa):
            xname = ("xattn_fused_kernel<NT,KS> (ONE persistent launch: 8 channel-slice workgroups per sample, write-through "
                     "reduce-scatter/all-gather of the logits between them) + xattn_text_planes_kernel (sentence bf16 planes)")
        elif mode in ("x3", "x2"):
            xname = "xattn_scores_x3_kernel + xattn_out_x3_kernel"
        else:
            xname = "xattn_scores_kernel + xattn_colsoftmax_kernel + xattn_out_kernel"
        xms = sum(r[2] for r in xa)
        roof_x = {"bound": "hbm", "achieved": round(xbytes / (xms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(xbytes / (xms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                  "kernel": xname + " (fused bilateral cross attention, forward, all launches, HIP events around the call)",
                  "single_launch": bool(fused and len(fused) == len(xa)),
                  "bound_note": ("B x 8 = 384 workgroups on 256 CUs: the doubly-loaded CUs fetch 2 x 1.6 MB at ~24 GB/s per CU "
                                 "(~19 us floor before the exchange); measured timeline per workgroup 31 us, 42-45 us per call "
                                 "(csrc/xattn_fused.hip header, DESIGN.md section 3 'Round 3'): 0.60 of 8 TB/s is not reachable at this size"),
                  "algorithmic_bytes_per_launch_pair": xbytes, "us": round(xms * 1e3, 1),
                  "mfma_tflops": round(sum(r[1] for r in xa) / (xms * 1e-3) / 1e12, 2)}

    if roof_x is not None and mode == "x3":
        try:   # HBM bytes of the cross-attention launches from the same --pmc passes (profiles/r3_pmc_hbm_traffic.csv)
            import csv as _csv
            tr = 0
            for r in _csv.DictReader(open(os.path.join(ROOT, "profiles", "r3_pmc_hbm_traffic.csv"))):
                if "xattn_" in r["Kernel"]:
                    tr += int(r["FetchBytesPerStep(x2 corrected)"]) + int(r["WriteBytesPerStep"])
            if tr:
                roof_x["traffic"] = tr
                roof_x["traffic_note"] = ("FETCH_SIZE x2 + WRITE_SIZE of xattn_fused_kernel + xattn_text_planes_kernel, one forward "
                                          f"(rocprofv3 --pmc, B=48): {tr / xbytes:.2f}x the algorithmic bytes (partial-logit exchange "
                                          "through the workspace, sentence planes)")
        except Exception:
            pass
    if rank == 0:
        out = {"metric": f"Stage-1 training images/sec @320px bs48 (TRIS {a.backbone}, 3 negatives)",
               "value": round(world * a.batch * a.steps / dt, 2), "unit": "img/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "gemm_mode": mode, "data": "synthetic",
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
               "streams": {"text_encoders_on_side_stream": os.environ.get("TRIS_TEXT_STREAM", "1") != "0",
                           "weight_gradients_on_side_stream": os.environ.get("TRIS_WGRAD_STREAM", "1") != "0"},
               "roofline": roof, "roofline_xattn": roof_x}
        if world == 1 and not a.no_pipeline:
            # input pipeline ahead of the step (HBM-resident uint8 dataset -> batch), measured apart from `value`
            try:
                from tools.pipeline_bench import measure as pipeline_measure
                out["input_pipeline"] = pipeline_measure(batch=a.batch)
            except Exception as e:  # reported, never hidden
                out["input_pipeline"] = {"error": repr(e)}
        if world == 1 and not a.no_pipeline and a.backbone == "clip-RN50":
            # evaluation throughput of configs[1] (validate.py's loop): one ref at a time vs batched, identical metrics
            try:
                from tools.eval_throughput import measure as eval_measure
                out["eval"] = eval_measure()
                out["eval_refs_per_s"] = out["eval"]["eval_refs_per_s"]
            except Exception as e:  # reported, never hidden
                out["eval"] = {"error": repr(e)}
        if world == 1 and not a.no_pipeline and reducer is None and os.environ.get("TRIS_STEP_GRAPH", "0") == "0":
            # the same step replayed from the chain of single-stream hipGraphs (TRIS_STEP_GRAPH=seg, tris_amd.graphs.
            # SegmentedTrainStep; bit-identical results, tests/test_gpu_step_graph.py), timed like `value`: what the host's
            # share of a step becomes, and what it costs / gains in step time on THIS box
            try:
                os.environ["TRIS_STEP_GRAPH"] = "seg"
                del losses
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                hi = time.perf_counter() - t0
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t0
                out["step_graph_segmented"] = {"ms_per_step": round(dtg / a.steps * 1e3, 3),
                                               "host_issue_ms_per_step": round(hi / a.steps * 1e3, 3),
                                               "img_per_s": round(a.batch * a.steps / dtg, 2), "steps": a.steps,
                                               "note": "opt-in (TRIS_STEP_GRAPH=seg); `value` above is the eager three-stream step"}
            except Exception as e:  # reported, never hidden
                out["step_graph_segmented"] = {"error": repr(e)}
            finally:
                os.environ["TRIS_STEP_GRAPH"] = "0"
        if world == 1 and not a.no_pipeline and reducer is None and os.environ.get("TRIS_STEP_GRAPH", "0") == "0" \
                and os.environ.get("TRIS_LINEAR_MODE", "") == "" and mode == "x3":
            # opt-in arithmetic for the Linear / 1x1 products (TRIS_LINEAR_MODE=h2: two fp16 pieces per operand, three f16 MFMAs
            # per product instead of six bf16 ones, power-of-two operand scales from device-side amaxes; results on the fp32 noise
            # floor -- tests/test_gpu_h2.py, profiles/r3_h2_study.txt), timed like `value` on the same model.  NOT `value`.
            try:
                os.environ["TRIS_LINEAR_MODE"] = "h2"
                for _ in range(3):
                    lh = step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    lh = step()
                hi = time.perf_counter() - t0
                torch.cuda.synchronize()
                dth = time.perf_counter() - t0
                out["linear_mode_h2"] = {"ms_per_step": round(dth / a.steps * 1e3, 3), "img_per_s": round(a.batch * a.steps / dth, 2),
                                         "host_issue_ms_per_step": round(hi / a.steps * 1e3, 3), "steps": a.steps,
                                         "losses_last_step": [round(v, 5) for v in lh.tolist()],
                                         "note": "opt-in (TRIS_LINEAR_MODE=h2); `value` above is all-x3"}
            except Exception as e:  # reported, never hidden
                out["linear_mode_h2"] = {"error": repr(e)}
            finally:
                os.environ.pop("TRIS_LINEAR_MODE", None)
        if world == 1 and not a.no_cpu_baseline and a.backbone == "clip-RN50":
            out["cpu_baseline"] = cpu_baseline(tuple(int(x) for x in a.cpu_batches.split(",")))
        line = json.dumps(out)
    if world > 1 or force:
        dist.barrier()
        __import__('tris_amd.comm', fromlist=['x']).shutdown()
        dist.destroy_process_group()
    if rank == 0:
        sys.stderr.flush()
        try:   # RCCL prints a version banner through C stdio (buffered when stdout is a pipe): push it out BEFORE the JSON
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)  # the ONE JSON line, last thing this process writes


if __name__ == "__main__":
    main()
