"""Run-time configuration of the package.

Every knob is read from the environment ONCE, when this module is imported, and lives on the object `cfg` as a plain
attribute.  Product code reads `cfg.<name>` -- nothing reads `os.environ` per call.  Tests and tools change a value by
assigning to it (`cfg.step_graph = "seg"`, `monkeypatch.setattr(cfg, "bn_pool", False)`) or for a scope with
`cfg.override(step_graph="seg")`.  Options of the kernel library itself (tile / kernel forcing, tune log) go through
`tris_amd.ops.set_option` -> `tris_set_option` (include/tris_hip.h), which follows the same rule on the C side.

  attribute          environment variable      meaning
  step_graph         TRIS_STEP_GRAPH           training step: "seg" = chain of single-stream hipGraphs, "1" = one hipGraph, "0" = eager
  hipgraph           TRIS_HIPGRAPH             evaluation forwards and the frozen aux text tower replayed from hipGraphs
  text_stream        TRIS_TEXT_STREAM          text encoders on a side stream
  wgrad_stream       TRIS_WGRAD_STREAM         weight gradients on a side stream
  stream_probe       TRIS_STREAM_PROBE         probe side streams onto their own hardware queues (ops.place_streams)
  stream_probe_log   TRIS_STREAM_PROBE_LOG     print the queue classes found
  text_at            TRIS_TEXT_AT              where TRIS.forward issues its text encoder: "layer4" | "layer2" | "start"
  bn_bwd_fuse        TRIS_BN_BWD_FUSE          BatchNorm-backward reductions in the consuming product's epilogue
  bn_pool            TRIS_BN_POOL              BatchNorm + ReLU + AvgPool2d(2) as one op
  mlp_fuse           TRIS_MLP_FUSE             QuickGELU forward / backward in the epilogues of the transformer MLP's two products
  grad_box           TRIS_GRAD_BOX             residual-branch gradients handed to the consuming product's epilogue
  mha                TRIS_MHA                  attention kernel: "auto" | "valu" | "mfma"
  mha_h2             TRIS_MHA_H2               flash-style attention on the 16-bit MFMA (two fp16 pieces) inside an h2 step; 0: the f32 MFMA kernel
  xattn_fused        TRIS_XATTN_FUSED          cross attention as one persistent launch where it applies
  xattn_px           TRIS_XATTN_PX             ... cut by pixel rows (csrc/xattn_px.hip); 0: the channel-slice form (csrc/xattn_fused.hip)
  xattn_bwd_px       TRIS_XATTN_BWD_PX         backward of the pair as one persistent launch cut by pixel rows; 0: the chain of batched products
  xattn_h2           TRIS_XATTN_H2             pixel-row launch in the h2 arithmetic inside an h2 step (two fp16 pieces); 0: always split-bf16
  hbm_loader         TRIS_HBM_LOADER           HBM-resident input pipeline (0: the reference's DataLoader)
  eval_group         TRIS_EVAL_GROUP           refs per batched evaluation group
  mbox_spin          TRIS_MBOX_SPIN            bound of the SyncBatchNorm mailbox spin (polls)
  syncbn_comm        TRIS_SYNCBN_COMM          "mailbox" | "c10d"
  ddp_check          TRIS_DDP_CHECK            NaN-poison check of the gradient reducer's release order
  syncbn_bound       TRIS_SYNCBN_BOUND         SyncBatchNorm mailbox exchanges also leave the bound word of the plane tensor written next (0: separate launches)
  ddp_seg_opt        TRIS_DDP_SEG_OPT          replayed data-parallel step: AdamW per reducer segment right behind its all-reduce (0: one AdamW behind the join)
  ddp_seg_poison     TRIS_DDP_SEG_POISON       checking mode of the above: a segment's parameters are NaN from its early AdamW until the end of the step (then the updated
                                               values are put back): any launch that still reads them shows up as NaN gradients / losses
  ddp_sparse_embed   TRIS_DDP_SPARSE_EMBED     token-embedding gradient as a sparse (ids, rows) exchange
  random_init        TRIS_RANDOM_INIT          clip.load may build an architecture without a weights file
  own_stream         TRIS_OWN_STREAM           the trainer / bench compute on a non-default stream (the default stream serialises with hipGraphs elsewhere)
  h2_planes          TRIS_H2_PLANES            h2 arithmetic: the RN50 trunk's activations / gradients / weights travel as fp16 operand planes
  aux_text_early     TRIS_AUX_TEXT_EARLY       replayed step: the frozen aux text tower right behind the TRIS text encoder (under the trunk), 0: behind the TRIS forward
  vit_token0         TRIS_VIT_TOKEN0           ViT image tower read at its class token: the last block's out_proj / MLP on that row only (0: all 50 rows)
  gemm_convert       TRIS_GEMM_CONVERT         h2 planes: products of a ViT tower with at least this many rows convert their fp32 A operand to planes first (0 = default: never -- measured slower in both configurations, profiles/r6_gemm_convert_ab.txt)
  text_pack          TRIS_TEXT_PACK            no-gradient text passes (the frozen aux tower) on packed rows: positions behind EOT are not computed
  bn_bitmask         TRIS_BN_BITMASK           operand planes: relu(bn(x) + identity) also leaves its ReLU mask as one byte per 8 channels; the consumer's fused
                                               BatchNorm-backward epilogue reads that instead of the plane element
  step_graph_rerecord TRIS_STEP_GRAPH_RERECORD replayed step: a batch shape other than the recorded one, seen this many steps IN A ROW, is recorded in its place
                                               (default 3: a recording made on a ragged first batch does not leave the run eager; 0: never)
  side_param_grads   TRIS_SIDE_PARAM_GRADS     bias / InstanceNorm-parameter gradients (column sums nothing in the backward waits for) on the weight-gradient stream
  fuse_splitk        TRIS_FUSE_SPLITK_PY       split-K products armed with a ticket array: the last block of a tile sums the slabs in the product's own launch (default 0: measured slower)
"""
import contextlib
import os


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None or v == "" else v != "0"


class _Config:
    def __setattr__(self, name, value):
        # (the trainer may pick its own default for step_graph only while nobody -- environment or code -- has chosen one)
        if name == "step_graph" and "step_graph" in self.__dict__:
            self.__dict__["step_graph_chosen"] = True
        self.__dict__[name] = value

    def __init__(self):
        e = os.environ.get
        self.step_graph = e("TRIS_STEP_GRAPH", "0")
        self.step_graph_chosen = "TRIS_STEP_GRAPH" in os.environ
        self.hipgraph = _flag("TRIS_HIPGRAPH", True)
        self.text_stream = _flag("TRIS_TEXT_STREAM", True)
        self.wgrad_stream = _flag("TRIS_WGRAD_STREAM", True)
        self.stream_probe = _flag("TRIS_STREAM_PROBE", True)
        self.stream_probe_log = e("TRIS_STREAM_PROBE_LOG") == "1"
        self.text_at = e("TRIS_TEXT_AT", "layer4")
        self.bn_bwd_fuse = _flag("TRIS_BN_BWD_FUSE", True)
        self.bn_pool = _flag("TRIS_BN_POOL", True)
        self.grad_box = _flag("TRIS_GRAD_BOX", True)
        self.mlp_fuse = _flag("TRIS_MLP_FUSE", True)
        self.mha = e("TRIS_MHA", "auto")
        self.mha_h2 = _flag("TRIS_MHA_H2", True)
        self.ddp_seg_opt = _flag("TRIS_DDP_SEG_OPT", True)
        self.ddp_seg_poison = _flag("TRIS_DDP_SEG_POISON", False)
        self.syncbn_bound = _flag("TRIS_SYNCBN_BOUND", True)
        self.xattn_fused = _flag("TRIS_XATTN_FUSED", True)
        self.xattn_px = _flag("TRIS_XATTN_PX", True)
        self.xattn_h2 = _flag("TRIS_XATTN_H2", True)
        self.xattn_bwd_px = _flag("TRIS_XATTN_BWD_PX", True)
        self.hbm_loader = _flag("TRIS_HBM_LOADER", True)
        self.eval_group = max(1, int(e("TRIS_EVAL_GROUP", "16")))
        self.mbox_spin = int(e("TRIS_MBOX_SPIN", "40000000"))
        self.syncbn_comm = e("TRIS_SYNCBN_COMM", "mailbox")
        self.ddp_check = e("TRIS_DDP_CHECK") == "1"
        self.ddp_sparse_embed = _flag("TRIS_DDP_SPARSE_EMBED", True)
        self.random_init = e("TRIS_RANDOM_INIT") == "1"
        self.h2_planes = _flag("TRIS_H2_PLANES", True)
        self.own_stream = _flag("TRIS_OWN_STREAM", True)
        self.fuse_splitk = _flag("TRIS_FUSE_SPLITK_PY", False)
        self.aux_text_early = _flag("TRIS_AUX_TEXT_EARLY", True)
        self.text_pack = _flag("TRIS_TEXT_PACK", True)
        self.bn_bitmask = _flag("TRIS_BN_BITMASK", True)
        self.gemm_convert = int(e("TRIS_GEMM_CONVERT", "0"))
        self.step_graph_rerecord = int(e("TRIS_STEP_GRAPH_RERECORD", "3"))
        self.side_param_grads = _flag("TRIS_SIDE_PARAM_GRADS", True)
        self.vit_token0 = _flag("TRIS_VIT_TOKEN0", True)

    @contextlib.contextmanager
    def override(self, **kw):
        old = {k: getattr(self, k) for k in kw}   # (AttributeError for an unknown name: no silent typos)
        try:
            for k, v in kw.items():
                setattr(self, k, v)
            yield self
        finally:
            for k, v in old.items():
                setattr(self, k, v)

    def key(self):
        """what a captured step depends on: tris_amd.train_stage1.train_step records the step again when this (or the arithmetic, the
        optimiser, the reducer, the aux model) changes; a batch of another shape runs eagerly next to the kept recording"""
        return tuple(sorted((k, v) for k, v in self.__dict__.items() if k not in ("step_graph_chosen", "step_graph_rerecord")))


cfg = _Config()
