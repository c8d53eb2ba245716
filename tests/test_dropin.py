"""The drop-in under the reference's own import paths (callers: /root/reference/train_stage1.py:12-31,
validate.py:12-24): after `tris_amd.dropin.install()` the reference's import statements resolve to the MI355X path, and
every caller-facing symbol takes the same parameters as the reference's (compared against the live reference when
/root/reference is present; importability and the committed expectations are checked everywhere)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

PROBE = os.path.join(ROOT, "tests", "_dropin_probe.py")


def _probe(which):
    r = subprocess.run([sys.executable, "-B", PROBE, which], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")][-1]
    return json.loads(line[6:])


def test_reference_import_paths_resolve_to_the_mi355x_path():
    got = _probe("dropin")
    assert got["TRIS.module"] == "tris_amd.model.model_stage1"
    assert got["TRIS.forward"] == ["x", "word_id"] and got["TRIS.__init__"] == ["args"]
    assert got["bilateral_prompt.__init__"] == ["vis_chans", "lan_chans", "m_chans"]
    assert got["clip.load"][:5] == ["name", "device", "jit", "download_root", "txt_length"]
    assert got["validate.validate"] == ["args", "data_loader", "model", "local_rank", "visualize", "logger", "save_cam"]
    assert got["train_stage1.train_one_epoch"][:9] == ["train_loader", "model", "optimizer", "epoch", "local_rank", "args",
                                                       "iteration", "clip_model", "lr_scheduler"]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="live reference only exists in the build container")
def test_signatures_match_the_live_reference():
    ref, got = _probe("reference"), _probe("dropin")
    for k, want in ref.items():
        if k in ("TRIS.module", "clip.has_tokenizer"):
            continue
        if k == "parser_dests":
            assert set(want) <= set(got[k]), sorted(set(want) - set(got[k]))
            continue
        have = got[k]
        # same leading parameters, same order; the drop-in may append optional ones (reducer=, logger=, tokenizer=...)
        assert have[:len(want)] == want, (k, want, have)


def test_install_refuses_to_shadow_and_uninstall_restores():
    import types
    import tris_amd.dropin as d
    d.uninstall()
    # other tests of this session may have imported the live reference under its own names (oracle.ref_shim): park those
    tops = {n.split(".")[0] for n in d.ALIASES}
    parked = {n: sys.modules.pop(n) for n in list(sys.modules) if n.split(".")[0] in tops}
    try:
        _refuse_and_restore(d, types)
    finally:
        sys.modules.update(parked)


def _refuse_and_restore(d, types):
    sentinel = types.ModuleType("args")
    sys.modules["args"] = sentinel
    try:
        with pytest.raises(ImportError):
            d.install()
        assert sys.modules["args"] is sentinel
        d.install(force=True)
        assert sys.modules["args"].__name__ == "tris_amd.args"
        d.uninstall()
        assert sys.modules["args"] is sentinel
    finally:
        d.uninstall()
        sys.modules.pop("args", None)
    d.install()
    try:
        from model.model_stage1 import TRIS
        assert TRIS.__module__ == "tris_amd.model.model_stage1"
    finally:
        d.uninstall()
    assert "model.model_stage1" not in sys.modules


def test_reference_syncbn_call_reaches_the_tris_batchnorm():
    """/root/reference/train_stage1.py:69-72 under the drop-in: `nn.SyncBatchNorm.convert_sync_batchnorm(model)` must not be the
    silent no-op it is on a module torch does not know -- it sets `process_group` on every tris_amd BatchNorm2d (and converts
    plain nn.BatchNorm2d children as torch always did), raises without a process group, and uninstall() restores torch's own."""
    import torch
    import torch.distributed as dist
    from torch import nn
    import tris_amd.dropin as d
    from tris_amd.CLIP.clip.model import BatchNorm2d
    orig = nn.SyncBatchNorm.__dict__["convert_sync_batchnorm"]
    tops = {n.split(".")[0] for n in d.ALIASES}
    parked = {n: sys.modules.pop(n) for n in list(sys.modules) if n.split(".")[0] in tops and not n.startswith("tris_amd")}
    d.install()
    try:
        model = nn.Sequential(BatchNorm2d(8), nn.Sequential(BatchNorm2d(4), nn.BatchNorm2d(4)))
        was_init = dist.is_initialized()
        if not was_init:
            with pytest.raises(RuntimeError, match="process group"):
                nn.SyncBatchNorm.convert_sync_batchnorm(model)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29631")
            dist.init_process_group("gloo", rank=0, world_size=1)
        try:
            # the reference's call sequence: model = nn.SyncBatchNorm.convert_sync_batchnorm(model); model.module afterwards
            out = nn.SyncBatchNorm.convert_sync_batchnorm(model)
            ours = [m for m in out.modules() if isinstance(m, BatchNorm2d)]
            assert len(ours) == 2 and all(m.process_group is dist.group.WORLD for m in ours)
            assert isinstance(out[1][1], nn.SyncBatchNorm)          # torch's own conversion still happens
            assert set(out.state_dict()) == set(model.state_dict())  # same keys: checkpoints keep loading
        finally:
            if not was_init:
                dist.destroy_process_group()
    finally:
        d.uninstall()
        sys.modules.update(parked)
    assert nn.SyncBatchNorm.__dict__["convert_sync_batchnorm"] is orig
