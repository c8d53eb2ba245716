"""Live cross-check against the real reference (only where /root/reference exists, i.e. the build container; skipped on
the GPU box).  Complements the committed golden vectors: the oracle and the drop-in argparse surface are compared with
the imported reference on inputs that are NOT in tests/golden."""
import os

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted on this machine")


def test_argparse_surface_matches_reference():
    import importlib.util
    from tris_amd.args import get_parser
    spec = importlib.util.spec_from_file_location("ref_args", os.path.join(REF, "args.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ours, ref = get_parser().parse_args([]), mod.get_parser().parse_args([])
    assert vars(ours) == vars(ref)
    argv = ["--batch_size", "48", "--size", "320", "--negative_samples", "3", "--weight_decay", "0.02", "--distributed",
            "--print-freq", "5", "--backbone", "clip-RN50"]
    assert vars(get_parser().parse_args(argv)) == vars(mod.get_parser().parse_args(argv))


def test_oracle_matches_live_reference_on_fresh_inputs():
    from oracle import ref_shim
    from oracle import tris_oracle as O
    from tris_amd.utils.synth import seed_fill, synthetic_batch
    torch.manual_seed(0)
    ref = ref_shim.make_tris()
    seed_fill(ref.state_dict(), 777)
    b = synthetic_batch(2, 320, 20, 0, seed=123)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        ref.eval()
        want = ref(b["img"], b["word_ids"])
        got = O.tris_forward(sd, b["img"], b["word_ids"], False)
    assert float((want - got).abs().max()) < 5e-5
    ref.train()
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        w = ref(b["img"], b["word_ids"])
        g = O.tris_forward(sd, b["img"], b["word_ids"], True)
    for a, c in zip(w, g):
        assert float((a - c).abs().max()) < 5e-5
