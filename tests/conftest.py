import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Deterministic rounding for the parity suite: the GEMM autotuner picks (tile, split-K) by timing, which changes the
# summation order from run to run; tests that sit on the fp32 noise floor (ReLU-kink flips) need a fixed order.  The
# autotuned configuration is exercised explicitly by tests/test_gpu_ops.py::test_gemm_autotune_*.
os.environ.setdefault("TRIS_AUTOTUNE", "0")
# the suite builds every architecture without a weights file and fills it with the seed-fill protocol
os.environ.setdefault("TRIS_RANDOM_INIT", "1")


def pytest_sessionstart(session):
    """Build infrastructure, not a fallback: if the in-tree library has not been built on this machine yet and hipcc is
    here, build it once (what `__graft_entry__.build()` does) so that the suite tests the product instead of failing on
    a missing artefact.  The product itself never builds or falls back at run time (tris_amd/_lib.py raises)."""
    import subprocess
    lib = os.path.join(ROOT, "tris_amd", "libtris_hip.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(lib) and os.path.exists(hipcc):
        subprocess.call(["bash", os.path.join(ROOT, "tris_amd", "csrc", "build.sh")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _reset_library_options(request):
    """developer options of the kernel library (tris_amd.ops.set_option) do not leak from one GPU test into the next"""
    yield
    if request.node.get_closest_marker("gpu") is not None and "tris_amd.ops" in sys.modules:
        o = sys.modules["tris_amd.ops"]
        for name in ("FORCE_TILE", "FORCE_PIPE", "PIPE", "CONV_DIRECT", "WGRAD_DIRECT", "BN_FOLD", "STEM_CONV1", "WG_BLOCKS", "FUSE_SPLITK"):
            o.set_option(name, None)
