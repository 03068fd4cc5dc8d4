import os

os.environ.setdefault("MIOPEN_FIND_MODE", "2")  # must precede `import torch` (read at library load)
# the fast mode's AI ("TunaNet") solver predictor aborted the process twice in ~30 runs of the fp32 configuration
# (abort() inside torch conv -> MIOpen, no message); the plain heuristic fallback has not
os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0")
os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "xcheck: runs against libvsr_hip_xcheck.so (superseded builds, switches, stamped diagnostics)")


# What only the cross-check library (include/vsr_hip_xcheck.h) exports.  A test whose body mentions one of these runs with
# `_lib.load()` returning libvsr_hip_xcheck.so, so that a switch it sets reaches the kernels the package's classes launch; every
# other test runs against the shipping library libvsr_hip.so (which has none of these symbols: a stray use fails loudly).
_XCHECK_WORDS = ("vsr_conv2d_tuning", "_variant(", "stamp_buffer", "vsr_sr_utd2_f16", "vsr_sr_tail_f16", "vsr_sr_tail_dec_f16",
                 "vsr_sr_fc_planes_f32")


@pytest.fixture(autouse=True)
def _cross_check_library(request):
    import inspect
    fn = getattr(request.node, "function", None)
    want = request.node.get_closest_marker("xcheck") is not None
    if not want and fn is not None:
        try:
            src = inspect.getsource(fn)
        except (OSError, TypeError):
            src = ""
        want = any(w in src for w in _XCHECK_WORDS)
    if not want:
        yield
        return
    from video_super_resolution_amd import _lib
    with _lib.xcheck():
        yield


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def cpu_vsr():
    """The product module tree built on the CPU with the seeded synthetic weights (parameters only:
    its forward needs the GPU).  Its state_dict doubles as the oracle's parameter dictionary."""
    import torch
    from video_super_resolution_amd import VSR
    from video_super_resolution_amd.weights import fill_module_
    torch.manual_seed(0)
    m = VSR().eval()
    fill_module_(m, seed=0)
    return m


@pytest.fixture(scope="session")
def oracle_params(cpu_vsr):
    return {k: v.detach() for k, v in cpu_vsr.state_dict().items()}


@pytest.fixture(scope="session")
def gpu_vsr(cpu_vsr):
    import copy
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    m = copy.deepcopy(cpu_vsr).cuda().eval()
    m.precision = m.model.precision = "fp32"  # the exact configuration: strict parity bars
    return m


@pytest.fixture(scope="session")
def gpu_vsr_f16(cpu_vsr):
    """Same weights, SR stack on the fp16/MFMA throughput path (the headline configuration)."""
    import copy
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    m = copy.deepcopy(cpu_vsr).cuda().eval()
    m.precision = m.model.precision = "fp16"
    return m
