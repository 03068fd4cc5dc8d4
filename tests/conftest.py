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
