"""video_super_resolution_amd -- MI355X (gfx950) implementation of the per-frame video-SR forward of
PlanNoa/video_super_resolution (`VSR.forward`), behind the reference's own module API.

    from video_super_resolution_amd import VSR          # or: from network.video_super_resolution import VSR

Device code lives in `csrc/` (hand-written HIP, one C-ABI shared library, `include/vsr_hip.h`).
"""
import os as _os

# The guidance trunks (FlowNet2 / depth / OSVOS) run on stock MIOpen convolutions.  Without a find-db for gfx950
# MIOpen's default "find" benchmarks every applicable solver per new shape (minutes on a fresh machine);
# mode 2 (FAST) takes the heuristic pick immediately, with the same steady-state speed measured here.
# NOTE: MIOpen reads the variable when it is loaded, so this only helps if this package is imported before
# `torch`; bench.py, tests/conftest.py and __graft_entry__.py therefore also set it before importing torch.
_os.environ.setdefault("MIOPEN_FIND_MODE", "2")
_os.environ.setdefault("MIOPEN_DEBUG_ENABLE_AI_IMMED_MODE_FALLBACK", "0")  # plain heuristic fallback (see bench.py)
_os.environ.setdefault("MIOPEN_LOG_LEVEL", "2")  # errors only: the fallback heuristic is chatty at warning level

from .vsr import VSR  # noqa: F401,E402
from .graph import GraphedVSR  # noqa: F401,E402
from .sr import SRProjectionModule  # noqa: F401,E402
from .flownet import FlowProjectionModule, FlowNet2  # noqa: F401,E402
from .depth import DepthProjectionModule  # noqa: F401,E402
from .vos import VOSProjectionModule  # noqa: F401,E402
from .ops import Resample2d, ChannelNorm, Correlation  # noqa: F401,E402

__all__ = ["VSR", "GraphedVSR", "SRProjectionModule", "FlowProjectionModule", "FlowNet2", "DepthProjectionModule",
           "VOSProjectionModule", "Resample2d", "ChannelNorm", "Correlation"]
