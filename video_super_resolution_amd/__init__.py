"""video_super_resolution_amd -- MI355X (gfx950) implementation of the per-frame video-SR forward of
PlanNoa/video_super_resolution (`VSR.forward`), behind the reference's own module API.

    from video_super_resolution_amd import VSR          # or: from network.video_super_resolution import VSR

Device code lives in `csrc/` (hand-written HIP, one C-ABI shared library, `include/vsr_hip.h`).
"""
from .vsr import VSR  # noqa: F401
from .sr import SRProjectionModule  # noqa: F401
from .flownet import FlowProjectionModule, FlowNet2  # noqa: F401
from .depth import DepthProjectionModule  # noqa: F401
from .vos import VOSProjectionModule  # noqa: F401
from .ops import Resample2d, ChannelNorm, Correlation  # noqa: F401

__all__ = ["VSR", "SRProjectionModule", "FlowProjectionModule", "FlowNet2", "DepthProjectionModule",
           "VOSProjectionModule", "Resample2d", "ChannelNorm", "Correlation"]
