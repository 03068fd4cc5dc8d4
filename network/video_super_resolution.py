"""Import-path shim: `from network.video_super_resolution import VSR` (reference main.py:9) resolves to the
gfx950 implementation, so the reference's driver code needs no edit to pick it up."""
from video_super_resolution_amd.vsr import VSR  # noqa: F401
