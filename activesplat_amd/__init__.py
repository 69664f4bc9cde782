"""activesplat_amd -- MI355X-native (gfx950) differentiable Gaussian-splatting rasteriser for the
ActiveSplat mapper hot path.

Public surface (mirrors the reference's `diff_gaussian_rasterization` import):
    GaussianRasterizationSettings, GaussianRasterizer      -- activesplat_amd.rasterizer
    setup_camera                                          -- activesplat_amd.camera
The compute path is the hand-written HIP library activesplat_amd/csrc -> libgsplat_hip.so, reached
through a plain C ABI (include/gsplat_hip.h).  There is no CPU fallback in this package.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401
from .camera import setup_camera  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "setup_camera"]
