"""Drop-in module name: the reference imports `GaussianRasterizer` / `GaussianRasterizationSettings`
from `diff_gaussian_rasterization` (src/mapper/splatam/splatam.py:22-23, utils/recon_helpers.py:2,
utils/eval_helpers.py:18).  Putting this repository on PYTHONPATH makes those imports resolve to the
MI355X-native implementation in activesplat_amd."""
from activesplat_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                        rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
