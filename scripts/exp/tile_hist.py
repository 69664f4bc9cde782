"""Development helper: distribution of the tile-list lengths of the bench scenes.  GPU box: N=2000000 python scripts/exp/tile_hist.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from activesplat_amd import GaussianRasterizer, setup_camera  # noqa: E402
from activesplat_amd import rasterizer as R  # noqa: E402
from activesplat_amd import synthetic as syn  # noqa: E402
from tests import util  # noqa: E402

dev = torch.device("cuda")
W, H = 640, 480
for N in (500_000, 1_000_000, 2_000_000, 2_500_000, 3_000_000):
    cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)._replace(debug=True)
    rv = {k: v.to(dev) for k, v in syn.activate(syn.make_params(N, W, H, seed=0)).items()}
    with torch.no_grad():
        GaussianRasterizer(raster_settings=cam)(means2D=torch.zeros(N, 3, device=dev), **rv)
    art = util.artefacts()
    rg = art["ranges"]
    n = (rg[:, 1] - rg[:, 0]).astype(np.int64)
    nc = art["n_contrib"].reshape(H, W)
    print(N, "tiles", len(n), "D", int(n.sum()), "min/mean/max", int(n.min()), int(n.mean()), int(n.max()), "pct<=4096: %.3f <=4608: %.3f <=5120: %.3f" % ((n <= 4096).mean(), (n <= 4608).mean(), (n <= 5120).mean()),
          "mean n_contrib %.1f max %d" % (nc.mean(), nc.max()), flush=True)
