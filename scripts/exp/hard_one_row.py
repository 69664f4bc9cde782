"""Development helper (checker run): ONE Gaussian of a hard-sweep scene rendered alone and with growing subsets of the others on the device; its gradient row against fp64.
usage: python scripts/exp/hard_one_row.py 180333 1.2 3429"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.gs_oracle import Oracle
from activesplat_amd import _lib
from tests import util
from tests.fuzz_scenes import sweep_scene
seed, hard, row = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
dev = os.environ.get("DEV", "cuda")
if dev == "cpu":
    util.use_emulated_kernels(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "hipemu", "libgsplat_emu.so"))
o32, o64 = Oracle("f32"), Oracle("f64"); o32.set_threads(8); o64.set_threads(8)
rs, rv = sweep_scene(seed, dev, None)
gen = torch.Generator().manual_seed(seed)
rv["scales"] = rv["scales"] * torch.exp(hard * torch.randn(rv["scales"].shape, generator=gen)).to(rv["scales"].device)
if seed % 2:
    rv["means3D"] = rv["means3D"] * torch.tensor([1.0, 1.0, 0.35], device=rv["means3D"].device)
H, W = int(rs.image_height), int(rs.image_width)
P = rv["means3D"].shape[0]
for name, dLgen in (("random dL", lambda: torch.randn(3, H, W, generator=torch.Generator().manual_seed(0))), ("dL = 1", lambda: torch.ones(3, H, W))):
    dL = dLgen()
    for n_other in (0, 100, 2000, P - 1):
        r = np.random.RandomState(1)
        others = [i for i in r.permutation(P).tolist() if i != row][:n_other]
        idx = torch.tensor([row] + others, device=rv["means3D"].device)
        sub = {k: v[idx].contiguous() for k, v in rv.items()}
        got = util.run_product(rs, sub, dL); r64 = util.run_oracle(o64, rs, sub, dL); r32 = util.run_oracle(o32, rs, sub, dL)
        line = []
        for k in ("means2D", "opacities", "colors_precomp"):
            if k not in got["grads"]: continue
            a, b, c = got["grads"][k][0], np.asarray(r64["grads"][k]).reshape(got["grads"][k].shape)[0], np.asarray(r32["grads"][k]).reshape(got["grads"][k].shape)[0]
            line.append(f"{k} kernel/fp64 {np.round(a / np.where(b == 0, 1, b), 4).tolist()} o32/fp64 {np.round(c / np.where(b == 0, 1, b), 4).tolist()}")
        print(f"{name}, {n_other} others (D {got['D']}): " + " | ".join(line), flush=True)
