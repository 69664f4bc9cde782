import os, sys, traceback
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle.gs_oracle import Oracle
from tests import parity_cases as pc, util
from tests.fuzz_scenes import sweep_scene
seed = int(os.environ["SEED"])
rs, rv = sweep_scene(seed, "cuda", None)
gen = torch.Generator().manual_seed(seed)
rv["scales"] = rv["scales"] * torch.exp(1.2 * torch.randn(rv["scales"].shape, generator=gen)).to(rv["scales"].device)
if seed % 2:
    rv["means3D"] = rv["means3D"] * torch.tensor([1.0, 1.0, 0.35], device=rv["means3D"].device)
o32, o64 = Oracle("f32"), Oracle("f64")
try:
    pc.check_forward(rs, rv, o32, oracle64=o64)
    print("forward ok")
except AssertionError:
    traceback.print_exc(limit=2)
try:
    pc.check_fused_rgbd(rs, rv, o64, seed=seed, oracle32=o32)
    print("rgbd ok")
except AssertionError:
    traceback.print_exc(limit=2)
