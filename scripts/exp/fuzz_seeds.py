"""Development helper (uses the oracle: a checker run): the checks of scripts/exp/fuzz_gpu.py on a given list of seeds, one line per seed -- to
compare two builds of the library on the scenes a sweep flagged.  GPU box: SEEDS=122026,122050 RGBD=1 PLAIN=2 python scripts/exp/fuzz_seeds.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gs_oracle import Oracle
from activesplat_amd import _lib
from tests import parity_cases as pc, util
from tests.fuzz_scenes import sweep_scene

o32, o64 = Oracle("f32"), Oracle("f64")
lib = _lib.get()
plain = os.environ.get("PLAIN")
if plain:
    _lib.check(lib.gs_set_half_quadrants(0)); _lib.check(lib.gs_set_backward_chain(3, 0))
for seed in [int(x) for x in os.environ["SEEDS"].split(",")]:
    rs, rv = sweep_scene(seed, "cuda", plain)
    try:
        pc.check_fused_rgbd(rs, rv, o64, seed=seed, oracle32=o32)
        from activesplat_amd import rasterizer as R
        print("seed", seed, "P", rv["means3D"].shape[0], "ok", "max tile list", R.last_stats.get("max_tile_instances"), flush=True)
    except Exception as e:
        from activesplat_amd import rasterizer as R
        print("seed", seed, "P", rv["means3D"].shape[0], "max tile list", R.last_stats.get("max_tile_instances"), "FAIL", repr(e)[:160].replace("\n", " "), flush=True)
