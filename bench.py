#!/usr/bin/env python
"""bench.py -- headline benchmark of the rasteriser hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: one "step" = one forward (colour + radii + depth + opacity) + one backward (all input gradients incl. means2D) of the LARGEST
single-GPU configuration of BASELINE.json -- configs[2]'s render, the north star's target: 2 M synthetic Gaussians with SH degree 3,
one 640x480 view, inputs and dL/dcolor resident in HBM.  The K steps run strictly one after the other on one
stream (the mapper's one-keyframe-per-Adam-step pattern); `value` = frames/s of that sequential loop.
N > 1: BASELINE.json configs[3] -- 2 M Gaussians replicated on every rank, a batch of 64 keyframes at 640x480 block-partitioned over
the ranks (64/N each); one "step" = one optimiser step over the batch: every rank renders its keyframes (activations inside the render's per-Gaussian kernels ->
single-pass RGB-D render -> fused loss -> backward), the flat [N,14] fp32 gradient is reduce-scattered over RCCL, each rank runs the
fused Adam on its 1/N row block and the updated rows are all-gathered (activesplat_amd/parallel.py).  `value` = keyframes rendered +
back-propagated per second by the whole job (strong scaling: the batch is fixed); rank 0 also times the same 64-keyframe step alone
(`single_gpu_same_workload_fps`; the N = 1 line carries the same figure as `configs3_single_gpu`).
Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only, after the timed region):
  roofline      : per-stage hipEvent timing through the C ABI's gs_profile_* hooks over a second pass of the same K steps; the
                  dominant stage's algorithmic bytes / its average duration against the 8 TB/s HBM peak; the whole
                  frame's algorithmic bytes / the sequential frame time (`frame_frac_sequential`, the north star's figure).
  cpu_baseline  : the C oracle (oracle/gs_oracle.c, OpenMP, up to 64 host threads) on whole frames of the same workload for ~12 s.
  configs0/1    : BASELINE configs[0] (10 k, forward, CPU PyTorch next to the HIP forward) and configs[1] (500 k, SH-0) rates.
  configs2_loop : configs[2]'s 100-iteration optimise loop with fused Adam and one densify event.
"""
import argparse
import glob
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")     # the CPU legs' OpenMP workers must not spin next to torch's pool

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CPU_SECONDS = 12.0           # CPU work the cpu_baseline leg is bounded to (whole frames of the same workload, all host cores)
VALU_ISSUE_PEAK = 890e9      # wave64 plain-fp32 VALU instructions/s of the chip, MEASURED (scripts/exp/valu_issue.hip, profiles/r02_valu_issue.txt:
                             # v_mul/v_add_f32 at 8 waves per SIMD; DPP / v_cndmask / v_cmp / packed fp32 issue at 0.45-0.65x of this)
FP32_PEAK = 157.3e12         # MI355X_MICROARCH.md: fp32 vector peak
XGMI_LINK = 153e9            # MI355X_MICROARCH.md / SURVEY App. B: one xGMI link, one direction; 7 links per GPU, fully connected
XGMI_PEAK = 7 * XGMI_LINK
METRIC = "render+backward frames/sec at 640x480, N Gaussians"


_T0 = time.perf_counter()


_RESULT_FD = None


def claim_stdout():
    """stdout carries ONE JSON line and nothing else: from here on file descriptor 1 points at stderr, so that whatever a library prints there
    (RCCL announces its version on stdout when a process group comes up) cannot end up next to the result; emit() writes to the real stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def note(msg):
    """progress line on stderr (the JSON line on stdout stays the only stdout output)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def stage_bytes(P, D, npix, sh=False):
    """Algorithmic HBM bytes per launch of each stage: the per-unit figures of SURVEY.md section 8d
    (b_g = 292 B/Gaussian -- 832 with 16 SH coefficients --, b_i = 160 B/instance, b_p = 48 B/pixel) split by stage."""
    sb = {
        "preprocess_forward+scan": P * (56 + 48 + 8),          # read inputs 56, write record 48, scan 8
        "tile_count+scan": P * 12,                             # read rect + tile count
        "tile_scatter+sort": P * 20 + D * 12 + D * 24,         # = emit (read 20/Gaussian, write key+value 12) + ideal one-pass sort 24
        "emit": P * 20 + D * 12,                               # (radix path) read rect/tiles/depth 20, write key+value 12
        "sort": D * 24,                                        # (radix path) ideal one-pass: read 12 + write 12
        "ranges": D * 8,
        "blend_forward": D * 40 + npix * 28,                   # record gather 40; write colour 12 + depth 4 + opacity 4 + T 4 + n 4
        "blend_backward": D * (40 + 36) + npix * 20,           # record gather 40 + grad accumulation 36; read dL 12 + T 4 + n 4
        "preprocess_backward": P * (56 + 36 + 68),             # read inputs 56 + reduced 2-D grads 36, write grads 68
    }
    if sh:                                                     # SH-3: 192 B of coefficients read forward; read + 192 B written backward
        sb["preprocess_forward+scan"] += P * 180               # colours (12 B) replaced by coefficient rows (192 B)
        sb["preprocess_backward"] += P * (180 + 180)
    return sb


def frame_bytes(P, D, npix, sh=False):
    return P * (832 if sh else 292) + D * 160 + npix * 48


def c4_roofline(n_gauss, D, npix, sh, world, KF, step_s, exchange_ms, exchange_bytes, single_fps=None):
    """The `roofline` block of a configs[3] line.  Per GPU: the keyframes one rank renders per step x SURVEY 8(d)'s algorithmic bytes of a
    forward + backward frame (N b_g + D b_i + px b_p: the metric's figure -- the loss kernels' and the depth/silhouette channels' extra
    traffic is NOT counted, so the fraction is a lower bound) over the WHOLE step time (render + loss + backward + exchange + Adam) against
    8 TB/s.  The exchange: bytes a rank puts on its links (reduce-scatter + all-gather: 2 S (ranks - 1) / ranks) over the event-timed
    exchange (pack -> reduce-scatter -> Adam on the row block -> all-gather -> unpack) against 7 x 153 GB/s."""
    G = 59 if sh else 14
    B = frame_bytes(n_gauss, D, npix, sh=sh)
    per_gpu = KF * B / max(world, 1) / step_s
    r = {"bound": "hbm", "kernel": "whole keyframe (forward + loss + backward), per GPU", "achieved": round(per_gpu / 1e9, 2), "peak": HBM_PEAK / 1e9,
         "unit": "GB/s", "frac": round(per_gpu / HBM_PEAK, 5), "frame_frac": round(per_gpu / HBM_PEAK, 5), "traffic": None,
         "alg_bytes_per_keyframe": int(B), "tile_instances_D_per_keyframe": int(D), "gaussians": int(n_gauss), "keyframes_per_gpu_per_step": KF / max(world, 1),
         "frame_frac_note": "keyframes x SURVEY 8(d) B_alg(keyframe) / (n_gpus x step time x 8 TB/s); the step time includes loss, exchange and Adam, "
                            "whose own bytes are not in B_alg"}
    if exchange_ms and exchange_bytes and world > 1:
        wire = 2.0 * exchange_bytes * (world - 1) / world
        # pack (gradients in, flat buffer out), unpack (flat in, parameters out): 4 S; Adam on 1/world of the rows: reduced shard in, p / m / v in and out, updated shard out
        hbm = exchange_bytes * 4 + 32.0 * G * n_gauss / world
        r.update(exchange_floats_per_gaussian=G, exchange_buffer_bytes=int(exchange_bytes), exchange_wire_bytes_per_rank=int(wire),
                 exchange_ms=round(exchange_ms, 4), exchange_frac_xgmi=round(wire / (exchange_ms * 1e-3) / XGMI_PEAK, 4), xgmi_peak_gbs=XGMI_PEAK / 1e9,
                 exchange_hbm_alg_bytes=int(hbm), exchange_frac_hbm=round(hbm / (exchange_ms * 1e-3) / HBM_PEAK, 4),
                 exchange_share_of_step=round(exchange_ms * 1e-3 / step_s, 4),
                 exchange_note="wire bytes a rank sends (reduce-scatter + all-gather) / the event-timed exchange INCLUDING pack, the sharded Adam and unpack "
                               "/ (7 links x 153 GB/s); gloo on one device (development knob) moves these bytes through host memory, not xGMI")
    if single_fps:
        r["single_gpu_frame_frac"] = round(single_fps * B / HBM_PEAK, 5)
        r["single_gpu_note"] = "the same batch on rank 0's GPU alone (no collective): keyframes/s x B_alg(keyframe) / 8 TB/s"
    return r


class RenderWorkload:
    """N synthetic Gaussians (SURVEY 8d generator, seed 0), one 640x480 view, forward + backward with a fixed dL/dcolor (seed 1)."""

    def __init__(self, N, W, H, dev, sh_degree=None, w2c=None):
        from activesplat_amd import setup_camera
        from activesplat_amd import synthetic as syn
        self.N, self.W, self.H, self.dev, self.sh = N, W, H, dev, sh_degree
        self.K = syn.intrinsics(W, H)
        self.cam = setup_camera(W, H, self.K, np.eye(4) if w2c is None else w2c, device=dev, sh_degree=sh_degree or 0)
        self.params = syn.make_params(N, W, H, seed=0, sh_degree=sh_degree)
        self.rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(self.params).items()}
        self.keys = list(self.rv.keys())
        self.dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
        # the dummy screen-space tensor the reference's callers pass (its VALUES are never read: only its gradient is produced); like the
        # other inputs it is resident before the timed region -- a fresh torch.zeros(N, 3) per frame is a 24 MB fill at 2 M Gaussians
        self.m2d = torch.zeros(N, 3, device=dev, requires_grad=True)

    def step(self):
        from activesplat_amd import GaussianRasterizer
        color = GaussianRasterizer(raster_settings=self.cam)(means2D=self.m2d, **self.rv)[0]
        # (the backward on the calling thread instead of autograd's device thread: no thread hand-over per frame -- at 2 M Gaussians the frame is GPU-bound
        # either way; at the reference's 256 x 256 frames the hand-over is what made the host the bound in the slow thread placement: INTEGRATION.md 3b)
        with torch.autograd.set_multithreading_enabled(False):
            return torch.autograd.grad(color, [self.rv[k] for k in self.keys] + [self.m2d], self.dL)

    def sequential(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / steps

    def two_streams(self, steps, warmup=4):
        pool = [torch.cuda.Stream(device=self.dev) for _ in range(2)]
        torch.cuda.synchronize()
        for i in range(warmup):
            with torch.cuda.stream(pool[i % 2]):
                self.step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(pool[i % 2]):
                self.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / steps

    def stages(self, lib, steps):
        """per-stage hipEvent averages (events recorded by the library on the caller's stream) + algorithmic bytes"""
        from activesplat_amd import _lib
        from activesplat_amd import rasterizer as R
        lib.gs_profile_enable(1)
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        lib.gs_profile_enable(0)
        D = int(R.last_stats["num_rendered"])
        sb = stage_bytes(self.N, D, self.W * self.H, sh=bool(self.sh))
        st = {k: {"avg_us": round(ms / c * 1e3, 2), "calls": c, "alg_bytes": sb.get(k),
                  "frac_hbm": round(sb[k] / (ms / c * 1e-3) / HBM_PEAK, 4) if k in sb and ms > 0 else None}
              for k, (ms, c) in prof.items() if c}
        return st, sb, D


def self_launch(n, same_dev):
    """Re-execute this command line as n ranks of one node through torch.distributed.run (127.0.0.1 rendezvous on a free port), pass the
    ranks' output through and return their exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not same_dev:
        print(f"bench.py: --gpus {n} but {have} device(s) visible; nothing measured", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if same_dev:
        env.setdefault("BENCH_BACKEND", "gloo")               # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"--gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd[1:9])} ...")
    return subprocess.call(cmd, env=env)


def mapping_iteration_ms(dev, params_cpu, W, H, flags, iters=20, warm=5):
    """One ActiveSplat mapping iteration (get_loss + backward + Adam + zero_grad) on the given map: (wall ms, hipEvent ms) per iteration."""
    from activesplat_amd import mapping as M, optim as O, setup_camera
    from activesplat_amd import synthetic as syn
    n = params_cpu["means3D"].shape[0]
    cam1 = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev)
    prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in params_cpu.items()}
    prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
    prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
    var = {k: torch.zeros(n, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
    tim, tdepth = syn.make_targets(W, H)
    data = dict(cam=cam1, im=tim.to(dev), depth=tdepth.to(dev), id=0, w2c=torch.eye(4, device=dev))
    opt = O.initialize_optimizer(prm, dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05,
                                           log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0))
    flags = dict(flags)
    in_bwd = flags.pop("adam", False)
    direct = flags.pop("direct", False)

    def it():
        if direct:                                             # the iteration's four library calls, no autograd
            M.mapping_iteration(prm, data, var, 0, dict(im=0.5, depth=1.0), opt, pose7=[1.0, 0, 0, 0, 0, 0, 0])
            return
        loss, _, _ = M.get_loss(prm, data, var, 0, dict(im=0.5, depth=1.0), pose7=[1.0, 0, 0, 0, 0, 0, 0] if flags else None,
                                fused_adam=opt if in_bwd else None, **flags)
        with M.backward_on_calling_thread():             # (the backward on the calling thread: 315-390 -> 199-223 us at 256 x 256 / 200 k, profiles/README.md)
            loss.backward(M.unit_gradient(loss) if flags else None)
        with torch.no_grad():
            opt.step(); opt.zero_grad(set_to_none=True)
    for _ in range(warm):
        it()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t1 = time.perf_counter(); a.record()
    for _ in range(iters):
        it()
    b.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t1) / iters * 1e3, a.elapsed_time(b) / iters


def pmc_file(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def profiled_kernels(leg):
    """What rocprofv3 says about a non-headline leg (scripts/gpu_round_end.sh -> scripts/prof_legs.py): the newest profiles/rNN_<leg>_kernel_stats.csv
    (kernel-trace averages) merged with profiles/rNN_<leg>_pmc_summary.json (FETCH_SIZE / WRITE_SIZE from their own passes, FETCH doubled per the
    guide's gfx950 correction; SQ counters) -> {kernel<template args>: {calls, avg_us, pmc_bytes, frac_hbm_pmc, valu_busy}} for the library's and RCCL's
    kernels, or None when this tree holds no such profile.  PMC counters cannot be collected inside this process: the figures are from a separate
    pass of the same workload and say so (`source`)."""
    import csv
    import re
    fk, fp = pmc_file(f"r[0-9][0-9]_{leg}_kernel_stats.csv"), pmc_file(f"r[0-9][0-9]_{leg}_pmc_summary.json")
    if not fk:
        return None
    norm = lambda n: re.sub(r"\s+", "", re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("gs::", ""))  # noqa: E731
    pm = {}
    if fp:
        pm = {norm(k): v for k, v in json.load(open(fp)).items()}
    out = {}
    with open(fk) as fh:
        for row in csv.DictReader(fh):
            name = row["Name"]
            if "gs::" not in name and "nccl" not in name.lower() and "rccl" not in name.lower():
                continue
            k = norm(name)
            us = float(row["AverageNs"]) / 1e3
            e = {"calls": int(row["Calls"]), "avg_us": round(us, 2)}
            c = pm.get(k) or pm.get(re.sub(r"<.*", "", k))
            if c and "traffic_bytes" in c:
                e["pmc_bytes"] = int(c["traffic_bytes"])
                e["frac_hbm_pmc"] = round(c["traffic_bytes"] / (us * 1e-6) / HBM_PEAK, 4)
            if c and "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
                e["valu_busy"] = round(float(c["SQ_ACTIVE_INST_VALU"]) * 4.0 / (1024.0 * float(c["GRBM_GUI_ACTIVE"]) / 8.0), 4)
            out[k] = e
    return {"kernels": out, "source": [os.path.basename(fk)] + ([os.path.basename(fp)] if fp else []),
            "note": "rocprofv3 --kernel-trace --stats averages; pmc_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes) from separate --pmc passes of the same "
                    "workload (scripts/prof_legs.py); frac_hbm_pmc = pmc_bytes / avg duration / 8 TB/s; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles)"}


def counters_for(prof, *patterns):
    """-> (pmc_bytes, avg_us) of the first profiled kernel whose normalised name contains every pattern (spaces removed), else (None, None)"""
    if not prof:
        return None, None
    for k, e in prof["kernels"].items():
        if all(p.replace(" ", "") in k for p in patterns):
            return e.get("pmc_bytes"), e.get("avg_us")
    return None, None


def suite_tally():
    """The tiered gradient rule's tally of the most recent `-m gpu` suite log kept under profiles/ (tests/conftest.py prints it): how many gradient
    tensors were compared with the fp64 oracle, how often the fp32-oracle rule and the decision-matched comparison decided instead of the stated bar."""
    import re
    f = pmc_file("r[0-9][0-9]_pytest_gpu*.log")
    if not f:
        return None
    m = re.search(r"check_backward: (\d+) gradient tensors .*? fp32 escape hatch fired (\d+) time\(s\)(?:.*?decision-(?:aware rule|matched comparison) .*? decided (\d+) time\(s\))?", open(f).read(), re.S)
    if not m:
        return None
    n, fired, dec = int(m.group(1)), int(m.group(2)), int(m.group(3) or 0)
    return {"gradient_tensors_compared": n, "fp32_oracle_rule_decided": fired, "decision_matched_comparison_decided": dec,
            "rate": round((fired) / max(n, 1), 4), "source": os.path.basename(f)}


def run_c4(args, dev, rank, world, ranks_info=None, sh_degree=None, light=False):
    """BASELINE configs[3]: 64 keyframes sharded over the ranks, RCCL gradient exchange, sharded fused Adam.
    sh_degree: overrides --c4-sh-degree (-1: `rgb_colors`, G = 14; 0..3: `shs` rows, G = 59); light: skip the non-uniform-scene leg of the
    one-GPU prediction.  The caller emits the line and tears the process group down."""
    import types
    if sh_degree is not None:
        args = types.SimpleNamespace(**dict(vars(args), c4_sh_degree=int(sh_degree)))
    import torch.distributed as dist
    from activesplat_amd import mapping as M, optim as O, parallel as PL, setup_camera
    from activesplat_amd import rasterizer as R
    from activesplat_amd import synthetic as syn
    W, H, N, KF = args.width, args.height, args.c4_gaussians, args.keyframes
    K = syn.intrinsics(W, H)
    raw = syn.shell_scene(N, seed=0, W=W, H=H)
    if args.c4_coherent:
        # a map grown frame by frame (add_new_gaussians appends the rows a frame sees) is coherent in memory: neighbours in the tensors are neighbours
        # in space.  The default scene is the pessimistic one (rows in random order: every wavefront of the per-Gaussian kernels holds some visible row)
        order = torch.argsort(torch.atan2(raw["means3D"][:, 0], raw["means3D"][:, 2]))
        raw = {k: v[order].contiguous() for k, v in raw.items()}
    lrs = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
    sh = args.c4_sh_degree >= 0
    if sh:
        # the map's colour as 16-coefficient SH rows (configs[2]'s map): the exchange carries G = 59 floats per Gaussian instead of 14
        g = torch.Generator().manual_seed(7)
        shs = 0.2 * torch.randn(N, 16, 3, generator=g)
        shs[:, 0, :] = (raw.pop("rgb_colors") - 0.5) / 0.28209479177387814
        raw["shs"] = shs
        lrs = {("shs" if k == "rgb_colors" else k): v for k, v in lrs.items()}

    def fresh():
        prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in raw.items()}
        prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
        prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
        var = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
        var["scene_radius"] = torch.tensor(10.0, device=dev)
        return prm, O.initialize_optimizer(prm, lrs), var
    params, opt, variables = fresh()
    cam = setup_camera(W, H, K, np.eye(4), device=dev, sh_degree=max(args.c4_sh_degree, 0))
    mine = set(PL.shard_keyframes(KF, rank, world)) if world > 1 else set(range(KF))
    keyframes = []
    for i in range(KF):
        a = 2 * np.pi * i / KF
        kf = dict(id=i, cam=cam, w2c=torch.eye(4, device=dev), pose7=[float(v) for v in syn.quat_from_yaw(a)] + [0.0, 0.0, 0.0])
        if i in mine or rank == 0 or (world > 1 and args.c4_partition == "lpt"):     # rank 0 also runs the whole batch alone afterwards; a
            # re-balanced partition can hand any keyframe to any rank (64 targets are 315 MB)
            im, depth = syn.make_targets(W, H, seed=100 + i)
            kf.update(im=im.to(dev), depth=depth.to(dev))
        keyframes.append(kf)
    weights = dict(im=0.5, depth=1.0)

    def loss_fn(p, kf, v):
        # fused_preprocess: the rasteriser's per-Gaussian kernels take the parameters themselves (no activation launches); accumulate_grads:
        # they add into .grad in the kernel (the multi-stream walk takes its gradients with autograd.grad instead)
        loss, v, _ = M.get_loss(p, kf, v, 0, weights, fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=True, pose7=kf["pose7"],
                                accumulate_grads=args.streams <= 1)
        return loss, v

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state = dict(v=variables, it=0)
    ex_ms, rd_ms = [], []
    every = int(args.c4_densify_every)
    ddict = dict(start_after=0, remove_big_after=0, stop_after=10 ** 9, densify_every=max(every, 1), grad_thresh=0.0, num_to_split_into=2,
                 removal_opacity_threshold=0.005, final_removal_opacity_threshold=0.005, reset_opacities=False, reset_opacities_every=10 ** 9)
    events = []

    def threshold_from_statistics(p, v, dd):
        # (the statistics are complete and identical on every rank here: the quantile is a function of replicated values)
        # a 4096-bin histogram instead of a selection (torch.kthvalue sorts: 10 ms at 2 M): the threshold is the upper edge of the bin in which
        # the cumulative count passes the quantile -- three small launches and one host read
        g = v["means2D_gradient_accum"] / v["denom"].clamp_min(1.0)
        hi = float(g.max())
        if not hi > 0.0:
            return dict(dd, grad_thresh=float("inf"))
        c = torch.cumsum(torch.histc(g, bins=4096, min=0.0, max=hi), 0)
        b = int(torch.searchsorted(c, torch.tensor([args.c4_densify_quantile * g.numel()], device=c.device, dtype=c.dtype)).clamp_max(4095))
        return dict(dd, grad_thresh=(b + 1) * hi / 4096.0)

    def densify_step(prm, o, st, w):
        """the event of step st['it'], if one is due: -> (N before, N after, grad_thresh) or None"""
        if not every or st["it"] % every:
            return None
        n0 = int(prm["means3D"].shape[0])
        seen_thr = {}

        def hook(p, v, dd):
            dd = threshold_from_statistics(p, v, dd); seen_thr["t"] = dd["grad_thresh"]
            return dd
        _, st["v"] = PL.sharded_densify(prm, st["v"], o, st["it"], ddict, accumulate=False, world=w, before_event=hook)
        return (n0, int(prm["means3D"].shape[0]), seen_thr.get("t"))

    if every:
        # one-time costs of the event path (code-object loads of the torch ops behind kthvalue / the index builds, allocator growth): paid on a
        # throwaway 4 k-Gaussian map before anything is timed -- a mapper that has densified once finds them paid
        try:
            small = {k: torch.nn.Parameter(v[:4096].detach().clone()) for k, v in params.items() if not k.startswith("cam_")}
            o_s = O.initialize_optimizer(small, {k: v for k, v in lrs.items() if k in small})
            for v_ in small.values():
                v_.grad = torch.zeros_like(v_)
            o_s.step(); o_s.zero_grad(set_to_none=True)
            v_s = {k: torch.rand(4096, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "timestep")}
            v_s.update(denom=torch.ones(4096, device=dev), scene_radius=torch.tensor(10.0, device=dev))
            densify_step(small, o_s, dict(v=v_s, it=every), 1)
            del small, o_s, v_s
        except Exception as e:
            note(f"densify warm-up failed: {e}")
    lpt = world > 1 and args.c4_partition == "lpt" and args.streams <= 1
    costs = PL.KeyframeCosts(KF) if lpt else None
    state["part"] = None

    def step():
        _, state["v"], _ = PL.sharded_keyframe_step(params, state["v"], keyframes, opt, loss_fn, rank=rank, world=world,
                                                    sharded_adam=True, streams=args.streams, timing=True, accumulate_statistics=bool(every),
                                                    partition=state["part"], costs=costs)
        state["it"] += 1
        if lpt and (state["it"] == 1 or state["it"] % max(args.c4_rebalance_every, 1) == 0):
            # every rank has recorded its own keyframes' tile-instance counts: one all-reduce of 64 floats, then the same LPT assignment everywhere
            state["part"] = costs.sync(dev).partition(world)
        if world > 1:
            ex_ms.append(PL.last_exchange.get("events"))
            rd_ms.append(PL.last_exchange.get("render_events"))
            ev = densify_step(params, opt, state, world)       # every k-th step: statistics all-reduced, moments gathered, rows moved, plan rebuilt
            if ev is not None:
                events.append(ev)
    if world > 1:
        for _ in range(args.warmup):
            step()
        barrier()
        ex_ms.clear(); rd_ms.clear(); events.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        D = int(R.last_stats["num_rendered"])
        exch = {k: v for k, v in PL.last_exchange.items() if k not in ("events", "render_events")}
        ex = [float(a.elapsed_time(b)) for a, b in (e for e in ex_ms if e)]
        # every rank's own keyframes (render + loss + backward, before the exchange), event-timed per step: the load balance of the shards
        rd = [float(a.elapsed_time(b)) for a, b in (e for e in rd_ms if e)]
        mine_ms = torch.tensor([float(np.mean(rd)) if rd else 0.0, float(np.mean(ex)) if ex else 0.0], device=dev, dtype=torch.float64)
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        per_rank = [round(float(t[0]), 3) for t in all_ms]
        per_rank_ex = [round(float(t[1]), 3) for t in all_ms]
        dd = torch.tensor([float(D)], device=dev, dtype=torch.float64)       # tile instances of every rank's last keyframe -> their mean
        dist.all_reduce(dd, op=dist.ReduceOp.SUM)
        D_mean = float(dd.item()) / world
        out = {
            "metric": METRIC + " (configs[3]: keyframes/s of the 64-keyframe optimiser step incl. loss, gradient exchange and Adam)",
            "value": round(KF * args.steps / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: {N} Gaussians, {KF} keyframes at {W}x{H} block-partitioned over {world} ranks "
                                   f"({KF // max(world, 1)} per rank), one optimiser step per batch, reduce-scatter of the [N,{'59' if sh else '14'}] fp32 gradient "
                                   "-> sharded fused Adam -> all-gather of the updated rows",
                       "gaussians": N, "width": W, "height": H, "keyframes_per_step": KF, "keyframes_per_rank_per_step": KF // max(world, 1),
                       "tile_instances_D_last_keyframe": D, "streams": args.streams, "exchange_floats_per_gaussian": 59 if sh else 14,
                       "grad_exchange": dict(exch, note="the collectives that actually ran (activesplat_amd.parallel.last_exchange)"),
                       "partition": args.c4_partition if lpt or args.c4_partition == "contiguous" else "contiguous (streams > 1)",
                       "keyframes_per_rank_last_step": [len(p_) for p_ in state["part"]] if state["part"] else [len(PL.shard_keyframes(KF, r, world)) for r in range(world)],
                       "densify_every": every, "densify_events_in_timed_region": [dict(n_before=a, n_after=b, grad_thresh=c) for a, b, c in events],
                       "gaussians_at_end": int(params["means3D"].shape[0]),
                       "loss": "fused mapping loss (L1 + SSIM + masked depth) through the single-pass RGB-D render",
                       "parallelism": f"keyframe-sharded x{world}"},
            "exchange_plus_adam_ms": round(float(np.mean(ex)), 4) if ex else None,
            "rccl_ranks": world if "nccl" in str(exch.get("backend")) else 0,
            "ranks": ranks_info,
            "per_rank_keyframes_ms": {"values": per_rank, "min": min(per_rank), "mean": round(float(np.mean(per_rank)), 3), "max": max(per_rank),
                                      "max_over_mean": round(max(per_rank) / max(float(np.mean(per_rank)), 1e-9), 4),
                                      "note": "a rank's own keyframes of one step (render + loss + backward, before the exchange), hipEvents, mean over the timed steps"},
            "per_rank_exchange_plus_adam_ms": per_rank_ex,
        }
        out["roofline"] = c4_roofline(int(params["means3D"].shape[0]), D_mean, W * H, sh, world, KF, dt / args.steps,
                                      float(np.mean(ex)) if ex else None, exch.get("bytes"))
    else:
        out = {}
    # the same batch on ONE GPU (rank 0 alone, no collectives, its own parameters and optimiser): the reference point of the
    # strong-scaling curve
    if rank == 0 and (world == 1 or not args.no_extras):
        p1, o1, v1 = fresh() if world > 1 else (params, opt, variables)
        s1 = dict(v=v1)

        def step1():
            _, s1["v"], _ = PL.sharded_keyframe_step(p1, s1["v"], keyframes, o1, loss_fn, rank=0, world=1, sharded_adam=True, streams=args.streams)
        n_ref = max(1, min(3, args.steps))
        step1()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(n_ref):
            step1()
        torch.cuda.synchronize()
        t_steps = time.perf_counter() - t1
        ev1 = None
        if every:
            # the single-GPU figure carries the same densify cadence as the sharded run: one event timed here (statistics of one more batch
            # accumulated first), its time spread over `every` steps
            _, s1["v"], _ = PL.sharded_keyframe_step(p1, s1["v"], keyframes, o1, loss_fn, rank=0, world=1, sharded_adam=True, streams=1, accumulate_statistics=True)
            s1["it"] = every
            torch.cuda.synchronize(); t2 = time.perf_counter()
            ev1 = densify_step(p1, o1, s1, 1)
            torch.cuda.synchronize()
            t_event = time.perf_counter() - t2
            t_steps += t_event * n_ref / every
        ref = KF * n_ref / t_steps
        if world > 1:
            out["single_gpu_same_workload_fps"] = round(ref, 2)
            out["speedup_vs_single_gpu_same_workload"] = round(out["value"] / ref, 3)
            out["roofline"]["single_gpu_frame_frac"] = round(ref * out["roofline"]["alg_bytes_per_keyframe"] / HBM_PEAK, 5)
            out["roofline"]["single_gpu_note"] = "the same batch on rank 0's GPU alone (no collective): keyframes/s x B_alg(keyframe) / 8 TB/s"
        else:
            out = {"workload": f"BASELINE configs[3] on ONE GPU: {N} Gaussians{' as 16-coefficient SH rows' if sh else ''}, {KF} keyframes per optimiser step (activations inside the per-Gaussian kernels of the RGB-D render -> "
                               "fused loss -> backward per keyframe, fused Adam), no collective", "keyframes_per_s": round(ref, 2),
                   "exchange_floats_per_gaussian": 59 if sh else 14,
                   "roofline": c4_roofline(int(p1["means3D"].shape[0]), int(R.last_stats["num_rendered"]), W * H, sh, 1, KF, KF / ref, None, None),
                   "ms_per_optimiser_step": round(KF / ref * 1e3, 3), "streams": args.streams,
                   "densify_every": every, "densify_event": None if ev1 is None else dict(n_before=ev1[0], n_after=ev1[1], grad_thresh=ev1[2], ms=round(t_event * 1e3, 3))}
            # ---- what ONE device can say about the 8-GPU run: every rank's shard of the batch timed alone (load balance: D differs per
            # view), and the exchange as a ONE-rank RCCL group runs it (pack -> reduce_scatter_tensor -> Adam -> all_gather_into_tensor ->
            # unpack; there the Adam still covers ALL rows and nothing crosses xGMI) ----
            try:
                R8 = args.predict_ranks
                per = []
                for r in range(R8):
                    ms = []
                    for rep in range(3):
                        PL.sharded_keyframe_step(p1, s1["v"], keyframes, o1, loss_fn, rank=r, world=R8, sharded_adam=True, streams=1, timing=True)
                        if rep:
                            ms.append(PL.render_ms())
                    per.append(round(float(np.mean(ms)), 3))
                pred = {"ranks": R8, "per_rank_ms": per, "max_over_mean": round(max(per) / float(np.mean(per)), 4),
                        "note": f"rank r's {KF // R8} keyframes of the batch (render + loss + backward) on this one device, hipEvents, mean of 2 steps"}
                try:
                    import socket
                    import torch.distributed as dist
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        port = sk.getsockname()[1]
                    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                    try:
                        xs = []
                        for rep in range(6):
                            PL.reduce_scatter_adam_step(p1, o1, timing=True)
                            if rep:
                                xs.append(PL.exchange_ms())
                        ex1 = float(np.mean(xs))
                        S = float(PL.last_exchange["bytes"])
                        wire = 2.0 * (S / R8) / 153e9 * 1e3           # direct reduce-scatter + all-gather: S/ranks per peer and direction, one xGMI link each
                        pred.update(exchange_floats_per_gaussian=59 if sh else 14, exchange_1rank_rccl_ms=round(ex1, 4), exchange_collectives=[PL.last_exchange.get("reduce"), PL.last_exchange.get("gather")],
                                    exchange_bytes=int(S), wire_model_ms=round(wire, 4),
                                    predicted_keyframes_per_s=round(KF / ((max(per) + ex1 + wire) * 1e-3), 1),
                                    predicted_speedup_vs_this_gpu=round(KF / ((max(per) + ex1 + wire) * 1e-3) / ref, 3),
                                    prediction="keyframes / (slowest rank's shard + the measured 1-rank RCCL exchange incl. Adam on ALL rows + the modelled xGMI "
                                               "time of the 8-rank collectives): a model from one device, not a measurement of 8")
                        # the same with the Adam launch at 1/ranks of the rows, as the 8-rank step runs it: pack + unpack as measured (they cover all rows on every
                        # rank), the Adam kernel's share from the rocprofv3 kernel table of this very exchange (profiles/rNN_c3step_kernel_stats.csv) divided by ranks
                        prk = profiled_kernels("c3step")
                        if prk:
                            tag = "64,64" if sh else "256,16"
                            us = {m: next((e["avg_us"] for k, e in prk["kernels"].items() if k.startswith(f"rows_kernel<{m},{tag}")), None) for m in (0, 1, 2)}
                            if all(us.values()):
                                ex8 = (us[0] + us[1] + us[2] / R8) * 1e-3 + wire
                                pred.update(exchange_kernels_us={"pack": us[0], "unpack": us[1], "adam_all_rows": us[2]},
                                            exchange_model_sharded_adam_ms=round(ex8, 4),
                                            predicted_keyframes_per_s_sharded_adam=round(KF / ((max(per) + ex8) * 1e-3), 1),
                                            predicted_speedup_sharded_adam=round(KF / ((max(per) + ex8) * 1e-3) / ref, 3),
                                            sharded_adam_note=f"pack + unpack (all rows, profiled) + Adam kernel / {R8} (each rank steps its row block) + the modelled wire time; "
                                                              "the figure above charges the Adam of ALL rows, which a one-rank group runs")
                    finally:
                        dist.destroy_process_group()
                except Exception as e:
                    pred["exchange_error"] = str(e)
                # ---- the same on a NON-uniform scene (three quarters of the Gaussians in half of the azimuth range: half the views see 3x the
                # tile instances): contiguous keyframe blocks against the LPT assignment on each keyframe's measured instance count ----
                try:
                    if light or sh:
                        raise RuntimeError("skipped on this leg (the uniform-scene figures above are the G = 59 column; the non-uniform scene runs on the G = 14 map)")
                    raw_u = syn.uneven_shell_scene(N, seed=0, W=W, H=H)
                    pu = {k: torch.nn.Parameter(v.to(dev)) for k, v in raw_u.items()}
                    pu["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
                    pu["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
                    ou = O.initialize_optimizer(pu, {k: v for k, v in lrs.items() if k in pu})
                    vu = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
                    kc = PL.KeyframeCosts(KF)
                    PL.sharded_keyframe_step(pu, vu, keyframes, ou, loss_fn, rank=0, world=1, sharded_adam=True, streams=1, costs=kc)   # one pass: every keyframe's D
                    kc.sync()
                    d_per = [c - kc.gw * N for c in kc.cost]
                    policies = {"contiguous": [list(PL.shard_keyframes(KF, r, R8)) for r in range(R8)], "lpt": kc.partition(R8)}
                    un = {"scene": "three quarters of the Gaussians in half of the azimuth range", "tile_instances_per_keyframe": {"min": int(min(d_per)), "max": int(max(d_per)),
                          "mean": int(np.mean(d_per))}}
                    for name, part in policies.items():
                        per_u = []
                        for r in range(R8):
                            ms = []
                            for rep in range(3):
                                PL.sharded_keyframe_step(pu, vu, keyframes, ou, loss_fn, rank=r, world=R8, sharded_adam=True, streams=1, timing=True, partition=part)
                                if rep:
                                    ms.append(PL.render_ms())
                            per_u.append(round(float(np.mean(ms)), 3))
                        un[name] = {"per_rank_ms": per_u, "keyframes_per_rank": [len(p_) for p_ in part], "max_over_mean": round(max(per_u) / float(np.mean(per_u)), 4)}
                    un["slowest_rank_speedup_lpt_vs_contiguous"] = round(max(un["contiguous"]["per_rank_ms"]) / max(un["lpt"]["per_rank_ms"]), 3)
                    pred["nonuniform_scene"] = un
                    del pu, ou, vu
                except Exception as e:
                    pred["nonuniform_scene"] = {"error": str(e)}
                out["eight_gpu_prediction"] = pred
            except Exception as e:
                out["eight_gpu_prediction"] = {"error": str(e)}
    if world > 1:
        barrier()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=2_000_000)
    ap.add_argument("--sh-degree", type=int, default=3, help="-1: precomputed colours (SH degree 0 of the reference's mapper)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--streams", type=int, default=1,
                    help="configs[3]: HIP streams a rank's keyframes are issued on in turn (1: with the chained backward blend a 640x480 keyframe fills the chip "
                         "by itself -- 1371-1442 keyframes/s on one stream against 1276-1278 on two at 2 M Gaussians)")
    ap.add_argument("--no-extras", action="store_true", help="skip every leg but the timed region")
    ap.add_argument("--prep-seconds", type=float, default=0.5, help="untimed preparation in front of the warm-up steps (clock ramp, allocator)")
    ap.add_argument("--cpu-threads", type=int, default=64, help="upper bound on the host threads of the cpu_baseline leg")
    ap.add_argument("--c4-gaussians", type=int, default=2_000_000, help="configs[3]: Gaussians of the replicated map")
    ap.add_argument("--keyframes", type=int, default=64, help="configs[3]: keyframes per optimiser step, sharded over the ranks")
    ap.add_argument("--c4-sh-degree", type=int, default=-1, help="configs[3]: -1 = `rgb_colors` (the reference mapper's map, G = 14 floats per Gaussian "
                                                                 "in the exchange); 0..3 = 16-coefficient SH rows (`shs`, G = 59)")
    ap.add_argument("--c4-one-map", action="store_true", help="configs[3]: time only the map --c4-sh-degree names (default: the rgb_colors map is `value` and the "
                                                             "same steps are timed again on the SH-3 map, reported as `sh3_map_G59`)")
    ap.add_argument("--predict-ranks", type=int, default=8, help="configs[3] on one GPU: ranks of the load-balance / exchange prediction leg")
    ap.add_argument("--c4-coherent", action="store_true", help="configs[3]: the map's rows ordered by azimuth (a map grown frame by frame) instead of randomly")
    ap.add_argument("--c4-partition", choices=("lpt", "contiguous"), default="lpt", help="configs[3]: keyframes to ranks by longest-processing-time "
                    "assignment on every keyframe's last tile-instance count (costs all-reduced every --c4-rebalance-every steps), or contiguous blocks")
    ap.add_argument("--c4-rebalance-every", type=int, default=10)
    ap.add_argument("--c4-densify-every", type=int, default=20, help="configs[3]: a densify event (all-reduced statistics -> gathered moments -> fused clone / "
                    "split / cull with the split offsets drawn from a replicated seed -> rebuilt shard plan) every k-th optimiser step; 0: never")
    ap.add_argument("--c4-densify-quantile", type=float, default=0.995, help="configs[3]: the event's gradient threshold is this quantile of the batch's "
                    "reduced mean-2D gradient statistic, so that an event clones / splits ~0.5 %% of the map and the workload stays configs[3]'s")
    ap.add_argument("--workload", choices=("auto", "c4"), default="auto", help="auto: configs[2]'s render on one GPU, configs[3] on several")
    args = ap.parse_args()

    # development knobs for exercising the N > 1 code path on a 1-GPU box: all ranks on device 0, gloo instead of RCCL
    same_dev = bool(os.environ.get("BENCH_SAME_DEVICE"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher's environment: start the N ranks ourselves (one process per GPU, RCCL) -- a bare run must
        # never measure one GPU and print it under --gpus N
        raise SystemExit(self_launch(args.gpus, same_dev))
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    dist_on = world > 1
    if dist_on and not same_dev and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but {torch.cuda.device_count()} device(s) visible (BENCH_SAME_DEVICE=1 puts every rank on device 0: development only)")
    torch.cuda.set_device(0 if same_dev else local_rank)
    dev = torch.device("cuda", 0 if same_dev else local_rank)
    ranks_info = None
    if dist_on:
        import torch.distributed as dist
        want = "gloo" if os.environ.get("BENCH_BACKEND") == "gloo" else "nccl"
        if want == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        # what is really running: N ranks, the backend asked for, N distinct devices
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        assert str(dist.get_backend()).lower() == want, dist.get_backend()
        prop = torch.cuda.get_device_properties(dev)
        me = dict(rank=rank, pid=os.getpid(), device=int(dev.index), name=prop.name,
                  uuid=str(getattr(prop, "uuid", "")), pci_bus_id=int(getattr(prop, "pci_bus_id", -1)))
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
        ids = {(r["device"], r["uuid"], r["pci_bus_id"]) for r in ranks_info}
        if not same_dev and len(ids) != world:
            raise SystemExit(f"bench.py: {world} ranks on {len(ids)} distinct device(s): {ranks_info}")

    from activesplat_amd import GaussianRasterizer, _lib, setup_camera
    from activesplat_amd import rasterizer as R
    from activesplat_amd import synthetic as syn
    lib = _lib.get()                                        # fail loudly if the HIP library is missing
    if os.environ.get("GS_CHAIN_OFF"):                      # development knob (A/B): one backward walker per quadrant, no chained pieces
        _lib.check(lib.gs_set_backward_chain(1, -1))
    if dist_on or args.workload == "c4":
        out = run_c4(args, dev, rank, world, ranks_info=ranks_info)
        if args.c4_sh_degree < 0 and not args.c4_one_map:
            # the second exchange SURVEY 8(d)/(e) names for configs[3]: the map's colour as 16-coefficient SH rows (configs[2]'s map), G = 59 floats
            # per Gaussian, 472 MB at 2 M -- the same timed region (same steps, same barriers) on that map, reported inside the one line
            o2 = run_c4(args, dev, rank, world, ranks_info=ranks_info, sh_degree=3, light=True)
            if dist_on and rank == 0:
                out["sh3_map_G59"] = {k: o2.get(k) for k in ("value", "unit", "ms_per_step", "exchange_plus_adam_ms", "per_rank_keyframes_ms", "per_rank_exchange_plus_adam_ms",
                                                             "single_gpu_same_workload_fps", "speedup_vs_single_gpu_same_workload", "roofline")}
                out["sh3_map_G59"]["config"] = {k: o2["config"].get(k) for k in ("workload", "exchange_floats_per_gaussian", "grad_exchange", "gaussians_at_end",
                                                                                 "densify_events_in_timed_region", "tile_instances_D_last_keyframe")}
            elif not dist_on:
                out["sh3_map_G59"] = o2
        if dist_on:
            import torch.distributed as dist
            if rank == 0:
                emit(out)
            dist.destroy_process_group()
        else:
            emit(out)
        return

    W, H, N = args.width, args.height, args.gaussians
    deg = args.sh_degree if args.sh_degree >= 0 else None
    wl = RenderWorkload(N, W, H, dev, sh_degree=deg)
    # untimed preparation in front of the W warm-up steps: code-object loads, the caching allocator's pools, the optimistic launch's
    # capacity guess -- and the device's clocks: a fresh process' first ~0.3 s of kernels run below the sustained clock (the same frames
    # measure 5 % slower there than a second later; scripts/exp/wall_vs_events.py)
    t_prep = time.perf_counter()
    while time.perf_counter() - t_prep < args.prep_seconds:
        for _ in range(8):
            wl.step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    D = int(R.last_stats["num_rendered"])
    ms_per_step = dt / args.steps * 1e3
    fps = args.steps / dt
    fb = frame_bytes(N, D, W * H, sh=deg is not None)
    which = "configs[2]'s render (the north star's target configuration)" if (N, deg) == (2_000_000, 3) else "custom"
    out = {
        "metric": METRIC, "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE {which}: {N} Gaussians, SH degree {deg if deg is not None else '0 (colors_precomp)'}, {W}x{H}, "
                               "forward (colour+radii+depth+opacity) + backward (all input gradients), frames strictly one after the other",
                   "gaussians": N, "sh_degree": deg, "width": W, "height": H, "tile_instances_D": D, "streams": 1,
                   "parallelism": "single GPU"},
    }

    if not args.no_extras:
        note(f"timed region done: {fps:.1f} frames/s; roofline leg")
        # ---- roofline leg: per-stage hipEvents over a second pass of the same K steps ----
        stages, sb, _ = wl.stages(lib, args.steps)
        dom = max((k for k in stages if k in sb), key=lambda k: stages[k]["avg_us"])
        ach = sb[dom] / (stages[dom]["avg_us"] * 1e-6)
        # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process; the most recent separate-pass
        # collection of THIS configuration (scripts/gpu_round_end.sh -> profiles/rNN_2m_pmc_summary.json) is reported if present
        traffic, traffic_src, valu, all_traffic = None, None, None, None
        try:
            f2 = pmc_file("r[0-9][0-9]_2m_pmc_summary.json")
            if f2 and (N, deg, W, H) == (2_000_000, 3, 640, 480):
                pm = json.load(open(f2))
                kern = {"blend_backward": "blend_backward_kernel", "blend_forward": "blend_forward_streams_kernel",
                        "preprocess_forward+scan": "preprocess_forward_kernel", "preprocess_backward": "preprocess_backward_kernel"}.get(dom)
                if kern in pm and "traffic_bytes" in pm[kern]:
                    traffic, traffic_src = int(pm[kern]["traffic_bytes"]), os.path.basename(f2)
                if kern in pm and "SQ_INSTS_VALU" in pm[kern]:
                    vi = float(pm[kern]["SQ_INSTS_VALU"])
                    valu = {"wave_insts_per_launch": int(vi), "achieved_ginst_s": round(vi / (stages[dom]["avg_us"] * 1e-6) / 1e9, 1),
                            "peak_ginst_s": VALU_ISSUE_PEAK / 1e9, "frac": round(vi / (stages[dom]["avg_us"] * 1e-6) / VALU_ISSUE_PEAK, 4),
                            "source": os.path.basename(f2),
                            # the denominator is this build's own micro-benchmark; the same instructions against the other candidates, so that the
                            # reader can re-derive `frac`: v_fma_f32 measured 651 Ginst/s, the guide's 2-cycle wave64 issue at 2.4 GHz 1228.8
                            "valu_peak_source": "measured v_mul_f32 / v_add_f32 issue rate, 890 Ginst/s (scripts/exp/valu_issue.hip, profiles/r02_valu_issue.txt); "
                                                "v_fma_f32 measured 651; MI355X_MICROARCH.md 2-cycle issue x 1024 SIMDs x 2.4 GHz = 1228.8",
                            "frac_vs_measured_v_fma_651": round(vi / (stages[dom]["avg_us"] * 1e-6) / 651e9, 4),
                            "frac_vs_guide_issue_1228_8": round(vi / (stages[dom]["avg_us"] * 1e-6) / 1228.8e9, 4)}
                all_traffic = {k: int(v["traffic_bytes"]) for k, v in pm.items() if "traffic_bytes" in v}
            # counter bytes next to the model's bytes, per stage (the binning stages share one kernel name in the counter tables: one figure for all)
            if all_traffic:
                first = lambda *names: next((all_traffic[n] for n in names if n in all_traffic), None)  # noqa: E731
                per_stage = {"preprocess_forward+scan": first("preprocess_forward_sh48_kernel", "preprocess_forward_kernel"),
                             "blend_forward": first("blend_forward_streams_kernel"), "blend_backward": first("blend_backward_kernel"),
                             "preprocess_backward": first("preprocess_backward_kernel")}
                for k, v in per_stage.items():
                    if v is not None and k in stages:
                        stages[k]["pmc_bytes"] = int(v)
                        stages[k]["frac_hbm_pmc"] = round(v / (stages[k]["avg_us"] * 1e-6) / HBM_PEAK, 4)
                binning = 2 * all_traffic.get("tile_bin_kernel", 0) + all_traffic.get("tile_colscan_kernel", 0) + all_traffic.get("tile_scan_kernel", 0) + \
                    all_traffic.get("tile_bucket_sort_kernel", 0)
                if binning and "tile_scatter+sort" in stages and "tile_count+scan" in stages:
                    us = stages["tile_scatter+sort"]["avg_us"] + stages["tile_count+scan"]["avg_us"]
                    stages["tile_scatter+sort"]["pmc_bytes_with_count_and_scans"] = int(binning)
                    stages["tile_scatter+sort"]["frac_hbm_pmc_with_count_and_scans"] = round(binning / (us * 1e-6) / HBM_PEAK, 4)
        except Exception:
            pass
        # Which roofline bounds the dominant kernel?  Counter traffic within 15 % of the algorithmic bytes (nothing re-read) AND the vector ALUs
        # busy in more than 60 % of the kernel's cycles (SQ_ACTIVE_INST_VALU x 4 cycles per wave64 instruction / 1024 SIMDs / the kernel's
        # cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs): the kernel is VALU-issue-bound, and the line says so -- `frac` is then against the
        # chip's measured wave64 VALU issue rate, with the HBM figures beside it (the line names the bound it reports against)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 5)}
        try:
            pk = pm.get(kern, {}) if traffic is not None else {}
            if valu and "SQ_ACTIVE_INST_VALU" in pk and "GRBM_GUI_ACTIVE" in pk:
                busy = float(pk["SQ_ACTIVE_INST_VALU"]) * 4.0 / (1024.0 * float(pk["GRBM_GUI_ACTIVE"]) / 8.0)
                valu["valu_busy_frac"] = round(busy, 4)
                if 0.85 <= traffic / sb[dom] <= 1.15 and busy > 0.6:
                    roof = {"bound": "valu", "kernel": dom, "achieved": valu["achieved_ginst_s"], "peak": valu["peak_ginst_s"], "unit": "Ginst/s (wave64 VALU instructions)",
                            "frac": valu["frac"], "valu_busy_frac": valu["valu_busy_frac"], "valu_peak_source": valu["valu_peak_source"],
                            "frac_vs_measured_v_fma_651": valu["frac_vs_measured_v_fma_651"], "frac_vs_guide_issue_1228_8": valu["frac_vs_guide_issue_1228_8"],
                            "why": f"counter traffic {traffic / sb[dom]:.2f} x the algorithmic bytes and the vector ALUs busy in {busy:.0%} of the kernel's cycles: "
                                   "not an HBM-bound kernel; instructions per launch from the PMC pass, duration live",
                            "frac_hbm": round(ach / HBM_PEAK, 5), "achieved_hbm_gbs": round(ach / 1e9, 2), "peak_hbm_gbs": HBM_PEAK / 1e9}
        except Exception:
            pass
        out["roofline"] = {**roof, "traffic": traffic, "traffic_source": traffic_src,
                           "alg_bytes_per_launch": sb[dom], "avg_us": stages[dom]["avg_us"], "valu_issue": valu,
                           "frame_alg_bytes": fb, "frame_frac_sequential": round(fb / (ms_per_step * 1e-3) / HBM_PEAK, 5),
                           "frame_frac_note": "whole forward+backward frame: SURVEY 8(d)'s algorithmic bytes / the timed region's sequential frame time "
                                              "(`value`); target of the north star >= 0.40",
                           "stage_sum_us": round(sum(s["avg_us"] for s in stages.values()), 1),
                           "pmc_traffic_bytes_by_kernel": all_traffic, "stages": stages}
        # ---- secondary figures: step-time spread by events, the two-stream batch rate, forward-only rate, blend flop rate ----
        try:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); wl.step(); b.record()
            torch.cuda.synchronize()
            ts = np.sort(np.array([a.elapsed_time(b) for a, b in ev]))
            out["step_ms_by_events"] = {"frames": len(ts), "mean": round(float(ts.mean()), 4), "p10": round(float(ts[len(ts) // 10]), 4),
                                        "p50": round(float(ts[len(ts) // 2]), 4), "p90": round(float(ts[(len(ts) * 9) // 10]), 4)}
            t2 = wl.two_streams(max(args.steps // 2, 10))
            out["two_stream_batch"] = {"frames_per_s": round(1.0 / t2, 1), "frame_frac_hbm": round(fb / t2 / HBM_PEAK, 4),
                                       "note": "the same frames as independent keyframes of a batch, issued on two HIP streams in turn (not `value`)"}
            rv_ng = {k: v.detach() for k, v in wl.rv.items()}
            m2d0 = torch.zeros(N, 3, device=dev)
            with torch.no_grad():
                for _ in range(5):
                    GaussianRasterizer(raster_settings=wl.cam)(means2D=m2d0, **rv_ng)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(args.steps):
                    GaussianRasterizer(raster_settings=wl.cam)(means2D=m2d0, **rv_ng)
                torch.cuda.synchronize()
                out["forward_only_fps"] = round(args.steps / (time.perf_counter() - t1), 1)
                with R.capture() as state:
                    GaussianRasterizer(raster_settings=wl.cam)(means2D=m2d0, **rv_ng)
            il = state["il"]
            ncontrib = state["image"][il.n_contrib:il.n_contrib + 4 * W * H].view(torch.int32)
            E = int(ncontrib.to(torch.int64).sum().item())          # list entries walked, summed over pixels
            del state
            tf, tb = stages["blend_forward"]["avg_us"] * 1e-6, stages["blend_backward"]["avg_us"] * 1e-6
            out["blend_flops"] = {"evaluations_E": E, "forward_frac_fp32_peak": round(E * 12 / tf / FP32_PEAK, 4),
                                  "backward_frac_fp32_peak": round(E * 40 / tb / FP32_PEAK, 4), "peak_tflops": FP32_PEAK / 1e12,
                                  "note": "SURVEY 8(d): F_alg = E*(12 fwd + 40 bwd) flop, E = sum of n_contrib"}
            # the dominant kernel against the fp32 vector peak, inside `roofline` (SURVEY 8(d)'s secondary figure for the blend)
            if dom in ("blend_backward", "blend_forward"):
                fl = E * (40 if dom == "blend_backward" else 12)
                out["roofline"]["alg_flops_per_launch"] = int(fl)
                out["roofline"]["frac_flops"] = round(fl / (stages[dom]["avg_us"] * 1e-6) / FP32_PEAK, 4)
                out["roofline"]["flops_peak_tflops"] = FP32_PEAK / 1e12
            del rv_ng, m2d0
        except Exception as e:
            out["secondary_error"] = str(e)
        # ---- cpu_baseline leg: the C oracle (a port of the path, OpenMP over the pixel rows) on ALL host cores, same workload ----
        try:
            from oracle.gs_oracle import Oracle
            from tests import util
            o = Oracle("f32")
            cores = min(os.cpu_count() or 1, args.cpu_threads)
            cpu_model = "unknown"
            try:
                with open("/proc/cpuinfo") as fh:
                    cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
            except OSError:
                pass
            o.set_threads(cores)
            note(f"cpu_baseline: oracle on {cores} threads")
            rv_cpu = {k: v.detach().cpu() for k, v in wl.rv.items()}
            dL_cpu = wl.dL.cpu()
            t1 = time.perf_counter()
            f = util.run_oracle(o, wl.cam, rv_cpu, dL_cpu)           # first frame: page-in, thread pool (timed too if it is the only one)
            t_first = time.perf_counter() - t1
            t1 = time.perf_counter(); n_cpu = 0
            while n_cpu < 50 and time.perf_counter() - t1 + t_first < CPU_SECONDS:
                f = util.run_oracle(o, wl.cam, rv_cpu, dL_cpu); n_cpu += 1
            tc = (time.perf_counter() - t1) / n_cpu if n_cpu else t_first
            n_cpu = max(n_cpu, 1)
            o.set_threads(1)
            try:
                # "PSNR vs ref" half of the metric: this device's render and gradients against the oracle's, same inputs
                with torch.no_grad():
                    col_gpu = GaussianRasterizer(raster_settings=wl.cam)(means2D=torch.zeros(N, 3, device=dev), **{k: v.detach() for k, v in wl.rv.items()})[0]
                mse = float(((col_gpu.cpu().double() - torch.from_numpy(np.asarray(f["color"])).double()) ** 2).mean())
                g_gpu = wl.step()
                rel = {}
                for k, g in zip(wl.keys, g_gpu[:len(wl.keys)]):
                    gk = {"shs": "shs", "colors_precomp": "colors_precomp"}.get(k, k)
                    ref = torch.from_numpy(np.asarray(f["grads"][gk])).double().reshape(g.shape)
                    rel[k] = float((g.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30))
                out["parity_vs_oracle"] = {"psnr_color_db": round(10 * np.log10(1.0 / max(mse, 1e-30)), 1),
                                           "grad_rel_l2_max": float(f"{max(rel.values()):.3g}"), "oracle": "oracle/gs_oracle.c fp32 build, same inputs",
                                           "status": "parity UNPINNED for the kernel arithmetic (the reference's rasteriser source is an empty submodule): oracle = restated published algorithm",
                                           "gradient_rule": "vs the fp64 oracle: rtol 1e-3 / atol 1e-6 |g|inf on >= 99.5 % of the elements and relative L2 < 1e-3; "
                                                            "else <= 1.5 x the fp32 oracle's own error; else the same two tiers against the oracles re-run with the OTHER "
                                                            "visibility decision at pixels of <= 3 Gaussians that provably sit within 1e-5 of the alpha = 1/255 threshold (DESIGN.md section 6)",
                                           "gpu_suite_tally": suite_tally()}
                del g_gpu
            except Exception as e:
                out["parity_vs_oracle"] = {"error": str(e)}
            out["cpu_baseline"] = {"value": round(1.0 / tc, 5), "unit": "frames/s", "cores": cores, "cpu_model": cpu_model, "host_cores_total": os.cpu_count(), "kind": "port",
                                   "sample": f"{n_cpu} frame(s) forward+backward of the same workload (N={N}, SH degree {deg}, {W}x{H}, D={f['D']}) "
                                             f"by oracle/gs_oracle.c (fp32, gcc -O2 -fopenmp, {cores} threads over pixel rows / Gaussians; "
                                             f"preprocess and the key sort are single-threaded), {tc:.2f} s per frame"}
            del rv_cpu, f
        except Exception as e:      # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        # ---- BASELINE configs[0]: 10k Gaussians, one 640x480 view, CPU PyTorch forward-only render (plumbing check), next to the HIP forward ----
        th0 = torch.get_num_threads()
        try:
            from oracle import dense_torch as DT
            from tests import util
            rs0, rv0 = util.scene(10_000, W, H, seed=0)
            cam0 = util.cam_dict(rs0)
            note("configs[0]: tiled CPU PyTorch render")
            torch.set_num_threads(min(os.cpu_count() or 1, 16))       # tile-sized tensors: more threads only add fork/join cost
            with torch.no_grad():
                t1 = time.perf_counter()
                ref0 = DT.render_dense(cam0, rv0["means3D"], rv0["opacities"], colors=rv0["colors_precomp"], scales=rv0["scales"],
                                       rotations=rv0["rotations"], tiled=True)
                t_cpu = time.perf_counter() - t1
                rs0d = setup_camera(W, H, wl.K, np.eye(4), device=dev)
                rv0d = {k: v.to(dev) for k, v in rv0.items()}
                m0 = torch.zeros(10_000, 3, device=dev)
                for _ in range(3):
                    got0 = GaussianRasterizer(raster_settings=rs0d)(means2D=m0, **rv0d)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(20):
                    got0 = GaussianRasterizer(raster_settings=rs0d)(means2D=m0, **rv0d)
                torch.cuda.synchronize()
                t_gpu = (time.perf_counter() - t1) / 20
            mse0 = float(((got0[0].cpu().double() - ref0["color"].double()) ** 2).mean())
            out["configs0"] = {"workload": "BASELINE configs[0]: 10k Gaussians, one 640x480 view, forward only",
                               "cpu_pytorch_frames_per_s": round(1.0 / t_cpu, 3), "cpu_threads": torch.get_num_threads(),
                               "cpu_path": "oracle/dense_torch.render_dense(tiled=True): plain PyTorch CPU ops, tile by tile",
                               "hip_forward_frames_per_s": round(1.0 / t_gpu, 1),
                               "psnr_hip_vs_cpu_pytorch_db": round(10 * np.log10(1.0 / max(mse0, 1e-30)), 1)}
            if isinstance(out.get("cpu_baseline"), dict):
                # "the reference's CPU PyTorch path" of the north star, next to the C port: SURVEY 8(d) -- the reference has no CPU render path, so this
                # is the build's dense PyTorch restatement on configs[0] (10 k Gaussians, forward only), the one size it finishes in seconds
                out["cpu_baseline"]["cpu_pytorch_configs0"] = {"value": round(1.0 / t_cpu, 3), "unit": "frames/s (forward only, 10k Gaussians, 640x480)",
                                                               "cores": torch.get_num_threads(), "kind": "port (oracle/dense_torch.py, plain PyTorch CPU ops)",
                                                               "hip_forward_same_scene_frames_per_s": round(1.0 / t_gpu, 1)}
        except Exception as e:
            out["configs0"] = {"error": str(e)}
        torch.set_num_threads(th0)
        params1 = None
        # ---- side legs: configs[1] (500 k, SH-0) and the 2 M map with precomputed colours ----
        del wl
        torch.cuda.empty_cache()
        try:
            note("configs[1] and 2M SH-0 legs")
            side = {}
            for name, n_, label in (("configs1_500k_sh0", 500_000, "BASELINE configs[1]: 500k Gaussians, 640x480 RGB+depth forward+backward, SH degree 0"),
                                    ("render_2M_sh0", 2_000_000, "2M Gaussians, precomputed colours (the reference mapper's sh_degree 0)")):
                w1 = RenderWorkload(n_, W, H, dev, sh_degree=None)
                for _ in range(4):
                    w1.step()
                t = w1.sequential(60, 10)
                st1, _, D1 = w1.stages(lib, 20)
                fb1 = frame_bytes(n_, D1, W * H)
                t2 = w1.two_streams(40)
                side[name] = {"workload": label, "sequential_frames_per_s": round(1.0 / t, 1), "ms_per_frame": round(t * 1e3, 4), "tile_instances_D": D1,
                              "frame_alg_bytes": fb1, "frame_frac_hbm": round(fb1 / t / HBM_PEAK, 4), "two_stream_frames_per_s": round(1.0 / t2, 1),
                              "stages": {k: {"avg_us": v["avg_us"], "frac_hbm": v["frac_hbm"]} for k, v in st1.items()}}
                try:
                    # the blend pair against the fp32 vector peak (F_alg = E x (12 fwd + 40 bwd) flop, E = sum of n_contrib): how much of a
                    # blend's time is WORK -- early termination saturates the evaluations per pixel, so E does not grow with the Gaussian count
                    with torch.no_grad(), R.capture() as st_c:
                        GaussianRasterizer(raster_settings=w1.cam)(means2D=torch.zeros(n_, 3, device=dev), **{k: v.detach() for k, v in w1.rv.items()})
                    il1 = st_c["il"]
                    E1 = int(st_c["image"][il1.n_contrib:il1.n_contrib + 4 * W * H].view(torch.int32).to(torch.int64).sum().item())
                    del st_c
                    side[name]["blend_flops"] = {"evaluations_E": E1, "forward_frac_fp32_peak": round(E1 * 12 / (st1["blend_forward"]["avg_us"] * 1e-6) / FP32_PEAK, 4),
                                                 "backward_frac_fp32_peak": round(E1 * 40 / (st1["blend_backward"]["avg_us"] * 1e-6) / FP32_PEAK, 4)}
                except Exception as e:
                    side[name]["blend_flops"] = {"error": str(e)}
                if n_ == 500_000:
                    params1 = w1.params
                    pr1 = profiled_kernels("c1")
                    if pr1:
                        for st_name, pat in (("preprocess_forward+scan", "preprocess_forward_kernel"), ("blend_forward", "blend_forward_streams_kernel"),
                                             ("blend_backward", "blend_backward_kernel"), ("preprocess_backward", "preprocess_backward_kernel")):
                            by, _ = counters_for(pr1, pat)
                            if by is not None and st_name in side[name]["stages"]:
                                side[name]["stages"][st_name]["pmc_bytes"] = by
                                side[name]["stages"][st_name]["frac_hbm_pmc"] = round(by / (side[name]["stages"][st_name]["avg_us"] * 1e-6) / HBM_PEAK, 4)
                        side[name]["rocprof"] = pr1
                del w1
                torch.cuda.empty_cache()
            out["side_legs"] = side
        except Exception as e:
            out["side_legs"] = {"error": str(e)}
        # ---- the reference's own operating points (config/datasets/gibson.json: 256 x 256; config/datasets/gibson_high_resolution.json +
        # config/env/activesplat_high_resolution_pointnav.yaml: 512 x 512, 10 iterations per mapped frame), SH degree 0 ----
        try:
            note("operating points 256x256 / 512x512")
            ops = {}
            for w_, n_ in ((256, 200_000), (256, 1_000_000), (512, 200_000), (512, 1_000_000)):
                w1 = RenderWorkload(n_, w_, w_, dev, sh_degree=None)
                for _ in range(4):
                    w1.step()
                t = w1.sequential(60, 10)
                st1, _, D1 = w1.stages(lib, 20)
                p1 = w1.params
                del w1
                torch.cuda.empty_cache()
                flags = dict(fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=True)
                wall, ev = mapping_iteration_ms(dev, p1, w_, w_, dict(flags, adam=True), iters=40, warm=10)
                wall0, ev0 = mapping_iteration_ms(dev, p1, w_, w_, flags, iters=40, warm=10)
                walld, evd = mapping_iteration_ms(dev, p1, w_, w_, dict(flags, direct=True), iters=40, warm=10)
                ops[f"{w_}x{w_}_{n_ // 1000}k"] = {"render_fwd_bwd_us": round(t * 1e6, 1), "tile_instances_D": D1,
                                                   "stages_us": {k: v["avg_us"] for k, v in st1.items()},
                                                   "mapping_iteration_ms": {"without_autograd": round(walld, 4), "without_autograd_gpu": round(evd, 4),
                                                                            "fused_adam_in_backward": round(wall, 4), "fused_adam_in_backward_gpu": round(ev, 4),
                                                                            "fused_separate_adam": round(wall0, 4), "fused_separate_adam_gpu": round(ev0, 4)}}
                torch.cuda.empty_cache()
            out["operating_points"] = dict(ops, note="the reference's frame sizes, precomputed colours; render = GaussianRasterizer forward + backward, frames one after "
                                                     "the other; mapping iteration = get_loss (single-pass RGB-D render from the parameters, fused loss) + backward "
                                                     "+ Adam + zero_grad: wall clock and hipEvent span per iteration")
        except Exception as e:
            out["operating_points"] = {"error": str(e)}
        # ---- configs[2]'s optimise loop: 2 M / SH-3, 100 iterations, fused Adam (5 tensors incl. the coefficients) + one densify event ----
        try:
            note("configs[2] optimise loop")
            from activesplat_amd.workloads import configs2_optimise_loop
            configs2_optimise_loop(4096, 12, "cuda", densify_every=5)          # one-time code-object loads of every kernel / torch op involved
            torch.manual_seed(0)
            lib.gs_profile_enable(1)
            r = configs2_optimise_loop(2_000_000, 100, "cuda", time_it=True)
            prof = _lib.profile_collect()
            lib.gs_profile_enable(0)
            n_after = r["counts"][-1] if r["counts"] else 2_000_000
            adam_ms, adam_calls = prof.get("adam", (0.0, 0))
            dens_ms = [round(x * 1e3, 3) for x in r["densify_seconds"]]
            dens_bytes = 2 * (59 + 118) * 4 * 0.5 * (2_000_000 + n_after)
            out["configs2_loop"] = {
                "workload": "BASELINE configs[2]: 2M Gaussians, SH degree 3, 640x480, 100-iteration optimise loop (activations inside the per-Gaussian kernels of the single-pass "
                            "RGB-D render -> fused loss -> backward with the Adam step inside its per-Gaussian kernel), densify every 50 iterations (one event in the loop; "
                            "that iteration takes the separate Adam step)",
                "iters": 100, "seconds": round(r["seconds"], 4), "iters_per_s": round(100 / r["seconds"], 2),
                "gaussians_start": 2_000_000, "gaussians_after_densify": r["counts"], "loss_first_last": r["losses"],
                "stages_us": {k: round(ms / c * 1e3, 1) for k, (ms, c) in prof.items() if c},
                "backward_with_adam": (lambda us: {
                    "kernel": "preprocess_backward_kernel<3, true, true> (gs_render_backward_raw_adam: per-Gaussian backward + the Adam step of all five tensors)",
                    "avg_us": us, "alg_bytes": int(2_000_000 * (152 + 59 * 24 + 12)),
                    "frac_hbm": round(2_000_000 * (152 + 59 * 24 + 12) / (us * 1e-6) / HBM_PEAK, 4) if us else None,
                    "note": "per Gaussian: 152 B of backward inputs (means, scales, rotation, opacity, radius, 64-byte gradient record, 40-byte SH Jacobian) + "
                            "59 parameters x (read p, m, v; write p, m, v) + 12 B of means2D gradient; the 99 non-event iterations of the loop"})(
                    round(prof["preprocess_backward"][0] / max(prof["preprocess_backward"][1], 1) * 1e3, 1) if "preprocess_backward" in prof else None),
                "adam": {"kernel": "adam_multi_kernel (event iterations only: the step of the other iterations is inside the backward)", "calls": adam_calls,
                         "avg_us": round(adam_ms / max(adam_calls, 1) * 1e3, 1)},
                "densify_event": {"ms": dens_ms, "alg_bytes": int(dens_bytes),
                                  "frac_hbm": round(dens_bytes / (dens_ms[0] * 1e-3) / HBM_PEAK, 4) if dens_ms else None,
                                  "note": "one classification launch, one index, one gather per tensor: 2 x (59 parameter + 118 moment floats) x N bytes"}}
            pr2 = profiled_kernels("c2loop")
            if pr2:
                c2 = out["configs2_loop"]
                by, _ = counters_for(pr2, "preprocess_backward_kernel<3,true,true>")
                if by is not None and c2["backward_with_adam"]["avg_us"]:
                    c2["backward_with_adam"]["pmc_bytes"] = by
                    c2["backward_with_adam"]["frac_hbm_pmc"] = round(by / (c2["backward_with_adam"]["avg_us"] * 1e-6) / HBM_PEAK, 4)
                npx = W * H
                for key, pat, alg in (("loss_stats", "loss_stats_kernel", npx * 36), ("loss_grad", "loss_grad_kernel", npx * 52)):
                    by, us = counters_for(pr2, pat)
                    if us:
                        c2[key] = {"kernel": pat, "avg_us": us, "alg_bytes": int(alg), "frac_hbm": round(alg / (us * 1e-6) / HBM_PEAK, 4), "pmc_bytes": by,
                                   "frac_hbm_pmc": round(by / (us * 1e-6) / HBM_PEAK, 4) if by else None,
                                   "note": "per pixel: colour 12 + target 12 + depth 4 + target depth 4 + depth^2 4 B read" + ("" if key == "loss_stats" else "; dL/dcolour 12 + dL/ddepth 4 B written")
                                           + "; duration from the rocprofv3 kernel table of the same loop"}
                ev = {}
                for pat in ("densify_classify_kernel", "compact3_count_kernel", "compact3_write_kernel", "gather_rows_vec4_kernel", "gather_rows_kernel",
                            "densify_children_kernel", "adam_multi_kernel"):
                    by, us = counters_for(pr2, pat)
                    if us:
                        ev[pat] = {"avg_us": us, "pmc_bytes": by, "frac_hbm_pmc": round(by / (us * 1e-6) / HBM_PEAK, 4) if by else None,
                                   "calls": next(e["calls"] for k, e in pr2["kernels"].items() if pat in k)}
                c2["densify_event"]["kernels"] = ev
                c2["rocprof"] = pr2
            torch.cuda.empty_cache()
        except Exception as e:
            out["configs2_loop"] = {"error": str(e)}
        # ---- configs[3]'s 64-keyframe optimiser step on this one GPU (the 1-GPU point of the driver's scaling curve) ----
        try:
            note("configs[3] on one GPU")
            out["configs3_single_gpu"] = run_c4(args, dev, 0, 1)
            torch.cuda.empty_cache()
            pr3 = profiled_kernels("c3step")
            if pr3:
                n3 = args.c4_gaussians
                rows = {}
                for k, e in pr3["kernels"].items():
                    if k.startswith("rows_kernel<"):
                        wide = "64,64" in k
                        G = 59 if wide else 14
                        mode = k.split("<")[1].split(",")[0]
                        # pack: gradients in, flat out; adam (one rank: ALL rows): flat in, p / m / v in and out, flat out; unpack: flat in, parameters out
                        alg = n3 * G * 4 * {"0": 2, "1": 2, "2": 8}.get(mode, 2)
                        rows[k] = dict(e, exchange_floats_per_gaussian=G, role={"0": "pack", "1": "unpack", "2": "adam on the row block"}.get(mode),
                                       alg_bytes=int(alg), frac_hbm=round(alg / (e["avg_us"] * 1e-6) / HBM_PEAK, 4))
                out["configs3_single_gpu"]["exchange_kernels"] = rows
                out["configs3_single_gpu"]["rccl_kernels"] = {k: e for k, e in pr3["kernels"].items() if "nccl" in k.lower() or "rccl" in k.lower()}
                out["configs3_single_gpu"]["rocprof"] = pr3
            if args.c4_sh_degree < 0:
                note("configs[3] on one GPU, SH-3 map (G = 59)")
                out["configs3_single_gpu"]["sh3_map_G59"] = run_c4(args, dev, 0, 1, sh_degree=3, light=True)
                torch.cuda.empty_cache()
        except Exception as e:
            out["configs3_single_gpu"] = {"error": str(e)}
        note("mapping-iteration leg")
        # ---- informational leg: one full ActiveSplat mapping iteration (get_loss + backward + Adam) on configs[1]'s scene,
        # with the reference's call pattern (two raster passes, torch loss, torch activations) vs this build's fused paths
        try:
            its = {}
            N1 = 500_000
            params1 = params1 if params1 is not None else syn.make_params(N1, W, H, seed=0)
            for name, flags in (("reference_call_pattern", {}), ("fused", dict(fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=True)),
                                ("fused_adam_in_backward", dict(fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=True, adam=True)),
                                ("fused_without_autograd", dict(fused=True, direct=True))):
                wall, ev = mapping_iteration_ms(dev, params1, W, H, flags)
                its[name + "_ms"] = round(wall, 4)
                its[name + "_gpu_ms"] = round(ev, 4)
            out["mapping_iteration"] = dict(its, note="get_loss (RGB + depth/silhouette render, L1+SSIM+depth loss) + backward + Adam on configs[1]'s "
                                            "scene; reference_call_pattern = splatam.py:172-301 op for op on this rasteriser")
        except Exception as e:
            out["mapping_iteration"] = {"error": str(e)}
    note("done")
    emit(out)


if __name__ == "__main__":
    main()
