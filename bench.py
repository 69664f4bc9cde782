#!/usr/bin/env python
"""bench.py -- headline benchmark of the rasteriser hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: one "step" = one forward (colour + radii + depth + opacity) + one backward (all input gradients incl.
means2D) of BASELINE.json configs[1]: 500k synthetic Gaussians, one 640x480 view, SH degree 0
(colors_precomp), inputs and dL/dcolor already resident in HBM.
N > 1: BASELINE.json configs[3] -- 2 M Gaussians replicated on every rank, a batch of 64 keyframes at 640x480
block-partitioned over the ranks (64/N each); one "step" = one optimiser step over the batch: every rank renders its
keyframes (fused activations -> single-pass RGB-D render -> fused loss -> backward), the flat [N,14] fp32 gradient is
reduce-scattered over RCCL, each rank runs the fused Adam on its 1/N row block and the updated rows are all-gathered
(activesplat_amd/parallel.py).  `value` = keyframes rendered+back-propagated per second by the whole job (strong scaling:
the batch is fixed); rank 0 also times the same 64-keyframe step alone (`single_gpu_same_workload_fps`).
Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only, after the timed region):
  roofline     : per-stage hipEvent timing through the C ABI's gs_profile_* hooks over a second pass of
                 the same K steps; the dominant stage's algorithmic bytes / its
                 average duration against the 8 TB/s HBM peak.
  cpu_baseline : the C oracle (oracle/gs_oracle.c, 1 host core) on the same workload, 1 frame.
"""
import argparse
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


_T0 = time.perf_counter()


def note(msg):
    """progress line on stderr (the JSON line on stdout stays the only stdout output)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def stage_bytes(P, D, npix):
    """Algorithmic HBM bytes per launch of each stage: the per-unit figures of SURVEY.md section 8d
    (b_g = 292 B/Gaussian, b_i = 160 B/instance, b_p = 48 B/pixel) split by stage."""
    return {
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


def run_c4(args, dev, rank, world):
    """BASELINE configs[3]: 64 keyframes sharded over the ranks, RCCL gradient exchange, sharded fused Adam."""
    import torch.distributed as dist
    from activesplat_amd import mapping as M, optim as O, parallel as PL, setup_camera
    from activesplat_amd import rasterizer as R
    from activesplat_amd import synthetic as syn
    W, H, N, KF = args.width, args.height, args.c4_gaussians, args.keyframes
    K = syn.intrinsics(W, H)
    raw = syn.shell_scene(N, seed=0, W=W, H=H)
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in raw.items()}
    params["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
    params["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
    lrs = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
    opt = O.initialize_optimizer(params, lrs)
    variables = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
    cam = setup_camera(W, H, K, np.eye(4), device=dev)
    mine = set(PL.shard_keyframes(KF, rank, world)) if world > 1 else set(range(KF))
    keyframes = []
    for i in range(KF):
        a = 2 * np.pi * i / KF
        kf = dict(id=i, cam=cam, w2c=torch.eye(4, device=dev), pose7=[float(v) for v in syn.quat_from_yaw(a)] + [0.0, 0.0, 0.0])
        if i in mine or rank == 0:                             # rank 0 also runs the whole batch alone afterwards
            im, depth = syn.make_targets(W, H, seed=100 + i)
            kf.update(im=im.to(dev), depth=depth.to(dev))
        keyframes.append(kf)
    weights = dict(im=0.5, depth=1.0)

    def loss_fn(p, kf, v):
        loss, v, _ = M.get_loss(p, kf, v, 0, weights, fused=True, fused_loss=True, fused_inputs=True, pose7=kf["pose7"])
        return loss, v

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state = dict(v=variables)

    def step(r=None, w=None):
        _, state["v"], _ = PL.sharded_keyframe_step(params, state["v"], keyframes, opt, loss_fn, rank=r, world=w,
                                                    sharded_adam=True, streams=args.streams)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    D = int(R.last_stats["num_rendered"])
    out = {
        "metric": "render+backward frames/sec at 640x480, N Gaussians", "value": round(KF * args.steps / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {N} Gaussians, {KF} keyframes at {W}x{H} block-partitioned over {world} ranks "
                               f"({KF // max(world, 1)} per rank), one optimiser step per batch, RCCL reduce-scatter of the [N,14] fp32 gradient "
                               "-> sharded fused Adam -> all-gather of the updated rows",
                   "gaussians": N, "width": W, "height": H, "keyframes_per_step": KF, "keyframes_per_rank_per_step": KF // max(world, 1),
                   "tile_instances_D_last_keyframe": D, "streams": args.streams, "grad_exchange": "reduce_scatter+all_gather, 112 MB at 2M",
                   "loss": "fused mapping loss (L1 + SSIM + masked depth) through the single-pass RGB-D render",
                   "parallelism": f"keyframe-sharded x{world}"},
    }
    # the same batch on ONE GPU (rank 0 alone, no collectives): the reference point of the strong-scaling curve
    ref = None
    if rank == 0 and not args.no_extras:
        n_ref = max(1, min(3, args.steps))
        step(0, 1)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(n_ref):
            step(0, 1)
        torch.cuda.synchronize()
        ref = KF * n_ref / (time.perf_counter() - t1)
        out["single_gpu_same_workload_fps"] = round(ref, 2)
        out["speedup_vs_single_gpu_same_workload"] = round(out["value"] / ref, 3)
    barrier()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--accum", type=int, default=8, help="keyframes accumulated per rank between gradient all-reduces")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the independent steps are issued on in turn (1 = strictly "
                    "one frame after the other)")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline and cpu_baseline legs")
    ap.add_argument("--cpu-threads", type=int, default=64, help="upper bound on the host threads of the cpu_baseline leg")
    ap.add_argument("--c4-gaussians", type=int, default=2_000_000, help="N > 1 (configs[3]): Gaussians of the replicated map")
    ap.add_argument("--keyframes", type=int, default=64, help="N > 1 (configs[3]): keyframes per optimiser step, sharded over the ranks")
    ap.add_argument("--workload", choices=("auto", "c2", "c4"), default="auto", help="auto: configs[1] on one GPU, configs[3] on several")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    dist_on = world > 1
    # development knobs for exercising the N > 1 code path on a 1-GPU box: all ranks on device 0, gloo instead of RCCL
    same_dev = bool(os.environ.get("BENCH_SAME_DEVICE"))
    torch.cuda.set_device(0 if same_dev else local_rank)
    dev = torch.device("cuda", 0 if same_dev else local_rank)
    if dist_on:
        import torch.distributed as dist
        if os.environ.get("BENCH_BACKEND") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # "nccl" is RCCL on ROCm

    from activesplat_amd import GaussianRasterizer, _lib, setup_camera
    from activesplat_amd import rasterizer as R
    from activesplat_amd import synthetic as syn
    _lib.get()                                              # fail loudly if the HIP library is missing
    if args.workload == "c4" or (args.workload == "auto" and dist_on):
        return run_c4(args, dev, rank, world)

    W, H, N = args.width, args.height, args.gaussians
    K = syn.intrinsics(W, H)
    # rank r looks at the replicated scene from a slightly different yaw (its own keyframe)
    yaw = np.deg2rad(2.0) * (rank - (world - 1) / 2.0)
    c, s = np.cos(yaw), np.sin(yaw)
    w2c = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)
    cam = setup_camera(W, H, K, w2c, device=dev)
    params = syn.make_params(N, W, H, seed=0)
    rv = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(params).items()}
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    keys = ["means3D", "colors_precomp", "rotations", "opacities", "scales"]
    # N > 1: per-key gradient accumulators (contiguous adds), packed into ONE flat [N,14] buffer per all-reduce
    acc = [torch.zeros_like(rv[k]) for k in keys] if dist_on else None

    def reduce_gradients():
        flat = torch.cat(acc, dim=1)                        # 3+3+4+1+3 columns: one collective instead of five
        dist.all_reduce(flat)
        for a in acc:
            a.zero_()
        return flat

    def step(i):
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        color, radii, depth, opacity = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)
        grads = torch.autograd.grad(color, [rv[k] for k in keys] + [m2d], dL)
        if dist_on:
            torch._foreach_add_(acc, list(grads[:5]))
            if (i + 1) % args.accum == 0:
                reduce_gradients()
        return grads

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The K steps are independent frames (keyframes of a batch: nothing of step i feeds step i+1), so they are issued on
    # `--streams` HIP streams in turn: one frame's kernels fill the tails and the workgroup-placement imbalance of the other's.
    # Every step still runs the complete forward + backward; --streams 1 serialises them (reported as sequential_fps).
    pool = [torch.cuda.Stream(device=dev) for _ in range(max(args.streams, 1))] if args.streams > 1 else None

    def run(n_steps, first=0):
        if pool is None:
            for i in range(first, first + n_steps):
                step(i)
            return
        main = torch.cuda.current_stream(dev)
        for s_ in pool:
            s_.wait_stream(main)
        for i in range(first, first + n_steps):
            with torch.cuda.stream(pool[i % len(pool)]):
                step_on_stream(i)
        for s_ in pool:
            main.wait_stream(s_)

    def exchange(lane):
        """Join the streams, fold every stream's accumulators into `lane`'s, one packed all-reduce, release the streams."""
        cur = torch.cuda.current_stream(dev)
        for s_ in pool:
            cur.wait_stream(s_)
        for other in range(len(pool)):
            if other != lane:
                torch._foreach_add_(acc_s[lane], acc_s[other])
                for a in acc_s[other]:
                    a.zero_()
        flat = torch.cat(acc_s[lane], dim=1)
        dist.all_reduce(flat)
        for a in acc_s[lane]:
            a.zero_()
        for s_ in pool:
            s_.wait_stream(cur)

    def step_on_stream(i):
        if not dist_on:
            return step(i)
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        color = GaussianRasterizer(raster_settings=cam)(means2D=m2d, **rv)[0]
        grads = torch.autograd.grad(color, [rv[k] for k in keys] + [m2d], dL)
        lane = i % len(pool)
        torch._foreach_add_(acc_s[lane], list(grads[:5]))
        if (i + 1) % args.accum == 0:
            exchange(lane)                  # a batch of `accum` keyframes is complete on this rank
        return grads

    acc_s = [[torch.zeros_like(rv[k]) for k in keys] for _ in range(len(pool))] if (pool is not None and dist_on) else None
    # untimed preparation in front of the W warm-up steps: every stream has its own pool in torch's caching allocator, and a stream's
    # first few frames still reach hipMalloc (with --warmup 5 on two streams that spilled into the timed region)
    run(4 * max(args.streams, 1))
    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    if dist_on and args.steps % args.accum:
        if pool is None:
            reduce_gradients()
        else:
            with torch.cuda.stream(pool[0]):
                exchange(0)
            torch.cuda.current_stream(dev).wait_stream(pool[0])
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    D = R.last_stats["num_rendered"]
    ms_per_step = dt / args.steps * 1e3
    fps = world * args.steps / dt
    out = {
        "metric": "render+backward frames/sec at 640x480, N Gaussians", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 500k Gaussians, 640x480 RGB+depth forward+backward, SH degree 0",
                   "gaussians": N, "width": W, "height": H, "tile_instances_D": int(D),
                   "keyframes_per_rank_per_step": 1, "streams": args.streams, "grad_allreduce_every": args.accum if dist_on else None,
                   "parallelism": f"keyframe-sharded x{world}" if dist_on else "single GPU"},
    }

    if rank == 0 and world == 1 and not args.no_extras:
        lib = _lib.get()
        note(f"timed region done: {fps:.1f} frames/s; roofline leg")
        # ---- roofline leg: per-stage hipEvents over a second pass of the same K steps ----
        lib.gs_profile_enable(1)
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        lib.gs_profile_enable(0)
        sb = stage_bytes(N, D, W * H)
        stages = {k: {"avg_us": round(ms / max(c, 1) * 1e3, 2), "calls": c, "alg_bytes": sb.get(k),
                      "frac_hbm": round(sb[k] / (ms / c * 1e-3) / HBM_PEAK, 4) if c and k in sb and ms > 0 else None}
                  for k, (ms, c) in prof.items() if c}
        dom = max((k for k in stages if k in sb), key=lambda k: stages[k]["avg_us"])
        ach = sb[dom] / (stages[dom]["avg_us"] * 1e-6)
        frame_bytes = N * 292 + D * 160 + W * H * 48
        # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process; the most recent
        # separate-pass collection (scripts/gpu_round_end.sh -> profiles/*_pmc_summary.json) is reported if present
        traffic, traffic_src, valu = None, None, None
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
            if files and N == 500_000 and (W, H) == (640, 480):
                kern = {"blend_backward": "blend_backward_kernel", "blend_forward": "blend_forward_streams_kernel"}.get(dom)
                pm = json.load(open(files[-1]))
                if kern in pm and "traffic_bytes" in pm[kern]:
                    traffic, traffic_src = int(pm[kern]["traffic_bytes"]), os.path.basename(files[-1])
                if kern in pm and "SQ_INSTS_VALU" in pm[kern]:
                    # the bound that actually limits the blend kernels: vector-ALU issue slots (measured peak, see VALU_ISSUE_PEAK)
                    vi = float(pm[kern]["SQ_INSTS_VALU"])
                    valu = {"wave_insts_per_launch": int(vi), "achieved_ginst_s": round(vi / (stages[dom]["avg_us"] * 1e-6) / 1e9, 1),
                            "peak_ginst_s": VALU_ISSUE_PEAK / 1e9, "frac": round(vi / (stages[dom]["avg_us"] * 1e-6) / VALU_ISSUE_PEAK, 4),
                            "source": os.path.basename(files[-1])}
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach / 1e9, 2), "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "frac": round(ach / HBM_PEAK, 5), "traffic": traffic, "traffic_source": traffic_src,
                           "alg_bytes_per_launch": sb[dom], "avg_us": stages[dom]["avg_us"],
                           "valu_issue": valu, "frame_alg_bytes": frame_bytes,
                           "frame_frac": round(frame_bytes / (ms_per_step * 1e-3) / HBM_PEAK, 5),
                           "frame_frac_note": f"`value` and frame_frac are the rate of independent frames issued on {args.streams} HIP streams in turn "
                                              "(keyframes of a batch); frame_frac_sequential is one frame strictly after the other (the mapper's "
                                              "one-keyframe-per-Adam-step loop)",
                           "stages": stages}
        # ---- secondary figures SURVEY 8(d) asks for: step-time spread, forward-only rate, blend flop rate ----
        try:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for i, (a, b) in enumerate(ev):
                a.record(); step(i); b.record()
            torch.cuda.synchronize()
            ts = np.sort(np.array([a.elapsed_time(b) for a, b in ev]))
            out["step_ms_percentiles"] = {"p10": round(float(ts[len(ts) // 10]), 4), "p50": round(float(ts[len(ts) // 2]), 4),
                                          "p90": round(float(ts[(len(ts) * 9) // 10]), 4)}
            # strictly sequential frames (the mapper's one-keyframe-per-Adam-step loop cannot overlap iterations)
            for i in range(args.warmup):
                step(i)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize()
            out["sequential_fps"] = round(args.steps / (time.perf_counter() - t1), 1)
            out["roofline"]["frame_frac_sequential"] = round(frame_bytes * out["sequential_fps"] / HBM_PEAK, 5)
            rv_ng = {k: v.detach() for k, v in rv.items()}
            m2d0 = torch.zeros(N, 3, device=dev)
            with torch.no_grad():
                for _ in range(args.warmup):
                    GaussianRasterizer(raster_settings=cam)(means2D=m2d0, **rv_ng)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(args.steps):
                    GaussianRasterizer(raster_settings=cam)(means2D=m2d0, **rv_ng)
                torch.cuda.synchronize()
                out["forward_only_fps"] = round(args.steps / (time.perf_counter() - t1), 1)
                GaussianRasterizer(raster_settings=cam._replace(debug=True))(means2D=m2d0, **rv_ng)
            il = R.last_debug["il"]
            ncontrib = R.last_debug["image"][il.n_contrib:il.n_contrib + 4 * W * H].view(torch.int32)
            E = int(ncontrib.to(torch.int64).sum().item())          # list entries walked, summed over pixels
            tf, tb = stages["blend_forward"]["avg_us"] * 1e-6, stages["blend_backward"]["avg_us"] * 1e-6
            out["blend_flops"] = {"evaluations_E": E, "forward_frac_fp32_peak": round(E * 12 / tf / FP32_PEAK, 4),
                                  "backward_frac_fp32_peak": round(E * 40 / tb / FP32_PEAK, 4), "peak_tflops": FP32_PEAK / 1e12,
                                  "note": "SURVEY 8(d): F_alg = E*(12 fwd + 40 bwd) flop, E = sum of n_contrib"}
        except Exception as e:
            out["secondary_error"] = str(e)
        # ---- cpu_baseline leg: the C oracle (a port of the path, OpenMP over the pixel rows) on ALL host cores, same workload ----
        try:
            from oracle.gs_oracle import Oracle
            from tests import util
            o = Oracle("f32")
            cores = min(os.cpu_count() or 1, args.cpu_threads)
            o.set_threads(cores)
            note(f"cpu_baseline: oracle on {cores} threads")
            rv_cpu = {k: v.detach().cpu() for k, v in rv.items()}
            f = util.run_oracle(o, cam, rv_cpu, dL.cpu())          # warm-up frame (page-in, thread pool)
            t1 = time.perf_counter(); n_cpu = 0
            while n_cpu < 50 and (n_cpu == 0 or time.perf_counter() - t1 < CPU_SECONDS):
                f = util.run_oracle(o, cam, rv_cpu, dL.cpu()); n_cpu += 1
            tc = (time.perf_counter() - t1) / n_cpu
            o.set_threads(1)
            try:
                # "PSNR vs ref" half of the metric: this device's render and gradients against the oracle's, same inputs
                with torch.no_grad():
                    col_gpu = GaussianRasterizer(raster_settings=cam)(means2D=torch.zeros(N, 3, device=dev), **{k: v.detach() for k, v in rv.items()})[0]
                mse = float(((col_gpu.cpu().double() - torch.from_numpy(np.asarray(f["color"])).double()) ** 2).mean())
                g_gpu = step(0)
                rel = {}
                for k, g in zip(keys, g_gpu[:5]):
                    ref = torch.from_numpy(np.asarray(f["grads"][k])).double().reshape(g.shape)
                    rel[k] = float((g.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30))
                out["parity_vs_oracle"] = {"psnr_color_db": round(10 * np.log10(1.0 / max(mse, 1e-30)), 1),
                                           "grad_rel_l2_max": float(f"{max(rel.values()):.3g}"), "oracle": "oracle/gs_oracle.c fp32 build, same inputs"}
            except Exception as e:
                out["parity_vs_oracle"] = {"error": str(e)}
            out["cpu_baseline"] = {"value": round(1.0 / tc, 5), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} frames forward+backward of the same workload (N={N}, {W}x{H}, D={f['D']}) "
                                             f"by oracle/gs_oracle.c (fp32, gcc -O2 -fopenmp, {cores} threads over pixel rows / Gaussians; "
                                             f"preprocess and the key sort are single-threaded), {tc:.2f} s per frame"}
        except Exception as e:      # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        # ---- BASELINE configs[0]: 10k Gaussians, one 640x480 view, CPU PyTorch forward-only render (plumbing check), next to the HIP forward ----
        try:
            from oracle import dense_torch as DT
            from tests import util
            rs0, rv0 = util.scene(10_000, W, H, seed=0)
            cam0 = util.cam_dict(rs0)
            note("configs[0]: tiled CPU PyTorch render")
            th0 = torch.get_num_threads()
            torch.set_num_threads(min(os.cpu_count() or 1, 16))       # tile-sized tensors: more threads only add fork/join cost
            with torch.no_grad():
                t1 = time.perf_counter()
                ref0 = DT.render_dense(cam0, rv0["means3D"], rv0["opacities"], colors=rv0["colors_precomp"], scales=rv0["scales"],
                                       rotations=rv0["rotations"], tiled=True)
                t_cpu = time.perf_counter() - t1
                rs0d = setup_camera(W, H, K, np.eye(4), device=dev)
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
        except Exception as e:
            out["configs0"] = {"error": str(e)}
        try:
            torch.set_num_threads(th0)
        except Exception:
            pass
    if rank == 0 and world == 1 and not args.no_extras:
        note("north-star 2M legs")
        # ---- the north-star's target configuration: forward+backward render of 2 M Gaussians at 640x480 (SH degree 0 and 3),
        # as frames/s and as fraction of the HBM roofline with SURVEY 8(d)'s algorithmic bytes (b_g = 292 / 832 B)
        try:
            ns = {}
            for name, deg, bg_bytes in (("sh0", None, 292), ("sh3", 3, 832)):
                N2 = 2_000_000
                p2 = syn.make_params(N2, W, H, seed=0, sh_degree=deg)
                rv2 = {k: v.to(dev).requires_grad_(True) for k, v in syn.activate(p2).items()}
                cam2 = setup_camera(W, H, K, np.eye(4), device=dev, sh_degree=deg or 0)
                keys2 = list(rv2.keys())

                def step2():
                    m2 = torch.zeros(N2, 3, device=dev, requires_grad=True)
                    col = GaussianRasterizer(raster_settings=cam2)(means2D=m2, **rv2)[0]
                    return torch.autograd.grad(col, [rv2[k] for k in keys2] + [m2], dL)
                for _ in range(5):
                    step2()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(30):
                    step2()
                torch.cuda.synchronize()
                t = (time.perf_counter() - t1) / 30
                D2 = int(R.last_stats["num_rendered"])
                fb = N2 * bg_bytes + D2 * 160 + W * H * 48
                ns[name] = {"frames_per_s": round(1.0 / t, 1), "ms_per_frame": round(t * 1e3, 4), "tile_instances_D": D2,
                            "frame_alg_bytes": fb, "frame_frac_hbm": round(fb / t / HBM_PEAK, 4)}
                # the same frames as a keyframe batch on two streams (what `value` does for configs[1])
                pool2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
                torch.cuda.synchronize()
                for i in range(4):
                    with torch.cuda.stream(pool2[i % 2]):
                        step2()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for i in range(30):
                    with torch.cuda.stream(pool2[i % 2]):
                        step2()
                torch.cuda.synchronize()
                t2 = (time.perf_counter() - t1) / 30
                ns[name].update(two_stream_frames_per_s=round(1.0 / t2, 1), two_stream_frame_frac_hbm=round(fb / t2 / HBM_PEAK, 4))
                # per-stage hipEvent averages of the sequential frames, against each stage's algorithmic bytes
                lib.gs_profile_enable(1)
                for _ in range(10):
                    step2()
                torch.cuda.synchronize()
                prof2 = _lib.profile_collect()
                lib.gs_profile_enable(0)
                sb2 = stage_bytes(N2, D2, W * H)
                if deg:                                        # SH-3: 192 B of coefficients read forward, read + 192 B written backward
                    sb2["preprocess_forward+scan"] += N2 * 180  # colours (12 B) replaced by coefficient rows (192 B)
                    sb2["preprocess_backward"] += N2 * (180 + 180)
                ns[name]["stages"] = {k: {"avg_us": round(ms / c * 1e3, 1), "alg_bytes": sb2.get(k),
                                          "frac_hbm": round(sb2[k] / (ms / c * 1e-3) / HBM_PEAK, 4) if k in sb2 else None}
                                      for k, (ms, c) in prof2.items() if c}
                del rv2, p2
            try:
                import glob
                f2 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_2m_pmc_summary.json")))
                if f2:
                    pm2 = json.load(open(f2[-1]))
                    ns["pmc_traffic_bytes_sh3"] = {k: int(v["traffic_bytes"]) for k, v in pm2.items() if "traffic_bytes" in v}
                    ns["pmc_source"] = os.path.basename(f2[-1])
            except Exception:
                pass
            ns["target"] = "north star: >= 40 % of the 8 TB/s HBM roofline on the forward+backward render of 2M Gaussians (frame_frac_hbm)"
            out["north_star_2M"] = ns
        except Exception as e:
            out["north_star_2M"] = {"error": str(e)}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_extras:
        note("mapping-iteration leg")
        # ---- informational leg: one full ActiveSplat mapping iteration (get_loss + backward + Adam) on the same scene,
        # with the reference's call pattern (two raster passes, torch loss, torch activations) vs this build's fused paths
        try:
            from activesplat_amd import mapping as M, optim as O
            its = {}
            for name, flags in (("reference_call_pattern", {}), ("fused", dict(fused=True, fused_loss=True, fused_inputs=True))):
                prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
                prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
                prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
                var = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
                tim, tdepth = syn.make_targets(W, H)
                data = dict(cam=cam, im=tim.to(dev), depth=tdepth.to(dev), id=0, w2c=torch.eye(4, device=dev))
                opt = O.initialize_optimizer(prm, dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05,
                                                       log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0))

                def it():
                    loss, _, _ = M.get_loss(prm, data, var, 0, dict(im=0.5, depth=1.0), pose7=[1.0, 0, 0, 0, 0, 0, 0] if flags else None, **flags)
                    loss.backward()
                    with torch.no_grad():
                        opt.step(); opt.zero_grad(set_to_none=True)
                for _ in range(5):
                    it()
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(20):
                    it()
                torch.cuda.synchronize()
                its[name + "_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 4)
            out["mapping_iteration"] = dict(its, note="get_loss (RGB + depth/silhouette render, L1+SSIM+depth loss) + backward + Adam on the "
                                            "bench scene; reference_call_pattern = splatam.py:172-301 op for op on this rasteriser")
        except Exception as e:
            out["mapping_iteration"] = {"error": str(e)}
    note("done")
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
