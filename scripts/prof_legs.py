"""The workloads behind the non-headline rocprofv3 passes of scripts/gpu_round_end.sh (kernel tables and PMC counters for the legs
that only had hipEvent stage times).  One process, one leg, few launches -- so that a counter pass stays short:

  python scripts/prof_legs.py c1       BASELINE configs[1]: 500 k Gaussians, SH-0, 640x480, forward + backward frames
  python scripts/prof_legs.py c2loop   configs[2]'s optimise loop: 2 M / SH-3, 20 iterations, densify every 10 (one event inside; Adam in the backward kernel
                                       on the other iterations, adam_multi_kernel on the event iteration)
  python scripts/prof_legs.py c3step   configs[3] on one GPU: two optimiser steps over 8 keyframes each (raw-parameter RGB-D render -> fused loss -> backward
                                       with in-kernel .grad accumulation), each closed by the ONE-rank RCCL exchange (pack -> reduce_scatter_tensor ->
                                       gs_adam_rows -> all_gather_into_tensor -> unpack); G = 14, then the same on the SH-3 map (G = 59)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
leg = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
W, H = 640, 480

if leg == "c1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    wl = bench.RenderWorkload(500_000, W, H, dev, sh_degree=None)
    for _ in range(int(os.environ.get("STEPS", 12))):
        wl.step()
    torch.cuda.synchronize()
elif leg == "c2loop":
    from activesplat_amd.workloads import configs2_optimise_loop
    configs2_optimise_loop(4096, 12, "cuda", densify_every=5)
    torch.manual_seed(0)
    r = configs2_optimise_loop(2_000_000, int(os.environ.get("ITERS", 20)), "cuda", densify_every=10, time_it=True)
    print("c2loop:", {k: r[k] for k in ("seconds", "counts", "losses")}, file=sys.stderr)
elif leg == "c3step":
    import socket
    import torch.distributed as dist
    from activesplat_amd import mapping as M, optim as O, parallel as PL, setup_camera
    from activesplat_amd import synthetic as syn
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    N, KF = 2_000_000, int(os.environ.get("KF", 8))
    for sh in (False, True):
        raw = syn.shell_scene(N, seed=0, W=W, H=H)
        lrs = dict(means3D=1e-4, rgb_colors=2.5e-3, unnorm_rotations=1e-3, logit_opacities=0.05, log_scales=1e-3, cam_unnorm_rots=0.0, cam_trans=0.0)
        if sh:
            g = torch.Generator().manual_seed(7)
            shs = 0.2 * torch.randn(N, 16, 3, generator=g)
            shs[:, 0, :] = (raw.pop("rgb_colors") - 0.5) / 0.28209479177387814
            raw["shs"] = shs
            lrs = {("shs" if k == "rgb_colors" else k): v for k, v in lrs.items()}
        prm = {k: torch.nn.Parameter(v.to(dev)) for k, v in raw.items()}
        prm["cam_unnorm_rots"] = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0], device=dev).reshape(1, 4, 1))
        prm["cam_trans"] = torch.nn.Parameter(torch.zeros(1, 3, 1, device=dev))
        var = {k: torch.zeros(N, device=dev) for k in ("max_2D_radius", "means2D_gradient_accum", "denom", "timestep")}
        opt = O.initialize_optimizer(prm, lrs)
        cam = setup_camera(W, H, syn.intrinsics(W, H), np.eye(4), device=dev, sh_degree=3 if sh else 0)
        kfs = []
        for i in range(KF):
            im, depth = syn.make_targets(W, H, seed=100 + i)
            kfs.append(dict(id=i, cam=cam, w2c=torch.eye(4, device=dev), im=im.to(dev), depth=depth.to(dev),
                            pose7=[float(v) for v in syn.quat_from_yaw(2 * np.pi * i / 64)] + [0.0, 0.0, 0.0]))

        def loss_fn(p, kf, v):
            loss, v, _ = M.get_loss(p, kf, v, 0, dict(im=0.5, depth=1.0), fused=True, fused_loss=True, fused_inputs=True, fused_preprocess=True,
                                    pose7=kf["pose7"], accumulate_grads=True)
            return loss, v
        for step in range(int(os.environ.get("STEPS", 3))):
            # world = 1 would skip the collectives: run the rank's keyframes (rank 0 of 1), then the exchange the 8-rank step runs
            opt.zero_grad(set_to_none=True)
            for kf in kfs:
                loss, var = loss_fn(prm, kf, var)
                loss.backward(M.unit_gradient(loss))
            PL.reduce_scatter_adam_step(prm, opt, timing=True)
        torch.cuda.synchronize()
        print(f"c3step G={59 if sh else 14}: exchange + Adam {PL.exchange_ms():.3f} ms, collectives {PL.last_exchange.get('reduce')}, {PL.last_exchange.get('gather')}", file=sys.stderr)
        del prm, opt, var, kfs
        torch.cuda.empty_cache()
    dist.destroy_process_group()
else:
    raise SystemExit(__doc__)
