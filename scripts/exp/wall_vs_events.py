"""Development helper: wall clock of K sequential frames against the sum of the per-frame event intervals (what the host adds)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
dev = torch.device("cuda")
wl = bench.RenderWorkload(2_000_000, 640, 480, dev, sh_degree=3)
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
import gc
for rep in range(5):
    if rep == 3:
        gc.disable()
    K = 120
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    host = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for a, b in ev:
        h0 = time.perf_counter()
        a.record(); wl.step(); b.record()
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e6
    evs = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    gaps = np.array([ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(K - 1)]) * 1e3
    host = np.array(host) * 1e6
    print(f"rep {rep} gc={'off' if rep >= 3 else 'on'} wall/frame {wall:.1f} us  events mean {evs.mean():.1f} p50 {np.median(evs):.1f} max {evs.max():.1f}  gaps mean {gaps.mean():.1f} max {gaps.max():.1f}  "
          f"host/step mean {host.mean():.1f} p50 {np.median(host):.1f} max {host.max():.1f}", flush=True)
