"""Adam kernel bandwidth on one big tensor (the [2M,16,3] SH coefficients of configs[2]).  GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from activesplat_amd import optim as O  # noqa: E402
dev = torch.device("cuda")
n = int(os.environ.get("ELEMS", 96_000_000))
p = torch.nn.Parameter(torch.randn(n, device=dev))
opt = O.GaussianAdam([{"params": [p], "name": "x", "lr": 1e-3}], lr=0.0, eps=1e-15)
p.grad = torch.randn(n, device=dev)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20)
best = min(ts)
print(f"adam {n/1e6:.0f}M elements: {best*1e6:.1f} us/step  {n*28/best/1e12:.2f} TB/s (28 B/element)")
