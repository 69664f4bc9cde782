"""Development helper: what plain element-wise kernels reach on this box's HBM (PyTorch's own copy / fill / sum / mul / add on 256 MB ... 4 GB tensors,
hipEvents): the yardstick for the per-Gaussian kernels' TB/s.  GPU box: python scripts/exp/hbm_bw_torch.py"""
import torch, time
dev='cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for mb in (256, 1024, 4096):
    n = mb*1024*1024//4
    a=torch.empty(n, device=dev); b=torch.empty(n, device=dev); a.normal_()
    tc=t(lambda: b.copy_(a)); tf=t(lambda: b.fill_(1.0)); ts=t(lambda: a.sum()); tm=t(lambda: torch.mul(a, 2.0, out=b)); ta=t(lambda: torch.add(a, b, out=b))
    print("%5d MB: copy %.2f TB/s, fill %.2f TB/s, sum(read) %.2f TB/s, mul(out) %.2f TB/s, add(2r+1w) %.2f TB/s" % (mb, 2*n*4/tc/1e12, n*4/tf/1e12, n*4/ts/1e12, 2*n*4/tm/1e12, 3*n*4/ta/1e12))
