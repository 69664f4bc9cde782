import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from activesplat_amd import lookaround as LA, synthetic as syn
dev = torch.device("cuda")
params = {k: v.to(dev) for k, v in syn.shell_scene(1_000_000, seed=2, W=LA.LOOK_W, H=LA.LOOK_H).items()}
for _ in range(12):
    LA.look_around(params, np.eye(4), fused=True, batched=True)
torch.cuda.synchronize()
