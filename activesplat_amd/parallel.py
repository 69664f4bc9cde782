"""Keyframe sharding across the GPUs of one node -- a NEW capability: the reference
is single-GPU and optimises one random keyframe per Adam step (src/mapper/splatam/__init__.py:450-480).

One process per GPU (torchrun), Gaussian parameters + Adam state replicated, keyframes block-partitioned
(64 keyframes / 8 GPUs = 8 contiguous keyframes per rank).  Every rank renders its keyframes, lets autograd
accumulate the per-Gaussian gradients locally, and ONE all-reduce(sum) of a flat fp32 [N, 14] buffer
(means3D 3 + rgb 3 + rotations 4 + opacity 1 + scales 3 -> 112 MB at N = 2M) over RCCL/xGMI precedes the
Adam step, so every rank applies the identical update.  Parity target: reduced gradient == the sum of the
per-keyframe gradients a single GPU computes sequentially (each keyframe loss is already a per-image mean).
Densification statistics are combined the same way (sum, sum, max).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

GRAD_KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


def shard_keyframes(num_keyframes: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first (num_keyframes % world) ranks take one extra keyframe."""
    base, extra = divmod(num_keyframes, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class FlatGradBuffer:
    """One contiguous [N, G] fp32 buffer that the per-key gradients are packed into for a single collective."""

    def __init__(self, params, keys=GRAD_KEYS):
        self.keys = [k for k in keys if k in params]
        self.widths = [params[k].shape[1] for k in self.keys]
        n = params[self.keys[0]].shape[0]
        self.flat = torch.zeros(n, sum(self.widths), dtype=torch.float32, device=params[self.keys[0]].device)

    def pack(self, params):
        col = 0
        for k, w in zip(self.keys, self.widths):
            g = params[k].grad
            self.flat[:, col:col + w] = 0 if g is None else g
            col += w
        return self.flat

    def unpack(self, params):
        col = 0
        for k, w in zip(self.keys, self.widths):
            params[k].grad = self.flat[:, col:col + w].contiguous()
            col += w


def all_reduce_gradients(params, buf: FlatGradBuffer | None = None, group=None):
    """Sum the locally accumulated .grad of every per-Gaussian tensor over all ranks (one all-reduce)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf
    buf = FlatGradBuffer(params) if buf is None else buf
    dist.all_reduce(buf.pack(params), op=dist.ReduceOp.SUM, group=group)
    buf.unpack(params)
    return buf


def all_reduce_statistics(variables, group=None):
    """Densification statistics live per rank; combine them before densify (sum, sum, max)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return variables
    for k, op in (("means2D_gradient_accum", dist.ReduceOp.SUM), ("denom", dist.ReduceOp.SUM), ("max_2D_radius", dist.ReduceOp.MAX)):
        if k in variables:
            dist.all_reduce(variables[k], op=op, group=group)
    return variables


def sharded_keyframe_step(params, variables, keyframes, optimizer, loss_fn, rank=None, world=None, buf=None):
    """One optimiser step over a batch of keyframes sharded across ranks.
    loss_fn(params, keyframe, variables) -> (loss, variables).  Returns the local loss sum."""
    on = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank() if on else 0) if rank is None else rank
    world = (dist.get_world_size() if on else 1) if world is None else world
    optimizer.zero_grad(set_to_none=True)
    total = 0.0
    for i in shard_keyframes(len(keyframes), rank, world):
        loss, variables = loss_fn(params, keyframes[i], variables)
        loss.backward()                       # autograd accumulates into .grad across this rank's keyframes
        total += float(loss.detach())
    buf = all_reduce_gradients(params, buf)
    optimizer.step()
    return total, variables, buf
