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


def sharded_keyframe_step(params, variables, keyframes, optimizer, loss_fn, rank=None, world=None, buf=None,
                          sharded_adam=False, streams=1, densify_statistics=False):
    """One optimiser step over a batch of keyframes sharded across ranks.
    loss_fn(params, keyframe, variables) -> (loss, variables).  Returns the local loss sum.

    streams > 1 (GPU only): this rank's keyframes are independent until the optimiser step, so they are rendered on
    `streams` HIP streams in turn -- one frame's kernels fill the tails and the placement imbalance of the other's
    (two streams: 2404 -> 3062 frames/s on BASELINE configs[1], `two_stream_fps` in the bench line).  Each keyframe's
    gradients are taken with autograd.grad on its own stream and summed after the streams have joined.
    autograd.grad does not populate `means2D.grad`, which the densifier's statistics read (optim.accumulate_mean2d_gradient):
    a caller that densifies from this step's statistics passes densify_statistics=True and gets the serial walk."""
    on = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank() if on else 0) if rank is None else rank
    world = (dist.get_world_size() if on else 1) if world is None else world
    optimizer.zero_grad(set_to_none=True)
    mine = list(shard_keyframes(len(keyframes), rank, world))
    dev = params[GRAD_KEYS[0]].device
    if streams > 1 and dev.type == "cuda" and len(mine) > 1 and not densify_statistics:
        keys = [k for k in GRAD_KEYS if k in params]
        main = torch.cuda.current_stream(dev)
        pool = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        for s in pool:
            s.wait_stream(main)
        partial = [None] * streams
        losses = []
        # per-stream copies of the one statistic get_loss updates in place (running max radius); merged below
        vs = [dict(variables, max_2D_radius=variables["max_2D_radius"].clone()) if "max_2D_radius" in variables else dict(variables)
              for _ in range(streams)]
        for n, i in enumerate(mine):
            with torch.cuda.stream(pool[n % streams]):
                loss, vs[n % streams] = loss_fn(params, keyframes[i], vs[n % streams])
                g = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
                g = [torch.zeros_like(params[k]) if x is None else x for k, x in zip(keys, g)]
                if partial[n % streams] is None:
                    partial[n % streams] = list(g)
                else:
                    torch._foreach_add_(partial[n % streams], list(g))
                losses.append(loss.detach())
        for s in pool:
            main.wait_stream(s)
        # the partial sums and losses were allocated on the side streams and are consumed (and later freed) on `main`:
        # tell the caching allocator, or a block could be handed out again while `main` still reads it
        for t in [x for p_ in partial if p_ is not None for x in p_] + losses:
            t.record_stream(main)
        variables = vs[(len(mine) - 1) % streams]                  # `means2D` / `seen` of the last keyframe, as in the serial loop
        if "max_2D_radius" in variables:
            for v in vs:
                variables["max_2D_radius"] = torch.maximum(variables["max_2D_radius"], v["max_2D_radius"])
        live = [p for p in partial if p is not None]
        for p in live[1:]:
            torch._foreach_add_(live[0], p)
        for k, g in zip(keys, live[0]):
            params[k].grad = g
        total = float(torch.stack(losses).sum())
    else:
        total = 0.0
        for i in mine:
            loss, variables = loss_fn(params, keyframes[i], variables)
            loss.backward()                       # autograd accumulates into .grad across this rank's keyframes
            total += float(loss.detach())
    if world <= 1 or not on:
        optimizer.step()                         # one rank (or a caller that overrides rank/world to run the batch alone): no collective
    elif sharded_adam:
        reduce_scatter_adam_step(params, optimizer)
    else:
        buf = all_reduce_gradients(params, buf)
        optimizer.step()
    return total, variables, buf


# ------------------------------------------------------------------------------------------------------------
# SURVEY 8(e) "preferred refinement": reduce-scatter -> Adam on this rank's 1/world row block -> all-gather of
# the updated rows.  Same bytes on the wire as the all-reduce, 1/world of the Adam HBM traffic per GPU, and the
# result is identical because Adam is element-wise.  Moments are only kept current for the owned rows;
# `gather_moments` refreshes the full tensors before row surgery (densify / prune, every ~100 iterations).
# ------------------------------------------------------------------------------------------------------------
def _row_block(n: int, rank: int, world: int):
    rows = (n + world - 1) // world
    return rows, min(rank * rows, n), min((rank + 1) * rows, n)


def _reduce_scatter_rows(flat_padded, rows, rank, group):
    out = torch.empty(rows, flat_padded.shape[1], dtype=flat_padded.dtype, device=flat_padded.device)
    try:
        dist.reduce_scatter_tensor(out, flat_padded, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):          # gloo (CPU tests) has no reduce-scatter
        dist.all_reduce(flat_padded, op=dist.ReduceOp.SUM, group=group)
        out.copy_(flat_padded[rank * rows:(rank + 1) * rows])
    return out


def _all_gather_rows(full_padded, mine, group):
    try:
        dist.all_gather_into_tensor(full_padded, mine, group=group)
    except (RuntimeError, NotImplementedError):
        parts = list(full_padded.chunk(dist.get_world_size(group), dim=0))
        dist.all_gather(parts, mine, group=group)
    return full_padded


def reduce_scatter_adam_step(params, optimizer, group=None):
    """Replaces `all_reduce_gradients(...); optimizer.step()` when every rank holds replicated parameters."""
    from . import _lib, optim as O
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buf = FlatGradBuffer(params)
    n, G = buf.flat.shape
    rows, lo, hi = _row_block(n, rank, world)
    padded = torch.zeros(rows * world, G, dtype=torch.float32, device=buf.flat.device)
    padded[:n] = buf.pack(params)
    gshard = _reduce_scatter_rows(padded, rows, rank, group)
    lib = _lib.get()
    pshard = torch.zeros(rows, G, dtype=torch.float32, device=buf.flat.device)
    groups = {g["name"]: g for g in optimizer.param_groups}
    col = 0
    with torch.no_grad():
        for k, w in zip(buf.keys, buf.widths):
            p, g = params[k], groups[k]
            st = optimizer.state.get(p)
            if st is None or len(st) == 0:
                st = optimizer.state[p] = {"step": torch.tensor(0.0), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            st["step"] = st["step"] + 1
            if hi > lo:
                grad = gshard[:hi - lo, col:col + w].contiguous()
                b1, b2 = g["betas"]
                _lib.check(lib.gs_adam_step((hi - lo) * w, p.data[lo:hi].data_ptr(), grad.data_ptr(), st["exp_avg"][lo:hi].data_ptr(),
                                            st["exp_avg_sq"][lo:hi].data_ptr(), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                            int(st["step"].item()), O._stream(p)))
                pshard[:hi - lo, col:col + w] = p.data[lo:hi]
            col += w
        full = _all_gather_rows(padded, pshard, group)     # reuse the padded buffer for the updated rows
        col = 0
        for k, w in zip(buf.keys, buf.widths):
            params[k].data.copy_(full[:n, col:col + w])
            col += w


def gather_moments(params, optimizer, group=None):
    """Make exp_avg / exp_avg_sq complete on every rank (each rank only advanced its own row block)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for k in GRAD_KEYS:
        st = optimizer.state.get(params.get(k))
        if not st:
            continue
        n = params[k].shape[0]
        rows, lo, hi = _row_block(n, rank, world)
        for name in ("exp_avg", "exp_avg_sq"):
            t = st[name].reshape(n, -1)
            mine = torch.zeros(rows, t.shape[1], dtype=t.dtype, device=t.device)
            mine[:hi - lo] = t[lo:hi]
            full = _all_gather_rows(torch.empty(rows * world, t.shape[1], dtype=t.dtype, device=t.device), mine, group)
            st[name].copy_(full[:n].reshape(st[name].shape))
