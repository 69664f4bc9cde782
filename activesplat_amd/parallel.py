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
#: maps whose colour is a row of SH coefficients (BASELINE configs[2]: `shs` [N,16,3]) exchange those rows in place of `rgb_colors`:
#: G = 3 + 48 + 4 + 1 + 3 = 59 floats per Gaussian, 472 MB at N = 2 M
SH_KEY = "shs"


def grad_keys(params):
    """The per-Gaussian tensors of `params` whose gradients one optimiser step exchanges, in flat-buffer column order."""
    return [k for k in ("means3D", "rgb_colors", SH_KEY, "unnorm_rotations", "logit_opacities", "log_scales") if k in params]


def _row_width(p) -> int:
    """floats per Gaussian of a per-Gaussian tensor ([N,3] -> 3, [N,16,3] -> 48)"""
    w = 1
    for d in p.shape[1:]:
        w *= int(d)
    return w


def shard_keyframes(num_keyframes: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first (num_keyframes % world) ranks take one extra keyframe."""
    base, extra = divmod(num_keyframes, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def balanced_partition(costs, world: int):
    """Longest-processing-time assignment of keyframes to ranks: keyframes in descending cost (ties by index) each go to the rank with the
    smallest load so far (ties by rank).  Deterministic in `costs` -- every rank computes the same lists from replicated costs (KeyframeCosts).
    -> list (per rank) of ascending keyframe indices.  Real keyframes differ in their tile-instance count by integer factors (a wall at one
    metre against a view down a corridor), which contiguous blocks (shard_keyframes) hand to the ranks as they come: SURVEY section 8e's >= 6x
    at 8 GPUs hinges on the slowest rank."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], len(out[j]), j))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(x) for x in out]


class KeyframeCosts:
    """Per-keyframe cost estimates for balanced_partition, kept identical on every rank: the owner of a keyframe records what its last render
    cost (`record`: tile instances D, plus `gaussian_weight` instance-equivalents per Gaussian for the per-Gaussian stages -- at 2 M Gaussians /
    4.84 M instances the per-Gaussian kernels are 265 us of a 660 us frame: 1.6), `sync` combines the ranks' records (one all-reduce of K
    floats: every keyframe has exactly one owner per step), `partition` is the LPT assignment.  Before any record: contiguous blocks."""

    def __init__(self, num_keyframes: int, gaussian_weight: float = 1.6):
        self.k, self.gw = int(num_keyframes), float(gaussian_weight)
        self.cost = [0.0] * self.k
        self._fresh = {}

    def record(self, index: int, tile_instances: int, gaussians: int = 0):
        self._fresh[int(index)] = float(tile_instances) + self.gw * float(gaussians)

    def sync(self, device="cpu", group=None):
        t = torch.zeros(self.k, dtype=torch.float64)
        for i, c in self._fresh.items():
            t[i] = c
        self._fresh = {}
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            t = t.cpu()
        for i in range(self.k):
            if float(t[i]) > 0.0:
                self.cost[i] = float(t[i])
        return self

    def partition(self, world: int):
        if not any(c > 0.0 for c in self.cost):
            return [list(shard_keyframes(self.k, r, world)) for r in range(world)]
        known = [c for c in self.cost if c > 0.0]
        fill = sum(known) / len(known)                    # a keyframe nobody has rendered yet counts as an average one
        return balanced_partition([c if c > 0.0 else fill for c in self.cost], world)


#: what the most recent gradient exchange of this process actually ran (bench.py prints it): backend, the collectives, and -- with
#: `timing=True` -- CUDA events around exchange + Adam
last_exchange = {}


def _backend(group=None) -> str:
    return str(dist.get_backend(group)).lower()


def _rccl_collectives(t, group=None) -> bool:
    """Does this group run the tensor collectives (reduce_scatter_tensor / all_gather_into_tensor) for `t`?  "nccl" (= RCCL on ROCm) in the
    backend's name decides; a group made without an explicit backend ("cpu:gloo,cuda:nccl", which some torch versions report as
    "undefined") is decided by where the tensor lives."""
    b = _backend(group)
    if "nccl" in b:
        return True
    if b == "gloo":
        return False
    return bool(t.is_cuda)


def _stream(t):
    import ctypes as C
    from . import _lib as L
    return L.stream_ptr(t.device)


def _row_tensors(params, keys, optimizer=None, target="param"):
    """HOST descriptor array of the per-key tensors for the gs_pack_columns / gs_adam_rows / gs_unpack_columns launches.
    target="param": rows are read/written in the parameters; target="grad": in their .grad tensors (all-reduce path)."""
    from . import _lib
    arr = (_lib.GsRowTensor * len(keys))()
    keep = []
    groups = {g["name"]: g for g in optimizer.param_groups} if optimizer is not None else {}
    for i, k in enumerate(keys):
        p = params[k]
        if not p.is_contiguous() or p.dtype != torch.float32:
            raise RuntimeError("keyframe-sharded step needs contiguous fp32 parameters")
        g = p.grad
        if g is not None and (not g.is_contiguous() or g.dtype != torch.float32):
            g = g.contiguous().float()
        keep.append(g)
        t = arr[i]
        t.width = _row_width(p)
        t.grad = g.data_ptr() if g is not None else None
        t.param = (g.data_ptr() if g is not None else None) if target == "grad" else p.data_ptr()
        t.step = 1
        if optimizer is not None:
            grp, st = groups[k], optimizer.state[p]
            b1, b2 = grp["betas"]
            t.exp_avg, t.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            t.lr, t.beta1, t.beta2, t.eps, t.step = float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), int(st["step"])
    return arr, keep


class FlatGradBuffer:
    """One contiguous [N_padded, G] fp32 buffer that the per-key gradients are packed into for a single collective
    (gs_pack_columns: ONE launch writes gradients, zero columns of keys without a gradient and the zero padding rows)."""

    def __init__(self, params, keys=None, pad_to=1):
        self.keys = grad_keys(params) if keys is None else [k for k in keys if k in params]
        self.widths = [_row_width(params[k]) for k in self.keys]
        if sum(self.widths) > 64:
            raise RuntimeError(f"keyframe-sharded step: {sum(self.widths)} floats per Gaussian, gs_pack_columns takes at most 64")
        self.n = int(params[self.keys[0]].shape[0])
        self.n_padded = (self.n + pad_to - 1) // pad_to * pad_to
        dev = params[self.keys[0]].device
        self.padded = torch.empty(self.n_padded, sum(self.widths), dtype=torch.float32, device=dev)
        self.flat = self.padded[:self.n]

    def pack(self, params):
        from . import _lib
        arr, keep = _row_tensors(params, self.keys)
        _lib.check(_lib.get().gs_pack_columns(len(self.keys), arr, self.n, self.n_padded, self.padded.data_ptr(), _stream(self.padded)))
        return self.flat

    def unpack(self, params):
        """flat -> the keys' .grad tensors (in place where a gradient tensor exists), one launch."""
        from . import _lib
        for k in self.keys:
            if params[k].grad is None or not params[k].grad.is_contiguous():
                params[k].grad = torch.empty_like(params[k])
        arr, keep = _row_tensors(params, self.keys, target="grad")
        _lib.check(_lib.get().gs_unpack_columns(len(self.keys), arr, self.n, self.padded.data_ptr(), _stream(self.padded)))


def all_reduce_gradients(params, buf: FlatGradBuffer | None = None, group=None, force=False):
    """Sum the locally accumulated .grad of every per-Gaussian tensor over all ranks (one all-reduce of the flat [N, 14] buffer).
    A one-rank group is a no-op unless force=True (the collective then really runs: RCCL's first contact in the 1-GPU tests)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return buf
    if buf is None or buf.n != int(params[buf.keys[0]].shape[0]):
        buf = FlatGradBuffer(params)
    dist.all_reduce(buf.pack(params), op=dist.ReduceOp.SUM, group=group)
    buf.unpack(params)
    last_exchange.update(backend=_backend(group), reduce="all_reduce", gather=None, bytes=buf.flat.numel() * 4)
    return buf


def all_reduce_statistics(variables, group=None):
    """Densification statistics live per rank as PARTIAL sums (every rank adds what its own keyframes saw); combine them -- once -- right before
    a densify event (sum, sum, max): sharded_densify does.  (Reducing accumulators that already hold reduced values would count them world times.)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return variables
    for k, op in (("means2D_gradient_accum", dist.ReduceOp.SUM), ("denom", dist.ReduceOp.SUM), ("max_2D_radius", dist.ReduceOp.MAX)):
        if k in variables:
            dist.all_reduce(variables[k], op=op, group=group)
    return variables


def sharded_keyframe_step(params, variables, keyframes, optimizer, loss_fn, rank=None, world=None, buf=None,
                          sharded_adam=False, streams=1, densify_statistics=False, timing=False, accumulate_statistics=False, partition=None,
                          costs=None):
    """One optimiser step over a batch of keyframes sharded across ranks.
    loss_fn(params, keyframe, variables) -> (loss, variables).  Returns the local loss sum.

    streams > 1 (GPU only): this rank's keyframes are independent until the optimiser step, so they are rendered on
    `streams` HIP streams in turn -- one frame's kernels fill the tails and the placement imbalance of the other's
    (round 2, one backward walker per quadrant: 2404 -> 3062 frames/s on BASELINE configs[1]; with the chained backward blend of round 3 a 640 x 480
    keyframe fills the chip by itself and ONE stream is faster -- configs[3] on one GPU: 1371-1442 keyframes/s against 1276-1278 on two; the option is
    for frames that do not, e.g. the reference's 256 x 256).  Each keyframe's
    gradients are taken with autograd.grad on its own stream and summed after the streams have joined.
    autograd.grad does not populate `means2D.grad`, which the densifier's statistics read (optim.accumulate_mean2d_gradient):
    a caller that densifies from this step's statistics passes densify_statistics=True and gets the serial walk.
    accumulate_statistics=True (implies the serial walk): EVERY keyframe of this rank adds its mean-2D gradient norm and visibility to the
    densifier's accumulators right after its backward (optim.accumulate_mean2d_gradient) -- the batch's statistic is then the sum over all of its
    keyframes whatever the number of ranks; the accumulators are rank-local partial sums until parallel.sharded_densify all-reduces them.
    partition: list of this step's keyframe indices per rank (parallel.balanced_partition); default: contiguous blocks (shard_keyframes).
    costs: a KeyframeCosts that the serial walk feeds with every rendered keyframe's tile-instance count (sync + partition are the caller's).
    (Round 6 built and removed a PIPELINED two-stream walk with in-kernel accumulation -- only the accumulating kernels ordered across the streams:
    -5 ... +4 % against this serial walk at 2 M Gaussians / 64 keyframes, kill criterion + 6 %: profiles/r06_ab_pipelined.txt.)"""
    on = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank() if on else 0) if rank is None else rank
    world = (dist.get_world_size() if on else 1) if world is None else world
    optimizer.zero_grad(set_to_none=True)
    mine = list(shard_keyframes(len(keyframes), rank, world)) if partition is None else list(partition[rank])
    densify_statistics = densify_statistics or accumulate_statistics
    keys = grad_keys(params)
    dev = params[keys[0]].device
    rev = None
    if timing and dev.type == "cuda":                   # this rank's keyframes (render + loss + backward), event-timed: bench.py's per-rank figure
        rev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        rev[0].record()
    if streams > 1 and dev.type == "cuda" and len(mine) > 1 and not densify_statistics:
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
                if all(x is None for x in g):
                    # a loss_fn with accumulate_grads=True (mapping.get_loss / rasterizer.render_rgbd_raw) adds its gradients into .grad inside
                    # the backward kernel and hands autograd nothing: on this multi-stream walk they would be replaced by zeros -- and the
                    # kernels of several side streams would be adding into one .grad without ordering
                    raise RuntimeError("sharded_keyframe_step(streams > 1): loss_fn returned no gradient for any per-Gaussian tensor -- in-kernel "
                                       "gradient accumulation (accumulate_grads=True) only works on the serial walk (streams=1)")
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
        from .mapping import backward_on_calling_thread, unit_gradient
        losses = []
        for i in mine:
            loss, variables = loss_fn(params, keyframes[i], variables)
            if costs is not None:                 # (the forward has read its counters back already: a host number, no sync)
                from . import rasterizer as _R
                costs.record(i, _R.last_stats.get("num_rendered", 0), _R.last_stats.get("P", 0))
            # dL/dloss = 1 from the cache (autograd would fill a fresh one per keyframe: a launch each; the fused loss skips its scaling
            # launch for this very tensor); autograd accumulates into .grad across this rank's keyframes
            with backward_on_calling_thread():
                loss.backward(unit_gradient(loss) if loss.dim() == 0 else None)
            losses.append(loss.detach())          # (read after the loop: a float() here would stall the host once per keyframe)
            if accumulate_statistics:
                from .optim import accumulate_mean2d_gradient
                variables = accumulate_mean2d_gradient(variables)
        total = float(torch.stack(losses).sum()) if losses else 0.0
    if rev is not None:
        rev[1].record()
        last_exchange["render_events"] = rev
    if world <= 1 or not on:
        optimizer.step()                         # one rank (or a caller that overrides rank/world to run the batch alone): no collective
    elif sharded_adam:
        reduce_scatter_adam_step(params, optimizer, timing=timing)
    else:
        buf = all_reduce_gradients(params, buf)
        optimizer.step()
    return total, variables, buf


# ------------------------------------------------------------------------------------------------------------
# SURVEY 8(e) "preferred refinement": reduce-scatter -> Adam on this rank's 1/world row block -> all-gather of
# the updated rows.  Same bytes on the wire as the all-reduce, 1/world of the Adam HBM traffic per GPU.  Adam is
# element-wise, so every row is stepped by exactly one rank and copied: all ranks end bit-identical at ANY world size.
# Against all-reduce + full Adam the result is equal up to the order in which the collective adds the `world` partial
# gradients (<= 1e-6 rel; to the bit only for two ranks or unpadded buffers).  Moments are only kept current for the owned rows;
# `gather_moments` refreshes the full tensors before row surgery (densify / prune, every ~100 iterations).
# ------------------------------------------------------------------------------------------------------------
def _row_block(n: int, rank: int, world: int):
    rows = (n + world - 1) // world
    return rows, min(rank * rows, n), min((rank + 1) * rows, n)


def _reduce_scatter_rows(flat_padded, out, rank, group):
    """out[rows, G] = this rank's row block of the sum over ranks.  RCCL: reduce_scatter_tensor; gloo (CPU tests, the same-device
    development knob of bench.py) has no reduce-scatter: all-reduce, then the row block.  The branch is decided by the BACKEND --
    a collective that fails raises."""
    rows = out.shape[0]
    if _rccl_collectives(out, group):
        dist.reduce_scatter_tensor(out, flat_padded, op=dist.ReduceOp.SUM, group=group)
        return "reduce_scatter_tensor"
    dist.all_reduce(flat_padded, op=dist.ReduceOp.SUM, group=group)
    out.copy_(flat_padded[rank * rows:(rank + 1) * rows])
    return "all_reduce+slice"


def _all_gather_rows(full_padded, mine, group):
    if _rccl_collectives(mine, group):
        dist.all_gather_into_tensor(full_padded, mine, group=group)
        return "all_gather_into_tensor"
    parts = list(full_padded.chunk(dist.get_world_size(group), dim=0))
    dist.all_gather(parts, mine, group=group)
    return "all_gather(list)"


def reduce_scatter_adam_step(params, optimizer, group=None, timing=False):
    """Replaces `all_reduce_gradients(...); optimizer.step()` when every rank holds replicated parameters:
    pack (1 launch) -> reduce-scatter -> fused Adam on the rank's row block of all keys (1 launch, writes the all-gather's send
    buffer) -> all-gather -> unpack (1 launch).  The buffers live on the optimizer and are reused from step to step."""
    from . import _lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    keys = grad_keys(params)
    n = int(params[keys[0]].shape[0])
    plan = getattr(optimizer, "_shard_plan", None)
    ident = (n, world, rank, id(group), tuple(keys), params[keys[0]].device)
    if plan is None or plan["ident"] != ident:
        rows, lo, hi = _row_block(n, rank, world)
        buf = FlatGradBuffer(params, pad_to=rows * world)          # whole row blocks: rows * world >= n
        G = buf.padded.shape[1]
        plan = optimizer._shard_plan = dict(ident=ident, n=n, world=world, rows=rows, lo=lo, hi=hi, buf=buf,
                                            gshard=torch.empty(rows, G, dtype=torch.float32, device=buf.padded.device),
                                            pshard=torch.empty(rows, G, dtype=torch.float32, device=buf.padded.device))
    buf, rows, lo, hi = plan["buf"], plan["rows"], plan["lo"], plan["hi"]
    lib = _lib.get()
    ev = None
    if timing and buf.padded.is_cuda:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
    buf.pack(params)
    how_r = _reduce_scatter_rows(buf.padded, plan["gshard"], rank, group)
    with torch.no_grad():
        for k in keys:
            p = params[k]
            st = optimizer.state.get(p)
            if st is None or len(st) == 0:
                st = optimizer.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            st["step"] = int(st["step"]) + 1
        arr, keep = _row_tensors(params, keys, optimizer)
        _lib.check(lib.gs_adam_rows(len(keys), arr, lo, hi - lo, rows, plan["gshard"].data_ptr(), plan["pshard"].data_ptr(), _stream(buf.padded)))
        how_g = _all_gather_rows(buf.padded, plan["pshard"], group)      # the send buffer of the reduce-scatter takes the updated rows
        _lib.check(lib.gs_unpack_columns(len(keys), arr, n, buf.padded.data_ptr(), _stream(buf.padded)))
    if ev is not None:
        ev[1].record()
    last_exchange.update(backend=_backend(group), reduce=how_r, gather=how_g, bytes=buf.padded.numel() * 4, events=ev)


def render_ms():
    """Milliseconds this rank's keyframes of the most recent timed step took (render + loss + backward, before the exchange); None if not timed."""
    ev = last_exchange.get("render_events")
    if not ev:
        return None
    ev[1].synchronize()
    return float(ev[0].elapsed_time(ev[1]))


def exchange_ms():
    """Milliseconds of the most recent timed exchange + sharded Adam (reduce_scatter_adam_step(timing=True)); None if not timed."""
    ev = last_exchange.get("events")
    if not ev:
        return None
    ev[1].synchronize()
    return float(ev[0].elapsed_time(ev[1]))


def gather_moments(params, optimizer, group=None):
    """Make exp_avg / exp_avg_sq complete on every rank (each rank only advanced its own row block)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for k in grad_keys(params):
        st = optimizer.state.get(params.get(k))
        if not st:
            continue
        n = params[k].shape[0]
        rows, lo, hi = _row_block(n, rank, world)
        for name in ("exp_avg", "exp_avg_sq"):
            t = st[name].reshape(n, -1)
            mine = torch.zeros(rows, t.shape[1], dtype=t.dtype, device=t.device)
            mine[:hi - lo] = t[lo:hi]
            full = torch.empty(rows * world, t.shape[1], dtype=t.dtype, device=t.device)
            _all_gather_rows(full, mine, group)
            st[name].copy_(full[:n].reshape(st[name].shape))


# ------------------------------------------------------------------------------------------------------------
# Row surgery in a keyframe-sharded loop ("decisions are deterministic functions of replicated state").
# Between events the densifier's accumulators are rank-local PARTIAL sums (every rank adds what its own keyframes saw) and the Adam moments
# are current only on the owner's row block; an event first makes both complete and identical everywhere -- all-reduce (sum, sum, max),
# gather_moments -- then every rank takes the same decisions on the same numbers and draws the split offsets from the same counter-based
# seed.  The accumulators come out of a densify event as zeros (a valid partial sum), the shard plan is rebuilt for the new N by the next step.
# ------------------------------------------------------------------------------------------------------------
def event_seed(base_seed: int, iter: int, n: int) -> int:
    """62-bit seed of an event's in-kernel split offsets: a splitmix64 mix of (run seed, iteration, Gaussian count) -- replicated state only."""
    x = (int(base_seed) ^ (int(iter) * 0x9E3779B97F4A7C15) ^ (int(n) * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (x ^ (x >> 31)) & (2 ** 62 - 1)


def _world(group, world):
    if world is not None:
        return int(world)
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def sharded_densify(params, variables, optimizer, iter, densify_dict, group=None, base_seed=0x5EED, accumulate=True, world=None, before_event=None):
    """optim.densify for a keyframe-sharded loop (call it on EVERY rank at the same iteration; slam_external.py:195-247).
    accumulate=True: this rank's most recent keyframe adds its statistics first (the reference's per-iteration accumulation; pass False when
    sharded_keyframe_step(accumulate_statistics=True) already added every keyframe's).  world=1 skips the collectives (one rank running alone).
    before_event(params, variables, densify_dict) -> densify_dict: called on event iterations once the statistics are complete and identical on
    every rank (an adaptive gradient threshold, logging); whatever it decides must be a function of those replicated values."""
    from . import optim as O
    if iter > densify_dict["stop_after"]:
        return params, variables
    if accumulate:
        variables = O.accumulate_mean2d_gradient(variables)
    if O.densify_event(iter, densify_dict):
        # only the restructuring branch consumes -- and zeroes -- the accumulators.  An opacity-reset-only iteration leaves them standing: reducing
        # them there would turn the partial sums into full ones on every rank, and the next real event would count them `world` times.  (The reset
        # writes one constant into every row of the opacities and zeroes their moments: nothing rank-local survives it, no gather either.)
        if _world(group, world) > 1 and O.densify_restructures(iter, densify_dict):
            all_reduce_statistics(variables, group)
            gather_moments(params, optimizer, group)
        if before_event is not None:
            densify_dict = before_event(params, variables, densify_dict)
        n = int(params["means3D"].shape[0])
        params, variables = O.densify(params, variables, optimizer, iter, densify_dict, seed=event_seed(base_seed, iter, n), accumulate=False)
        optimizer._shard_plan = None
    return params, variables


def sharded_prune(params, variables, optimizer, iter, prune_dict, group=None, world=None):
    """optim.prune_gaussians for a keyframe-sharded loop (slam_external.py:171-192): the moments are completed on every rank before rows move;
    the decisions read parameters only (opacity, scale), which are replicated.  The statistics stay rank-local partial sums, compacted like the rest."""
    from . import optim as O
    if O.prune_event(iter, prune_dict):
        if _world(group, world) > 1:
            gather_moments(params, optimizer, group)
        params, variables = O.prune_gaussians(params, variables, optimizer, iter, prune_dict)
        optimizer._shard_plan = None
    return params, variables
