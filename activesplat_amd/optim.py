"""Per-Gaussian optimiser: fused Adam + prune / densify surgery on the HIP library.

Host-side mirror of the reference's optimiser handling (same names, argument meaning and results; pinned
by tests/golden/adam.npz and prune.npz generated from the reference):

  initialize_optimizer            src/mapper/splatam/splatam.py:118-124
                                  (torch.optim.Adam(param_groups, lr=0.0, eps=1e-15): one group per key, betas
                                  (0.9, 0.999), no weight decay; tensors whose .grad is None are skipped)
  accumulate_mean2d_gradient      src/mapper/splatam/utils/slam_external.py:100-108
  update_params_and_optimizer     :111-123     cat_params_to_optimizer  :126-140
  remove_points                   :143-164     inverse_sigmoid          :167
  prune_gaussians                 :171-192     densify                  :195-247

What is native here: GaussianAdam.step() is ONE fused HIP kernel per parameter tensor (gs_adam_step, 28 B of
HBM traffic per element) instead of torch's multi-kernel foreach path, and every row surgery is a
gs_compact_index + gs_gather_rows pair shared by the parameter, both Adam moments and the densification
statistics.  The optimiser keeps torch.optim's `param_groups` / `state[param]` layout so that code written
against the reference's surgery functions keeps working.

Fused surgery: the clone / split / cull decisions of one densify or prune event are taken by ONE kernel
(gs_densify_classify), their three masks become index lists through the ballot/popcount compaction, and every tensor --
parameter, both Adam moments, statistics -- is produced by ONE row gather (gs_gather_rows_zero_tail: rows of new Gaussians
zero-filled in the moments); the split children's offset and scale are applied in place by gs_densify_children.  Same rows in
the same order as the step-by-step formulation (clone -> cat -> split -> cat -> remove -> cull -> remove, the reference's call pattern: four
index builds and ~40 torch launches per event; kept in tests/reference_pattern.py as the comparison baseline of the parity tests).

densify: the reference's slam_external.densify only executes for isotropic scales without a 'timestep'
variable.  This mirror reproduces that case exactly and defines the missing ones the way
the reference's own dead copy (utils/gs_external.py:191-253) and the original 3DGS do: anisotropic split
noise uses the per-axis scales, clones and splits inherit the parent's timestep.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_SKIP = ("cam_unnorm_rots", "cam_trans")


def _stream(t):
    from . import _lib as L
    return L.stream_ptr(t.device)


class GaussianAdam:
    """Drop-in for torch.optim.Adam(param_groups, lr=0.0, eps=1e-15) as the reference configures it."""

    def __init__(self, param_groups, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr); g.setdefault("betas", betas); g.setdefault("eps", eps)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self):
        """All parameter tensors that hold a gradient advance in ONE kernel launch (gs_adam_step_multi)."""
        lib = _lib.get()
        live, keep, stream = [], [], None
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("GaussianAdam needs contiguous fp32 parameters")
                st = self.state.get(p)
                if st is None or len(st) == 0:
                    st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                # the step counter is a host NUMBER (torch.optim.Adam keeps a host tensor; every use here and in the reference --
                # float(), +, comparison -- works on both, and a 0-dim tensor increment costs 3.6 us per parameter per step)
                st["step"] = int(st["step"]) + 1
                grad = p.grad
                if not grad.is_contiguous() or grad.dtype != torch.float32:
                    grad = grad.contiguous().float()
                keep.append(grad)
                live.append((g, p, st, grad))
        if not live:
            return
        # The descriptor array of the launch is kept from step to step: while the same parameter and moment tensors take part, only the
        # gradient pointers, step counters and hyper-parameters are refreshed (building seven ctypes structures per step was a good part
        # of this function's host time; a mapping iteration at the reference's 256 x 256 is bound by host time, not by the GPU).
        # (keyed on addresses AND sizes: after a prune + densify in one iteration the caching allocator can hand a freed address to a tensor
        # with another row count)
        ident = tuple((p.numel(), p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for _, p, st, _ in live)
        cache = getattr(self, "_launch_cache", None)
        if cache is None or cache[0] != ident:
            arr = (_lib.GsAdamTensor * len(live))()
            for i, (g, p, st, grad) in enumerate(live):
                b1, b2 = g["betas"]
                arr[i] = _lib.GsAdamTensor(p.numel(), p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                           float(g["lr"]), float(b1), float(b2), float(g["eps"]), 0, 0)
            cache = self._launch_cache = (ident, arr)
        arr = cache[1]
        for i, (g, p, st, grad) in enumerate(live):
            t = arr[i]
            b1, b2 = g["betas"]
            t.n = p.numel(); t.grad = grad.data_ptr(); t.step = st["step"]; t.lr = float(g["lr"])
            t.beta1 = float(b1); t.beta2 = float(b2); t.eps = float(g["eps"])
        _lib.check(lib.gs_adam_step_multi(len(live), arr, _stream(live[0][1])))


    @torch.no_grad()
    def backward_step_descriptors(self, tensors):
        """For the optimiser step INSIDE the rasteriser's backward (rasterizer.render_rgbd_raw(adam=...), gs_render_backward_raw_adam): the
        GsAdamTensor descriptors of `tensors` (parameters of this optimiser, in the kernel's order), with this step's bookkeeping done --
        state created on first use, step counters advanced -- exactly as step() would.  The kernel then updates parameter and moments in
        place; the tensors keep .grad = None, so a later step() skips them."""
        # (the descriptor array is kept from call to call while the same tensors take part: only step counters and hyper-parameters are refreshed)
        cache = getattr(self, "_backward_cache", None)
        ident = tuple((id(p), p.data_ptr(), p.numel()) for p in tensors)
        if cache is None or cache[0] != ident:
            by_id = {id(p): g for g in self.param_groups for p in g["params"]}
            arr = (_lib.GsAdamTensor * len(tensors))()
            entries = []
            for i, p in enumerate(tensors):
                g = by_id.get(id(p))
                if g is None:
                    raise RuntimeError("fused Adam: a rendered tensor is not a parameter of this optimiser")
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise RuntimeError("GaussianAdam needs contiguous fp32 parameters")
                st = self.state.get(p)
                if st is None or len(st) == 0:
                    st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                arr[i] = _lib.GsAdamTensor(p.numel(), p.data_ptr(), None, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), 0.0, 0.0, 0.0, 0.0, 1, 0)
                entries.append((g, st, p))
            cache = self._backward_cache = (ident, arr, entries, tuple((st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for _, st, _ in entries))
        _, arr, entries, moments = cache
        # validate EVERY entry before any counter moves: a refusal at entry i must not leave the counters of the entries before it advanced
        for i, (g, st, p) in enumerate(entries):
            if p.grad is not None:
                raise RuntimeError("fused Adam: a parameter already holds a gradient (accumulated keyframes?) -- step() it or zero_grad() first")
            if (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) != moments[i]:      # (row surgery replaced the moments: rebuild)
                self._backward_cache = None
                return self.backward_step_descriptors(tensors)
        for i, (g, st, p) in enumerate(entries):
            st["step"] = int(st["step"]) + 1
            b1, b2 = g["betas"]
            t = arr[i]
            t.lr = float(g["lr"]); t.beta1 = float(b1); t.beta2 = float(b2); t.eps = float(g["eps"]); t.step = int(st["step"])
        return arr

    def rollback_backward_step(self, tensors):
        """Undo the bookkeeping of backward_step_descriptors(tensors) for a launch that failed: the step counters go back by one, so that moments
        and counters stay in step."""
        for p in tensors:
            st = self.state.get(p)
            if st:
                st["step"] = int(st["step"]) - 1


def densify_event(iter, densify_dict) -> bool:
    """Does densify(..., iter, densify_dict) restructure the map or reset parameters at this iteration (as opposed to only accumulating
    statistics)?  A loop that steps the optimiser inside the backward kernel takes the separate step on such iterations: row surgery
    replaces the parameter tensors between backward and step, and the reference's step then skips them (slam_external.py:110-193)."""
    if iter > densify_dict["stop_after"]:
        return False
    if iter >= densify_dict["start_after"] and iter % densify_dict["densify_every"] == 0:
        return True
    return bool(iter > 0 and iter % densify_dict["reset_opacities_every"] == 0 and densify_dict.get("reset_opacities", False))


def densify_restructures(iter, densify_dict) -> bool:
    """The clone / split / remove branch of densify_event alone (slam_external.py:199): the one that consumes AND zeroes the accumulators.  An
    opacity-reset-only iteration (slam_external.py:243-246) leaves them standing."""
    return bool(iter <= densify_dict["stop_after"] and iter >= densify_dict["start_after"] and iter % densify_dict["densify_every"] == 0)


def prune_event(iter, prune_dict) -> bool:
    """Does prune_gaussians(..., iter, prune_dict) remove rows or reset opacities at this iteration?  (see densify_event)"""
    if iter > prune_dict["stop_after"]:
        return False
    if iter >= prune_dict["start_after"] and iter % prune_dict["prune_every"] == 0:
        return True
    return bool(iter > 0 and iter % prune_dict["reset_opacities_every"] == 0 and prune_dict["reset_opacities"])


def initialize_optimizer(params, lrs_dict, tracking=False):
    groups = [{"params": [v], "name": k, "lr": lrs_dict[k]} for k, v in params.items()]
    return GaussianAdam(groups) if tracking else GaussianAdam(groups, lr=0.0, eps=1e-15)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def visibility_stats(radius: torch.Tensor, max_2D_radius: torch.Tensor) -> torch.Tensor:
    """seen = radius > 0 (returned, bool) and max_2D_radius = max(max_2D_radius, radius) IN PLACE -- splatam.py:296-298 in one
    launch (gs_visibility_stats).  Falls back to torch ops for tensors the kernel does not take (dtype / layout)."""
    P = int(radius.shape[0])
    if radius.dtype == torch.int32 and radius.is_contiguous() and max_2D_radius.dtype == torch.float32 and max_2D_radius.is_contiguous() \
            and max_2D_radius.shape == (P,):
        seen = torch.empty(P, dtype=torch.bool, device=radius.device)
        _lib.check(_lib.get().gs_visibility_stats(P, radius.data_ptr(), seen.data_ptr(), max_2D_radius.data_ptr(), _stream(radius)))
        return seen
    seen = radius > 0
    max_2D_radius.copy_(torch.maximum(max_2D_radius, radius.to(max_2D_radius.dtype)))
    return seen


def accumulate_mean2d_gradient(variables):
    """slam_external.py:100-108: accum[seen] += ||means2D.grad[seen, :2]||, denom[seen] += 1 -- one launch
    (gs_accumulate_grad2d) instead of boolean-index gathers and a host sync on `seen.sum()`; identical values."""
    g = variables["means2D"].grad
    if g is None or g.shape[0] != variables["means2D"].shape[0] or g.shape[1] < 2:
        return variables
    seen, accum, denom = variables["seen"], variables["means2D_gradient_accum"], variables["denom"]
    P = int(g.shape[0])
    if g.dtype == torch.float32 and g.is_contiguous() and g.shape[1] == 3 and seen.dtype == torch.bool and seen.is_contiguous() \
            and accum.dtype == torch.float32 and accum.is_contiguous() and denom.dtype == torch.float32 and denom.is_contiguous():
        _lib.check(_lib.get().gs_accumulate_grad2d(P, g.data_ptr(), seen.data_ptr(), accum.data_ptr(), denom.data_ptr(), _stream(g)))
        return variables
    norm = torch.norm(g[:, :2], dim=-1)
    accum += torch.where(seen, norm, torch.zeros_like(norm))
    denom += seen.to(denom.dtype)
    return variables


# ---- row surgery on the HIP library ---------------------------------------------------------------------
def build_index(keep: torch.Tensor) -> torch.Tensor:
    """keep mask [n] (bool) -> int32 tensor of kept row indices, ascending (gs_compact_index)."""
    lib = _lib.get()
    n = keep.numel()
    k8 = keep.to(torch.uint8).contiguous()
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=keep.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=keep.device)
    scratch = torch.empty(int(lib.gs_compact_scratch_bytes(n)), dtype=torch.uint8, device=keep.device)
    _lib.check(lib.gs_compact_index(n, k8.data_ptr(), idx.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), _stream(keep)))
    return idx[: int(cnt.item())]


def gather_rows(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """dst[r] = src[index[r]] for a contiguous fp32 tensor whose first dimension is the Gaussian (gs_gather_rows)."""
    lib = _lib.get()
    src = src.detach().contiguous().float()
    n_out = index.numel()
    row = src.numel() // max(src.shape[0], 1)
    dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n_out:
        _lib.check(lib.gs_gather_rows(n_out, row, index.data_ptr(), src.data_ptr(), dst.data_ptr(), _stream(src)))
    return dst


def gather_rows_zero_tail(src: torch.Tensor, index: torch.Tensor, n_copy: int) -> torch.Tensor:
    """dst[r] = src[index[r]] for r < n_copy, zeros for n_copy <= r < len(index) -- one launch."""
    lib = _lib.get()
    src = src.detach().contiguous().float()
    n_out = index.numel()
    row = src.numel() // max(src.shape[0], 1)
    dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n_out:
        _lib.check(lib.gs_gather_rows_zero_tail(n_out, n_copy, row, index.data_ptr(), src.data_ptr(), dst.data_ptr(), _stream(src)))
    return dst


def _index_of(mask_u8: torch.Tensor) -> torch.Tensor:
    lib = _lib.get()
    n = mask_u8.numel()
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=mask_u8.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=mask_u8.device)
    scratch = torch.empty(int(lib.gs_compact_scratch_bytes(n)), dtype=torch.uint8, device=mask_u8.device)
    _lib.check(lib.gs_compact_index(n, mask_u8.data_ptr(), idx.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), _stream(mask_u8)))
    return idx, cnt


def _classify(params, variables, opacity_thresh, remove_big, densify_args=None):
    """gs_densify_classify -> uint8 masks [4, N]: keep_orig, keep_clone, keep_child, split (the last three zero when pruning)."""
    lib = _lib.get()
    ls, lo = params["log_scales"].detach().contiguous(), params["logit_opacities"].detach().contiguous()
    N, dev = int(ls.shape[0]), ls.device
    masks = torch.zeros(4, max(N, 1), dtype=torch.uint8, device=dev)
    radius = variables["scene_radius"]
    radius = (radius if torch.is_tensor(radius) else torch.tensor(float(radius))).to(device=dev, dtype=torch.float32).reshape(1)
    acc = den = None
    thr, nsplit = 0.0, 1
    if densify_args is not None:
        acc = variables["means2D_gradient_accum"].contiguous().float()
        den = variables["denom"].contiguous().float()
        thr, nsplit = densify_args
    _lib.check(lib.gs_densify_classify(N, int(ls.shape[1]), ls.data_ptr(), lo.data_ptr(), None if acc is None else acc.data_ptr(),
                                       None if den is None else den.data_ptr(), radius.data_ptr(), float(thr), float(opacity_thresh),
                                       1 if remove_big else 0, int(nsplit), masks[0].data_ptr(),
                                       masks[1].data_ptr() if acc is not None else None, masks[2].data_ptr() if acc is not None else None,
                                       masks[3].data_ptr() if acc is not None else None, _stream(ls)))
    return masks


def _apply_index(index, n_copy, params, variables, optimizer, zero_stats):
    """ONE gather per tensor: parameters copy all rows of `index`; Adam moments copy the first n_copy rows (the Gaussians that were
    there before) and start the appended ones from zero; statistics are reset (densify) or compacted (prune)."""
    n_out, dev = int(index.numel()), index.device
    for k in [k for k in params.keys() if k not in _SKIP]:
        g = _group(optimizer, k)
        old = g["params"][0]
        st = optimizer.state.get(old, None)
        new_st = None
        if st is not None and len(st):
            new_st = dict(st, exp_avg=gather_rows_zero_tail(st["exp_avg"], index, n_copy),
                          exp_avg_sq=gather_rows_zero_tail(st["exp_avg_sq"], index, n_copy))
        params[k] = _replace_param(optimizer, g, gather_rows(old, index), new_st)
    for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
        if k in variables:
            variables[k] = torch.zeros(n_out, device=dev) if zero_stats else gather_rows(variables[k], index)
    if "timestep" in variables:
        variables["timestep"] = gather_rows(variables["timestep"], index)      # clones and children inherit the parent's
    return params, variables


def _replace_param(optimizer, group, new_tensor, new_state):
    old = group["params"][0]
    optimizer.state.pop(old, None)
    p = torch.nn.Parameter(new_tensor.requires_grad_(True))
    group["params"][0] = p
    if new_state is not None:
        optimizer.state[p] = new_state
    return p


def _group(optimizer, name):
    return [g for g in optimizer.param_groups if g["name"] == name][0]


def update_params_and_optimizer(new_params, params, optimizer):
    """Replace tensors wholesale; moments restart from zero, the step counter is kept (slam_external.py:111-123)."""
    for k, v in new_params.items():
        g = _group(optimizer, k)
        st = optimizer.state.get(g["params"][0], None)
        new_st = None if st is None else dict(st, exp_avg=torch.zeros_like(v), exp_avg_sq=torch.zeros_like(v))
        params[k] = _replace_param(optimizer, g, v.detach().clone(), new_st)
    return params


def cat_params_to_optimizer(new_params, params, optimizer):
    """Append rows; new rows get zero moments, the per-tensor step counter is preserved (slam_external.py:126-140)."""
    for k, v in new_params.items():
        g = _group(optimizer, k)
        old = g["params"][0]
        st = optimizer.state.get(old, None)
        new_st = None
        if st is not None and len(st):
            new_st = dict(st, exp_avg=torch.cat((st["exp_avg"], torch.zeros_like(v)), dim=0),
                          exp_avg_sq=torch.cat((st["exp_avg_sq"], torch.zeros_like(v)), dim=0))
        params[k] = _replace_param(optimizer, g, torch.cat((old.detach(), v.detach()), dim=0), new_st)
    return params


def remove_points(to_remove, params, variables, optimizer):
    """Compact every per-Gaussian tensor, its Adam moments and the statistics with ONE index build."""
    index = build_index(~to_remove)
    for k in [k for k in params.keys() if k not in _SKIP]:
        g = _group(optimizer, k)
        old = g["params"][0]
        st = optimizer.state.get(old, None)
        new_st = None
        if st is not None and len(st):
            new_st = dict(st, exp_avg=gather_rows(st["exp_avg"], index), exp_avg_sq=gather_rows(st["exp_avg_sq"], index))
        params[k] = _replace_param(optimizer, g, gather_rows(old, index), new_st)
    for k in ("means2D_gradient_accum", "denom", "max_2D_radius", "timestep"):
        if k in variables:
            variables[k] = gather_rows(variables[k], index)
    return params, variables


def prune_gaussians(params, variables, optimizer, iter, prune_dict):
    if iter <= prune_dict["stop_after"]:
        if iter >= prune_dict["start_after"] and iter % prune_dict["prune_every"] == 0:
            thr = prune_dict["final_removal_opacity_threshold"] if iter == prune_dict["stop_after"] \
                else prune_dict["removal_opacity_threshold"]
            remove_big = iter >= prune_dict["remove_big_after"]
            idx, cnt = _index_of(_classify(params, variables, thr, remove_big)[0])
            n = int(cnt.item())
            params, variables = _apply_index(idx[:n], n, params, variables, optimizer, zero_stats=False)
        if iter > 0 and iter % prune_dict["reset_opacities_every"] == 0 and prune_dict["reset_opacities"]:
            new = {"logit_opacities": inverse_sigmoid(torch.ones_like(params["logit_opacities"]) * 0.01)}
            params = update_params_and_optimizer(new, params, optimizer)
    return params, variables


def _densify_fused(params, variables, optimizer, iter, densify_dict, samples, seed=None):
    """One classification, one index, one gather per tensor.  Output rows, as the reference orders them: surviving originals
    (split parents gone), surviving clones, then num_to_split_into blocks of surviving children (parents ascending in a block)."""
    lib = _lib.get()
    n_split = int(densify_dict["num_to_split_into"])
    thr = densify_dict["final_removal_opacity_threshold"] if iter == densify_dict["stop_after"] \
        else densify_dict["removal_opacity_threshold"]
    masks = _classify(params, variables, thr, iter >= densify_dict["remove_big_after"], (densify_dict["grad_thresh"], n_split))
    # the three masks -> ONE index list [originals | clones | n_split blocks of split parents] in one count / scan / write sequence
    N, dev = int(masks.shape[1]), masks.device
    idx = torch.empty(max(N, 1) * (2 + n_split), dtype=torch.int32, device=dev)
    cnt = torch.zeros(3, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(lib.gs_compact3_scratch_bytes(N)), dtype=torch.uint8, device=dev)
    _lib.check(lib.gs_compact_index3(N, masks[0].data_ptr(), masks[1].data_ptr(), masks[2].data_ptr(), n_split, idx.data_ptr(), cnt.data_ptr(),
                                     scratch.data_ptr(), _stream(masks)))
    n_orig, n_clone, n_child = (int(v) for v in cnt.tolist())                                        # the event's one host read
    index = idx[:n_orig + n_clone + n_child * n_split]
    params, variables = _apply_index(index, n_orig, params, variables, optimizer, zero_stats=True)
    n_kids = n_child * n_split
    if n_kids:
        base = n_orig + n_clone
        ls = params["log_scales"].detach()
        kid_ls = ls[base:]
        kid_samples, seed = None, 0
        if samples is None:
            # drawn inside the kernel (counter-based): the event's seed comes from torch's CPU generator, so torch.manual_seed makes a run
            # repeatable; no sample tensor, no torch.normal / exp / repeat launches.  A keyframe-sharded loop hands in `seed` -- a function of
            # replicated state (parallel.event_seed) -- so that every rank draws the same offsets without a broadcast
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()) if seed is None else int(seed) & (2 ** 62 - 1)
        else:
            # injected offsets are indexed like the reference's un-culled split list: block c, rank of the parent among ALL split parents
            rank = torch.cumsum(masks[3].to(torch.int64), 0) - 1
            n_all = int(masks[3].sum().item())
            parents = index[base:base + n_child].to(torch.int64)
            rows = torch.cat([c * n_all + rank[parents] for c in range(n_split)])
            kid_samples = samples.to(dev).float()[rows].contiguous()
        means, rots = params["means3D"].detach(), params["unnorm_rotations"].detach()
        kid_rots = rots[base:].contiguous()
        _lib.check(lib.gs_densify_children(n_kids, int(ls.shape[1]), n_split, kid_rots.data_ptr(),
                                           None if kid_samples is None else kid_samples.data_ptr(), seed,
                                           means[base:].data_ptr(), kid_ls.data_ptr(), _stream(means)))
    return params, variables


def densify(params, variables, optimizer, iter, densify_dict, samples=None, seed=None, accumulate=True):
    """Clone small / split large high-gradient Gaussians, then cull (slam_external.py:195-247): one classification kernel, one index, one row
    gather per tensor (the reference's step-by-step call pattern lives in tests/reference_pattern.py as the comparison baseline).
    `samples` optionally injects the N(0, scale) split offsets ([n_split * num_to_split_into, 3]) so that a run
    can be replayed exactly; otherwise they are drawn in the kernel from `seed` (None: a seed from torch's CPU generator).
    accumulate=False: the caller has already added this iteration's mean-2D gradient statistics (parallel.sharded_densify, whose
    accumulators are rank-local partial sums until the event's all-reduce)."""
    if iter > densify_dict["stop_after"]:
        return params, variables
    if accumulate:
        variables = accumulate_mean2d_gradient(variables)
    if iter >= densify_dict["start_after"] and iter % densify_dict["densify_every"] == 0:
        params, variables = _densify_fused(params, variables, optimizer, iter, densify_dict, samples, seed)
    if iter > 0 and iter % densify_dict["reset_opacities_every"] == 0 and densify_dict.get("reset_opacities", False):
        new = {"logit_opacities": inverse_sigmoid(torch.ones_like(params["logit_opacities"]) * 0.01)}
        params = update_params_and_optimizer(new, params, optimizer)
    return params, variables
