"""Autograd glue between the adapter modules and the HIP primitives.

Device tensors only: everything here ends in a call through the C-ABI
(``_C.require()`` raises if the library is missing).  The frozen dense
contractions (``X @ W^T``, ``G @ W``) are either part of the fused MFMA kernels
(``csrc/gemm_ws.hip``, ``csrc/gemm_fused.hip``: one launch for frozen product + low-rank
branch, dropout included) or plain library GEMMs (hipBLASLt via ``F.linear``/``matmul``)
followed by the streaming kernels of ``csrc/linear_fused.hip`` / ``csrc/linear.hip``;
which one a site gets is a fixed function of its shape (``_C.static_fwd_choice``).
Convolution sites: ``csrc/conv.hip`` (NCHW) or ``csrc/conv_nhwc.hip`` (channels_last).
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn.functional as F

from . import _C

def next_dropout_stream(device) -> tuple:
    """(seed, offset) of one dropout mask.  ``seed`` is torch's seed (a launch-time scalar); ``offset`` is a
    1-element int64 DEVICE tensor drawn from torch's device generator, which the kernels read at run time
    (``offset_dev`` of the C-ABI).  Drawing it with torch's generator is what keeps the mask correct everywhere the
    forward can be re-executed without this Python code running again, or with it running twice:

    * hipGraph replay (``trainer.GraphedForwardBackward``): the draw is a captured kernel on torch's graph-safe Philox
      state, so every replay sees a fresh value (a host counter would be baked into the graph);
    * activation checkpointing: ``torch.utils.checkpoint`` restores the generator state before the recompute, so the
      recomputed forward regenerates the SAME masks as the original one;
    * ``torch.manual_seed`` makes the whole sequence reproducible.

    The backward regenerates the mask from the same pair: no [M, N] mask tensor ever exists in HBM."""
    seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    pool = _POOL.get("buf")
    if pool is not None and pool.device == torch.device(device):
        i = _POOL["next"]
        if i >= pool.numel():  # more masked sites than the pool holds: one more draw for the next batch of them
            pool = _POOL["buf"] = torch.empty(_POOL_SIZE, dtype=torch.int64, device=device).random_(0, 1 << 62)
            i = 0
        _POOL["next"] = i + 1
        return seed, pool[i:i + 1]
    off = torch.empty(1, dtype=torch.int64, device=device).random_(0, 1 << 62)
    return seed, off


# One draw per STEP instead of one per masked site (configs[3]: 224 one-element RNG launches per step): inside
# ``dropout_pool`` the offsets are consecutive elements of one tensor filled by a single ``random_`` launch.  Only the
# training step opens it (trainer.forward_backward): the per-site draw above is what keeps activation checkpointing
# correct for everyone else (the recompute must see the SAME offsets, which torch's restored generator state gives it; a
# pool would hand the recompute fresh ones).  hipGraph capture: the fill is a captured kernel, every replay refills.
_POOL = {}
_POOL_SIZE = 512


class dropout_pool:
    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        if self.device.type == "cuda":
            _POOL["buf"] = torch.empty(_POOL_SIZE, dtype=torch.int64, device=self.device).random_(0, 1 << 62)
            _POOL["next"] = 0
        return self

    def __exit__(self, *exc):
        _POOL.pop("buf", None)
        return False


# nn.Dropout on the branch runs inside the weight-stationary kernels; under dropout they also take the shapes whose p = 0
# choice is another kernel, forward and backward (the alternative there is the rank-16 VALU kernel on top of the library
# GEMM; configs[3], same box: 21.26 -> 22.20 -> 22.44 steps/s, round 2).  Module constants (tests flip them), no switches.
WS_DROPOUT = WS_DROPOUT_WIDE = WS_DROPOUT_WIDE_BWD = True
# the dropout sites around the attention core in head-padded rows on the weight-stationary kernel (round 6; False = the
# unpack -> kernel -> pack copies of rounds 3-5: the A/B and the parity twin)
WS_HEADS = True


# When a list: every adapter forward / backward appends (phase, kernel path, M, K, N, r) — what bench.py turns into
# ``config.kernel_choices`` and the adapter-path byte count.  None (default): nothing is recorded.
PATH_LOG: Optional[list] = None


def _log(phase: str, path: str, M: int, K: int, N: int, r: int) -> None:
    if PATH_LOG is not None:
        PATH_LOG.append((phase, path, int(M), int(K), int(N), int(r)))


class GradSink:
    """Where one adapter's backward leaves its parameter gradients when a trainer owns them:
    views into the flat gradient buffer + persistent partial-sum workspaces.  The partials are summed
    into the flat buffer by ONE batched launch per step (``FlatLoraState.reduce_pending``)."""

    __slots__ = ("down_grad", "up_grad", "ws", "rows", "pending", "owner")

    def __init__(self, down_grad: torch.Tensor, up_grad: torch.Tensor, owner=None):
        self.down_grad, self.up_grad = down_grad, up_grad
        self.ws = {}        # shape key -> tuple of persistent workspaces
        self.rows = {}      # shape key -> reduce_batched rows folding that workspace's partials into the grads
        self.pending = None  # key of the workspace holding not-yet-reduced partials
        self.owner = owner

    def _new(self, key, w, rows):
        self.ws[key], self.rows[key] = w, rows
        if self.owner is not None:
            self.owner._reduce_table = None  # table must be rebuilt with the new buffers
        return w

    def workspace(self, key, plan, device):
        """Linear site, key = (M, K, N, r): (gt_part, up_part, down_part)."""
        w = self.ws.get(key)
        if w is None:
            w = tuple(torch.empty(max(int(n), 1), dtype=torch.float32, device=device)
                      for n in (plan.gt_part_floats, plan.up_part_floats, plan.down_part_floats))
            _, K, N, r = key
            self._new(key, w, [(w[1], self.up_grad, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 1.0),
                               (w[2], self.down_grad, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 1.0)])
        return w

    def self_workspace(self, key, plan, device):
        """Merged-weight Linear site, key = ("self", M, K, N, r, row blocks): (up_part, down_part) of
        ``linear_bwd_factors_self`` / the deferred one-launch pass."""
        w = self.ws.get(key)
        if w is None:
            w = tuple(torch.empty(max(int(n), 1), dtype=torch.float32, device=device)
                      for n in (plan.up_part_floats, plan.down_part_floats))
            _, _, K, N, r, _ = key
            self._new(key, w, [(w[0], self.up_grad, plan.nparts, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 1.0),
                               (w[1], self.down_grad, plan.nparts, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 1.0)])
        return w

    def conv_workspace(self, key, plan, device):
        """Conv site, key = ("conv", B, Ci, Co, H, W, ks, r): (t_part, gt_part, gt, up_part, down_part)."""
        w = self.ws.get(key)
        if w is None:
            _, B, Ci, Co, H, W, ks, r = key
            w = conv_buffers(plan, B, r, H * W, device)
            self._new(key, w, conv_reduce_rows(w, plan, Ci, Co, ks, r, self.up_grad, self.down_grad, 1.0))
        return w

    def conv3_nhwc_workspace(self, key, lplan, cplan, device):
        """Channels-last 3x3 site, key = ("conv3n", B, Ci, Co, H, W, r): (gt_part, up_part, gt, down_part)."""
        w = self.ws.get(key)
        if w is None:
            _, B, Ci, Co, H, W, r = key
            w = conv3_nhwc_buffers(lplan, cplan, B * H * W, r, device)
            self._new(key, w, conv3_nhwc_reduce_rows(w, lplan, cplan, Ci, Co, r, self.up_grad, self.down_grad, 1.0))
        return w

    def reduce_rows(self, key, plan=None):
        return self.rows[key]

    def flush(self):
        """Reduce this site's pending partials now (a second backward is about to overwrite them)."""
        if self.pending is None:
            return
        mw = getattr(self.owner, "merged", None)
        if mw is not None:
            mw.flush_factors()  # the partials may still be owed by the deferred one-launch pass
        table, n, total = _C.make_reduce_table(self.rows[self.pending], self.up_grad.device)
        _C.reduce_batched(table, n, total)
        self.pending = None


def conv_buffers(plan, B: int, r: int, HW: int, device):
    f = lambda n: torch.empty(max(int(n), 1), dtype=torch.float32, device=device)  # noqa: E731
    return (f(plan.t_part_floats), f(plan.gt_part_floats), f(B * r * HW), f(plan.up_part_floats),
            f(plan.down_part_floats))


def conv_reduce_rows(w, plan, Ci, Co, ks, r, up_grad, down_grad, beta):
    return [(w[3], up_grad, plan.ngroups_out, plan.rank_pad, Co, r, _C.FACTOR_KR, 1.0, beta),
            (w[4], down_grad, plan.ngroups_in, plan.rank_pad, Ci * ks * ks, r, _C.FACTOR_RK, 1.0, beta)]


def conv3_nhwc_buffers(lplan, cplan, M: int, r: int, device):
    f = lambda n: torch.empty(max(int(n), 4), dtype=torch.float32, device=device)  # noqa: E731
    return (f(lplan.gt_part_floats), f(lplan.up_part_floats), f(M * r), f(cplan.down_part_floats))


def conv3_nhwc_reduce_rows(w, lplan, cplan, Ci, Co, r, up_grad, down_grad, beta):
    return [(w[1], up_grad, lplan.nparts_up, lplan.rank_tile, Co, r, _C.FACTOR_KR, 1.0, beta),
            (w[3], down_grad, cplan.nsplit, cplan.rank_pad, Ci * 9, r, _C.FACTOR_RK, 1.0, beta)]


def _rows2d(t: torch.Tensor, cols: int) -> torch.Tensor:
    t2 = t.reshape(-1, cols)
    if t2.stride(-1) != 1 or (t2.shape[0] > 1 and t2.stride(0) != cols):
        t2 = t2.contiguous()
    return t2


# ---- ranks beyond the kernels' LORA_AMD_MAX_RANK (64): the reference accepts any r <= min(in, out), and rank-joined LoRA
# files (lora_manager.lora_join, LoRAManager) pass 64 quickly.  The primitives are linear in the rank dimension, so the
# rank is cut into chunks of <= 64; a selector (an [r, r] matrix across ALL ranks) is applied to the assembled T with
# one small matmul in between.
def _rank_chunks(r: int):
    return [(a, min(a + _C.MAX_RANK, r)) for a in range(0, r, _C.MAX_RANK)]


def _slice_factor(f: torch.Tensor, layout: int, a: int, b: int) -> torch.Tensor:
    return (f[a:b] if layout == _C.FACTOR_RK else f[:, a:b]).contiguous()


def rowdot_any(x, f, layout, scale=1.0, sel=None, sel_t=False, p=0.0, seed=0, off=0):
    r = f.shape[0] if layout == _C.FACTOR_RK else f.shape[1]
    if r <= _C.MAX_RANK:
        return _C.rowdot(x, f, layout, scale, sel, sel_t, p, seed, off)
    t = torch.cat([_C.rowdot(x, _slice_factor(f, layout, a, b), layout, scale, None, False, p, seed, off)
                   for a, b in _rank_chunks(r)], dim=1)
    if sel is not None:
        t = t @ (sel.float() if sel_t else sel.float().t())
    return t.contiguous()


def rank_update_any_(y, t, f, layout, scale=1.0, p=0.0, seed=0, off=0):
    r = t.shape[1]
    if r <= _C.MAX_RANK:
        return _C.rank_update_(y, t, f, layout, scale, p, seed, off)
    if p > 0.0:
        # the mask belongs to the SUM over ranks, so the chunks cannot be masked one by one: draw the mask itself with a
        # rank-1 launch of the same kernel (ones x ones -> 0 or 1/(1-p) per element, indexed by (row, column) of the
        # [M, N] output exactly as the backward's rowdot / colreduce regenerate it), then apply it to the full product
        M, N = y.shape
        one_t = torch.ones((M, 1), dtype=torch.float32, device=y.device)
        one_f = torch.ones((1, N) if layout == _C.FACTOR_RK else (N, 1), dtype=torch.float32, device=y.device)
        mask = _C.rank_update_(torch.zeros((M, N), dtype=torch.float32, device=y.device), one_t, one_f, layout, 1.0, p,
                               seed, off)
        ff = f.float()
        y.add_((mask * (t @ (ff if layout == _C.FACTOR_RK else ff.t()))).mul_(scale).to(y.dtype))
        return y
    for a, b in _rank_chunks(r):
        _C.rank_update_(y, t[:, a:b].contiguous(), _slice_factor(f, layout, a, b), layout, scale, 0.0, 0, 0)
    return y


def colreduce_any(x, t, layout, scale=1.0, out=None, beta=0.0, p=0.0, seed=0, off=0):
    r = t.shape[1]
    if r <= _C.MAX_RANK:
        return _C.colreduce(x, t, layout, scale, out=out, beta=beta, dropout_p=p, seed=seed, offset=off)
    parts = [_C.colreduce(x, t[:, a:b].contiguous(), layout, scale, dropout_p=p, seed=seed, offset=off)
             for a, b in _rank_chunks(r)]
    full = torch.cat(parts, dim=0 if layout == _C.FACTOR_RK else 1)
    if out is None:
        return full
    out.mul_(beta).add_(full.view_as(out))
    return out


class LoraLinearFunction(torch.autograd.Function):
    """y = x W^T + b + scale * dropout((x A^T) S^T B^T)   (lora.py:53-58) and its gradient.

    Saves X and the [M, r] projection T only — never an [M, N] tensor.  16-byte-friendly shapes take the
    fused kernels (1 launch forward, 2 backward); anything else the three primitives.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, down, up, sel, scale, dropout_p, sink, in_heads=None, out_heads=None):
        _C.require()
        N, K = weight.shape
        r = down.shape[0]
        x2 = _rows2d(x, _C.heads_width(K, in_heads))
        seed = off = 0
        if dropout_p > 0.0:
            seed, off = next_dropout_stream(x.device)
        down_c, up_c = down.contiguous(), up.contiguous()
        ctx.in_heads, ctx.out_heads = in_heads, out_heads
        if in_heads is not None or out_heads is not None:
            # a dropout site around the attention core (ws_heads_route_ok was the caller's check): the weight-stationary
            # kernel reads / writes the head-padded rows itself — no pad / slice copy on either side
            y, t = _C.linear_ws_fwd(x2, weight, bias, down_c, up_c, scale, 0, dropout_p, seed, off, x_heads=in_heads,
                                    y_heads=out_heads)
            _log("fwd", "ws_heads", x2.shape[0], K, N, r)
            ctx.save_for_backward(x2, weight, down, up, t, sel)
            ctx.scale, ctx.p, ctx.seed, ctx.off = float(scale), float(dropout_p), seed, off
            ctx.has_bias, ctx.x_shape, ctx.sink, ctx.fused = bias is not None, x.shape, sink, True
            return y.view(*x.shape[:-1], y.shape[1])
        tile = 0
        if (sel is None and down_c.dtype == torch.float32 and up_c.dtype == torch.float32
                and x2.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x2.dtype):
            # ONE launch on the matrix cores (frozen GEMM + low-rank branch): which kernel is a fixed function of the shape
            tile = _C.gemm_choice(x2, weight, bias, down_c, up_c, scale)
            if dropout_p > 0.0:
                # nn.Dropout on the branch: only the weight-stationary kernel regenerates the mask; with WS_DROPOUT_WIDE it
                # also takes the shapes whose p = 0 choice is another kernel (the alternative here is the rank-16 VALU
                # kernel on top of the library GEMM), except the 1280-deep GEGLU projections
                if WS_DROPOUT_WIDE and tile != _C.WS_TILE and _C.ws_supported(x2, K, N, r) and not (K == 1280 and N >= 4 * K):
                    tile = _C.WS_TILE
                if tile != _C.WS_TILE or N % 8 or not WS_DROPOUT:
                    tile = 0
        if tile == _C.WS_TILE:
            y, t = _C.linear_ws_fwd(x2, weight, bias, down_c, up_c, scale, 0, dropout_p, seed, off)
            fused = _C.fused_ok(x2, N, r)
            _log("fwd", "ws", x2.shape[0], K, N, r)
        elif tile:
            y, t = _C.linear_gemm_fwd(x2, weight, bias, down_c, up_c, scale, tile)
            fused = _C.fused_ok(x2, N, r)
            _log("fwd", f"ring{tile}", x2.shape[0], K, N, r)
        else:
            y = F.linear(x2, weight, bias)  # frozen dense GEMM (MFMA, hipBLASLt)
            fused = down_c.dtype == up_c.dtype and _C.fused_ok(x2, N, r) and _C._rows_ok(y)
            if fused:
                t = _C.linear_fwd_(x2, y, down_c, up_c, scale, sel, dropout_p, seed, off)
                _log("fwd", "lib+linear_fwd", x2.shape[0], K, N, r)
            else:
                t = rowdot_any(x2, down_c, _C.FACTOR_RK, 1.0, sel, False)
                rank_update_any_(y, t, up_c, _C.FACTOR_KR, scale, dropout_p, seed, off)
                _log("fwd", "lib+primitives", x2.shape[0], K, N, r)
        ctx.save_for_backward(x2, weight, down, up, t, sel)
        ctx.scale, ctx.p, ctx.seed, ctx.off = float(scale), float(dropout_p), seed, off
        ctx.has_bias, ctx.x_shape, ctx.sink, ctx.fused = bias is not None, x.shape, sink, fused
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def _backward_heads(ctx, g):
        """Backward of a dropout site whose forward ran in head-padded rows: dX / Gt on the weight-stationary kernel in the
        same layouts and both factor gradients in the step's deferred matrix-core pass (which takes the head layouts and
        regenerates the mask); anything else — no trainer state, a shape the kernels refuse — detours through dense copies
        and the regular backward."""
        x2, weight, down, up, t, sel = ctx.saved_tensors
        in_heads, out_heads = ctx.in_heads, ctx.out_heads
        N, K = weight.shape
        r, M = down.shape[0], x2.shape[0]
        g2 = _rows2d(g, _C.heads_width(N, out_heads))
        if not _C._rows_ok(g2):
            g2 = g2.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        s, p, seed, off, sink = ctx.scale, ctx.p, ctx.seed, ctx.off, ctx.sink
        down_c, up_c = down.contiguous(), up.contiguous()
        mw = getattr(getattr(sink, "owner", None), "merged", None)
        plan_m = None
        if (mw is not None and mw.defer_factors and _C.FACTORS_MFMA and DEFER_MASKED_FACTORS and not need_w
                and g2.dtype == x2.dtype):
            plan_m = _C.factors_mfma_plan(M, K, N, r, g2.dtype)
            if not plan_m.supported:
                plan_m = None
        dx_ok = (not need_x) or (WS_DROPOUT and _C.ws_supported(g2, N, K, r) and weight.is_contiguous()
                                 and _C.ws_heads_ok(N, K, out_heads, in_heads))
        if plan_m is None or not dx_ok:
            from types import SimpleNamespace

            x_log = unpack_heads(x2, in_heads) if in_heads else x2
            g_log = unpack_heads(g2, out_heads) if out_heads else g2
            fake = SimpleNamespace(saved_tensors=(x_log, weight, down, up, t, sel), in_heads=None, out_heads=None,
                                   needs_input_grad=tuple(ctx.needs_input_grad[:5]) + (False,) * 6, scale=s, p=p,
                                   seed=seed, off=off, sink=sink, fused=_C.fused_ok(x_log, N, r), has_bias=ctx.has_bias,
                                   x_shape=x_log.shape)
            dx, dw, db, d_down, d_up = LoraLinearFunction.backward(fake, g_log)[:5]
            if dx is not None:
                if in_heads:
                    dx = pack_heads(dx, in_heads)
                dx = dx.view(*ctx.x_shape[:-1], dx.shape[-1])
            return dx, dw, db, d_down, d_up, None, None, None, None, None, None
        if sink.pending is not None:
            sink.flush()
        dx = None
        if need_x:
            dx2, _gt = _C.linear_ws_dx(g2, weight, down_c, up_c, s, 0, p, seed, off, g_heads=out_heads, dx_heads=in_heads)
            dx = dx2.view(*ctx.x_shape[:-1], dx2.shape[1])
        key = ("mfma", M, K, N, r, int(plan_m.nparts))
        up_part_m, down_part_m = sink.self_workspace(key, plan_m, g2.device)
        mw.owe(g2, x2, down_c, up_c, up_part_m, down_part_m, s, out_heads, in_heads, "mfma", plan_m, (p, seed, off))
        _log("bwd", ("ws_heads_dx+" if need_x else "") + "factors_deferred_mfma", M, K, N, r)
        sink.pending = key
        db = None
        if ctx.has_bias and need_b:
            db = (unpack_heads(g2, out_heads) if out_heads else g2).sum(0)
        return dx, None, db, None, None, None, None, None, None, None, None

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if getattr(ctx, "in_heads", None) is not None or getattr(ctx, "out_heads", None) is not None:
            return LoraLinearFunction._backward_heads(ctx, g)
        x2, weight, down, up, t, sel = ctx.saved_tensors
        N, K = weight.shape
        r = down.shape[0]
        M = x2.shape[0]
        g2 = _rows2d(g, N)
        need_x, need_w, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        s, p, seed, off = ctx.scale, ctx.p, ctx.seed, ctx.off
        sink = ctx.sink
        d_up = d_down = dx = None
        down_c, up_c = down.contiguous(), up.contiguous()

        if ctx.fused and _C._rows_ok(g2) and (need_down or need_up or need_x):
            key = (M, K, N, r)
            plan = _C.linear_plan(*key)
            if sink is not None and sink.pending is not None:
                sink.flush()
            mw = getattr(getattr(sink, "owner", None), "merged", None)
            plan_m = None
            if (mw is not None and mw.defer_factors and _C.FACTORS_MFMA and DEFER_MASKED_FACTORS and sel is None
                    and down_c.dtype == torch.float32 and up_c.dtype == torch.float32
                    and g2.dtype in (torch.bfloat16, torch.float16) and x2.dtype == g2.dtype):
                plan_m = _C.factors_mfma_plan(M, K, N, r, g2.dtype)
                if not plan_m.supported:
                    plan_m = None
            if not need_x and not need_w and plan_m is not None:
                # no input gradient wanted (cross-attention k / v on the text states, the time-embedding projections):
                # the backward of the site is its two factor gradients, and both are the deferred pass's — no launch here
                # (configs[3]: 54 sites x two latency-bound launches of 10-23 us each)
                key = ("mfma", M, K, N, r, int(plan_m.nparts))
                up_part_m, down_part_m = sink.self_workspace(key, plan_m, g2.device)
                mw.owe(g2, x2, down_c, up_c, up_part_m, down_part_m, s, None, None, "mfma", plan_m, (p, seed, off))
                _log("bwd", "factors_deferred_mfma", M, K, N, r)
                sink.pending = key
                db = g2.sum(0) if (ctx.has_bias and need_b) else None
                return None, None, db, None, None, None, None, None, None, None, None
            # the per-site partial slabs (Gt column tiles, dUp / dDown row blocks) exist only on the paths that write them: a
            # site whose factor gradients go to the step's deferred matrix-core pass registers nothing here (the
            # dead slabs of every dropout site, and a reset of the trainer's reduce table per site)
            key0, _bufs = key, []

            def bufs():
                if not _bufs:
                    if sink is not None:
                        _bufs.append(sink.workspace(key0, plan, g2.device))
                    else:
                        _bufs.append(tuple(torch.empty(max(int(n), 1), dtype=torch.float32, device=g2.device)
                                           for n in (plan.gt_part_floats, plan.up_part_floats, plan.down_part_floats)))
                return _bufs[0]
            tile = 0
            if (need_x and sel is None and down_c.dtype == torch.float32 and up_c.dtype == torch.float32
                    and weight.is_contiguous() and g2.dtype in (torch.bfloat16, torch.float16)
                    and weight.dtype == g2.dtype):
                tile = _C.gemm_choice_bwd(g2, x2, weight, t, down_c, up_c, s, bufs)
                if (p > 0.0 and WS_DROPOUT_WIDE_BWD and tile != _C.WS_TILE and M >= 256
                        and _C.ws_supported(g2, N, K, r) and K % 8 == 0):
                    tile = _C.WS_TILE
                if p > 0.0 and (tile != _C.WS_TILE or not WS_DROPOUT):
                    tile = 0  # the mask of the forward: weight-stationary kernel or the three-launch path
            if tile:
                # dX = G W + s (G up) down and Gt = s G up in ONE MFMA launch (weight-stationary on W^T packed in
                # fragment order, or the LDS-ring kernel on the resident W^T); what remains are the parameter-gradient
                # partials (G^T T and Gt^T X), both in one more launch
                if tile == _C.WS_TILE:
                    dx2, gt = _C.linear_ws_dx(g2, weight, down_c, up_c, s, 0, p, seed, off)
                else:
                    dx2, gt = _C.linear_gemm_dx(g2, _C.weight_t(weight), down_c, up_c, s, tile)
                # both factor gradients: deferred to the step's one-launch matrix-core pass when a trainer state runs one
                # (T recomputed from X, the forward's dropout mask regenerated on G inside the pass), else one launch here
                if plan_m is not None:
                    key = ("mfma", M, K, N, r, int(plan_m.nparts))
                    up_part_m, down_part_m = sink.self_workspace(key, plan_m, g2.device)
                    mw.owe(g2, x2, down_c, up_c, up_part_m, down_part_m, s, None, None, "mfma", plan_m, (p, seed, off))
                    _log("bwd", ("ws" if tile == _C.WS_TILE else f"ring{tile}") + "_dx+factors_deferred_mfma", M, K, N, r)
                else:
                    gt_part, up_part, down_part = bufs()
                    _C.linear_bwd_factors(g2, t, up_part, x2, gt, down_part, r, s, dropout=(p, seed, off))
                    _log("bwd", ("ws" if tile == _C.WS_TILE else f"ring{tile}") + "_dx+factors", M, K, N, r)
            elif plan_m is not None and need_x and r > 8 and not need_w:
                # ranks 9..16 off the fused-GEMM tables (the GEGLU projections, M = 144 sites): Gt and the low-rank term of
                # dX on the matrix-core primitives around the frozen GEMM, both factor gradients deferred
                gt = rowdot_any(g2, up_c, _C.FACTOR_KR, s, None, True, p, seed, off)
                dx2 = g2 @ weight  # frozen dense GEMM
                if not _C._rows_ok(dx2):
                    dx2 = dx2.contiguous()
                rank_update_any_(dx2, gt, down_c, _C.FACTOR_RK, 1.0)
                key = ("mfma", M, K, N, r, int(plan_m.nparts))
                up_part_m, down_part_m = sink.self_workspace(key, plan_m, g2.device)
                mw.owe(g2, x2, down_c, up_c, up_part_m, down_part_m, s, None, None, "mfma", plan_m, (p, seed, off))
                _log("bwd", "rowdot+lib+rank_update+factors_deferred_mfma", M, K, N, r)
            else:
                _log("bwd", "g+lib+x", M, K, N, r)
                gt_part, up_part, down_part = bufs()
                _C.linear_bwd_g(g2, t, up_c, gt_part, up_part, s, p, seed, off)
                dx2 = (g2 @ weight) if need_x else None  # frozen dense GEMM
                if dx2 is not None and not _C._rows_ok(dx2):
                    dx2 = dx2.contiguous()
                _C.linear_bwd_x(x2, dx2, gt_part, plan.nct_g, down_c, sel, down_part)
            if need_x:
                dx = dx2.view(ctx.x_shape)
            if sink is not None:
                sink.pending = key  # summed into the flat grad buffer by the trainer's batched reduce
            else:
                gt_part, up_part, down_part = bufs()
                d_up = torch.empty((N, r), dtype=torch.float32, device=g2.device)
                d_down = torch.empty((r, K), dtype=torch.float32, device=g2.device)
                rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                        (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
                table, n, total = _C.make_reduce_table(rows, g2.device)
                _C.reduce_batched(table, n, total)
                d_up, d_down = d_up.to(up.dtype), d_down.to(down.dtype)
        else:
            gt = None
            _log("bwd", "primitives", M, K, N, r)
            if need_x or need_down:
                gt = rowdot_any(g2, up_c, _C.FACTOR_KR, s, sel, True, p, seed, off)  # dT = s*(G.*mask) @ B @ S
            if need_up:
                if sink is not None:
                    colreduce_any(g2, t, _C.FACTOR_KR, s, out=sink.up_grad, beta=1.0, p=p, seed=seed, off=off)
                else:
                    d_up = colreduce_any(g2, t, _C.FACTOR_KR, s, p=p, seed=seed, off=off).to(up.dtype)
            if need_down:
                if sink is not None:
                    colreduce_any(x2, gt, _C.FACTOR_RK, 1.0, out=sink.down_grad, beta=1.0)
                else:
                    d_down = colreduce_any(x2, gt, _C.FACTOR_RK, 1.0).to(down.dtype)
            if need_x:
                dx2 = g2 @ weight  # frozen dense GEMM
                rank_update_any_(dx2, gt, down_c, _C.FACTOR_RK, 1.0)
                dx = dx2.view(ctx.x_shape)
        dw = g2.t() @ x2 if need_w else None
        db = g2.sum(0) if (ctx.has_bias and need_b) else None
        return dx, dw, db, d_down, d_up, None, None, None, None, None, None


# Module constants (tests and A/B scripts flip them; not environment switches):
# input gradients of the merged-weight sites as F.linear(G, W_eff^T-stored) instead of G @ W_eff (MergedWeights.lookup);
TRANSPOSED_DX = True
# factor gradients of the dropout sites that run the fused MFMA input-gradient kernels: deferred to the step's matrix-core
# pass (mask regenerated inside) instead of one linear_bwd_factors launch per site (configs[3]: 22.54 -> 24.09 steps/s,
# same box, profiles/r04_cfg3_ab.txt);
DEFER_MASKED_FACTORS = True
# q / k / v (k / v) of an attention block on ONE concatenated scratch weight: one forward GEMM per group (+0.8 %, same box)
CONCAT_GROUPS = True
# ranks 9..16 on 16-bit activations: matrix-core forms of rowdot / rank_update / linear_fwd / linear_bwd_g
# (csrc/rank16_mfma.hip; 0 = the VALU kernels, through the library's lora_amd_rank16_mfma hook)
RANK16_MFMA = True
# the one-launch factor pass as one launch per REGISTER class (class 1 = the M = 16384 sites: three workgroups per CU,
# csrc/factor_mfma.hip) instead of one launch of the two-per-CU kernel over every site (rounds 4-5)
FM_TWO_CLASSES = True
# the factor-pass launches of one flush on forked streams (their tails could overlap) instead of back to back on the launch
# stream.  OFF: measured, same box, the captured step with the fork / join edges runs 31.3 steps/s against 35.6 without
# (profiles/r06_concurrent_factor_launches_ab.txt: a multi-stream hipGraph costs the whole step 3.9 ms, far more than three tails)
CONCURRENT_FACTOR_LAUNCHES = False
# the factor-pass tables ordered longest row blocks first (flush_factors) instead of in backward order; ..._CLASS1: also the
# class-1 table.  Same box, kernel traces (profiles/r06_fm_table_order_ab.txt): class 2 278-285 -> 248-249 us, class 1
# 319-321 -> 323 us (its blocks differ 5x, but the wide ones are a third of them and already interleaved): class 2 only
FM_LONGEST_FIRST = True
FM_LONGEST_FIRST_CLASS1 = False
# the factor pass finds a workgroup's site through the plan's block -> site map (one scalar load) instead of copying the table's
# block prefix to LDS and searching it (third session of round 6; False = the search: the A/B)
FM_BLOCK_MAP = True
# the channels-last 3x3 site as ONE forward launch (csrc/conv_nhwc.hip, round 6: batched pack once per optimiser step + the
# fused down-conv / fold / up-projection / dropout / add kernel) and its G pass with the Gt fold inside the launch; False =
# the launch sequence of rounds 3-5 (pack + down [+ sum_parts] + rank_update; bwd_g + sum_parts): the A/B and the parity twin
CONV3_FUSED = True


def apply_ab_overrides(spec: str, namespace: dict) -> dict:
    """``LORA_AMD_AB="NAME=0,OTHER=1"``: the ONE measurement switch for same-box A/B runs — flips the module constants
    above (and only those) without a code edit; every A/B log under profiles/ names the spec it ran with."""
    allowed = ("TRANSPOSED_DX", "DEFER_MASKED_FACTORS", "CONCAT_GROUPS", "RANK16_MFMA", "FM_TWO_CLASSES", "CONCURRENT_FACTOR_LAUNCHES", "FM_LONGEST_FIRST", "FM_LONGEST_FIRST_CLASS1", "FM_BLOCK_MAP", "CONV3_FUSED", "WS_HEADS", "WS_DROPOUT",
               "WS_DROPOUT_WIDE",
               "WS_DROPOUT_WIDE_BWD")
    done = {}
    for item in filter(None, (s.strip() for s in spec.split(","))):
        name, _, val = item.partition("=")
        if name not in allowed:
            raise ValueError(f"LORA_AMD_AB: unknown constant {name!r} (one of {', '.join(allowed)})")
        namespace[name] = done[name] = val.strip() not in ("0", "", "false", "False")
    return done


apply_ab_overrides(os.environ.get("LORA_AMD_AB", ""), globals())
if not RANK16_MFMA and _C.available():
    _C.rank16_mfma(0)

# Rounding of the in-step merge of 16-bit weights (csrc/merge_step.hip): "dither" (default) = nearest with a fixed
# per-element dither, so that a delta below half an ulp of the frozen weight survives in the row sums; "once" = nearest even
# (one of the two documented USER options that are environment variables read once at import — the other is
# LORA_AMD_FACTORS_MFMA in _C.py; they choose numerics / a pass, not a measurement variant, so they are not
# part of LORA_AMD_AB; tests flip the module attribute)
MERGE_ROUNDING = _C.ROUND_ONCE if os.environ.get("LORA_AMD_MERGE_ROUNDING", "dither") == "once" else _C.ROUND_DITHER


def _gemm_range():
    """Profiler range around the library GEMMs of the merged-weight path (only while bench.py's adapter-path profile is
    recording: PATH_LOG is a list), so that their device time can be attributed to the adapter path."""
    if PATH_LOG is not None:
        return torch.profiler.record_function("lora_amd::merged_gemm")
    import contextlib

    return contextlib.nullcontext()


class MergedWeights:
    """The "fused W + alpha up down" path INSIDE the training step (lora.py:635-669's merge used the way lora.py:53-58 is
    used): once per step ONE K3 launch (``lora_amd_merge_batched``, HBM-bound, every site) writes
    ``W_eff = W + scale * up @ down`` of every eligible Linear adapter into a scratch buffer; the adapter's forward is then
    the frozen dense GEMM on ``W_eff`` (``Y = X W_eff^T + b`` = ``X W^T + b + scale (X down^T) up^T``, no dropout), its
    input gradient the dense GEMM ``G W_eff``, and the parameter gradients of ALL sites one
    ``linear_bwd_factors_self_ragged`` launch after the backward (``flush_factors``; without a trainer state: one
    ``linear_bwd_factors_self`` launch per site, in its backward).
    Eligible: device tensors, dropout not in effect, no selector, frozen weight, f32 factors, rank <= 16.

    Head-padded activations (``forward_heads``): the scratch weight is laid out for them — rows of a head-padded OUTPUT
    are written head by head (pad rows stay zero), columns of a head-padded INPUT through the merge kernel's
    ``out_heads`` mapping (pad columns stay zero) — so the GEMM itself reads / writes the padded layout.

    ``refresh()`` must run after every optimiser step and before the next forward (``trainer.forward_backward`` does;
    it is part of the captured hipGraph).  Entries appear lazily on an adapter's first forward in a given layout."""

    def __init__(self, state=None):
        self.entries = {}    # (id(module), in_heads, out_heads, dtype) -> dict
        self._plans = None   # [(MergePlan, alpha)]
        self.refreshes = 0
        self._state = state  # the optimiser state whose ``step_count`` says when the merged weights went stale
        self._fresh_at = None
        # factor gradients of every site in ONE launch after the backward (trainer: FlatLoraState.reduce_pending):
        # the backward of a site only records (G, X, factors, partial slabs); see flush_factors
        self.defer_factors = state is not None
        self._owed = []
        self._tables = {}    # (pass, dtype, rank tile, LDS class, table bytes) -> [eager (pinned, device) pair, copy event, spare pairs]
        self._packs = {}     # (down ptr, up ptr, dtype) -> (pk_down, pk_up, down, up): fragment packs of the matrix-core pass
        self._pack_tables = {}
        self._graph_keep = []
        self.groups = {}     # (ids of the adapters of one input, out_heads, dtype) -> concatenated scratch weights
        self._keys = {}      # id(adapter) -> dither key (see _dither_key)

    def lookup(self, module, w, b, dt, in_heads, out_heads, need_dx: bool = True):
        """(w_eff, bias_eff, w_eff_t) for this adapter and layout; creates (and fills) the entry on first use.
        ``w_eff_t`` = the same merged weight stored transposed ([K, N]: what the library's faster [out, in]-operand GEMM
        wants for the input gradient G W_eff — 12.4 vs 14.4 us at 16384 x 320 x 320, 45.6 vs 80.7 us at 1024 x 10240 -> 1280,
        profiles/r03_kbench_gemmlayout.log); None when no input gradient is needed."""
        e = self._entry(module, w, b, dt, in_heads, out_heads)
        # f32 weights go through the collapse kernel, whose transposed sites need 32 | N (its column tiles)
        if need_dx and TRANSPOSED_DX and e["w_eff_t"] is None and (e["step"] or (in_heads is None and w.shape[0] % 32 == 0)):
            self._add_transposed(e)
        return e["w_eff"], e["b_eff"], e["w_eff_t"]

    def _refresh_if_stale(self) -> None:
        """An optimiser step since the last merge (an eager forward outside ``trainer.forward_backward``: validation,
        sampling after ``state.step()``): the scratch weights of EVERY site are re-merged before the first one is read."""
        if (self._state is not None and self._fresh_at != self._state.step_count
                and not torch.cuda.is_current_stream_capturing()):
            self.refresh()

    def _entry(self, module, w, b, dt, in_heads, out_heads, out=None):
        self._refresh_if_stale()
        key = (id(module), in_heads, out_heads, dt)
        e = self.entries.get(key)
        # the factors were re-bound to new storage (FlatLoraState aliases them into its flat buffer), the frozen weight
        # was replaced, or tune_lora_scale changed the scale: the entry is rebuilt
        if e is not None and (e["ptrs"] != (module.lora_up.weight.data_ptr(), module.lora_down.weight.data_ptr(),
                                            w.data_ptr()) or e["scale"] != float(module.scale)):
            if e.get("group") is not None:
                self.groups.pop(e["group"], None)
            e = None
        if e is None:
            e = self._create(module, w, b, in_heads, out_heads, out)
            self.entries[key] = e
        return e

    def lookup_group(self, modules, ws, bs, dt, out_heads, need_dx: bool):
        """Several adapters that read ONE tensor (attn1's to_q / to_k / to_v, attn2's to_k / to_v): their scratch weights
        are row ranges of ONE buffer (and column ranges of one transposed buffer), so that the forward of the group is
        one GEMM ``X [W_q; W_k; W_v]^T`` whose output columns are the sites' outputs.  Returns the group record
        (``cat`` [sum n_out, K], ``bias`` or None, ``splits``) or None when the sites' entries already exist apart."""
        self._refresh_if_stale()  # the first LoRA call of a UNet forward is a group (attn1 q / k / v)
        gkey = tuple(id(m) for m in modules) + (out_heads, dt)
        g = self.groups.get(gkey)
        if g is not None:
            ok = all(self.entries.get((id(m), None, out_heads, dt)) is not None and
                     self.entries[(id(m), None, out_heads, dt)].get("group") == gkey and
                     self.entries[(id(m), None, out_heads, dt)]["ptrs"] == (m.lora_up.weight.data_ptr(),
                                                                         m.lora_down.weight.data_ptr(), w.data_ptr()) and
                     self.entries[(id(m), None, out_heads, dt)]["scale"] == float(m.scale)
                     for m, w in zip(modules, ws))
            if not ok:
                for m in modules:
                    self.entries.pop((id(m), None, out_heads, dt), None)
                self.groups.pop(gkey, None)
                g = None
        if g is None:
            if any((id(m), None, out_heads, dt) in self.entries for m in modules) or ws[0].dtype == torch.float32:
                return None
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MergedWeights: a new adapter group appeared during hipGraph capture")
            K = ws[0].shape[1]
            n_outs = [(out_heads[0] * out_heads[2] if out_heads else w.shape[0]) for w in ws]
            cat = torch.zeros((sum(n_outs), K), dtype=ws[0].dtype, device=ws[0].device)
            g = dict(cat=cat, cat_t=None, splits=n_outs, bias=None)
            pos = 0
            for m, w, b, n_o in zip(modules, ws, bs, n_outs):
                e = self._entry(m, w, b, dt, None, out_heads, out=cat[pos:pos + n_o])
                e["group"] = gkey
                pos += n_o
            es = [self.entries[(id(m), None, out_heads, dt)] for m in modules]
            if any(e["b_eff"] is not None for e in es):
                g["bias"] = torch.cat([e["b_eff"] if e["b_eff"] is not None else
                                       torch.zeros(n_o, dtype=cat.dtype, device=cat.device) for e, n_o in zip(es, n_outs)])
            self.groups[gkey] = g
        es = [self.entries[(id(m), None, out_heads, dt)] for m in modules]
        if need_dx and TRANSPOSED_DX and g["cat_t"] is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MergedWeights: a group's transposed scratch weight must exist before hipGraph capture")
            g["cat_t"] = torch.zeros((g["cat"].shape[1], g["cat"].shape[0]), dtype=g["cat"].dtype, device=g["cat"].device)
            pos = 0
            for e, n_o in zip(es, g["splits"]):
                self._add_transposed(e, out_t=g["cat_t"][:, pos:pos + n_o])
                pos += n_o
        g["entries"] = es
        return g

    def _create(self, module, w, b, in_heads, out_heads, out=None):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("MergedWeights: a new adapter / layout appeared during hipGraph capture; run the step "
                               "eagerly once first (GraphedForwardBackward's warm-up does)")
        N, K = w.shape
        n_out = out_heads[0] * out_heads[2] if out_heads else N
        k_out = in_heads[0] * in_heads[2] if in_heads else K
        w_eff = out if out is not None else torch.zeros((n_out, k_out), dtype=w.dtype, device=w.device)  # pads stay zero
        b_eff = None
        if b is not None:
            b_eff = pack_heads(b.detach(), out_heads).contiguous() if out_heads else b.detach()
        up, down = module.lora_up.weight, module.lora_down.weight
        step = w.dtype in (torch.bfloat16, torch.float16)  # the in-step merge kernel (csrc/merge_step.hip)
        e = dict(module=module, w=w.detach(), w_eff=w_eff, b_eff=b_eff, w_eff_t=None, w_t=None, scale=float(module.scale),
                 ptrs=(up.data_ptr(), down.data_ptr(), w.data_ptr()), in_heads=in_heads, out_heads=out_heads, step=step,
                 key=self._dither_key(module), group=None)
        if not step:  # f32 weights: the collapse kernel's sites (round 3's form; head sub-ranges, separate transposed site)
            heads_in = (in_heads[1], in_heads[2]) if in_heads else None
            sites = []
            if out_heads:
                h, d, D = out_heads
                for i in range(h):
                    sites.append((w.detach()[i * d:(i + 1) * d], w_eff[i * D:i * D + d], up.detach()[i * d:(i + 1) * d],
                                  down.detach()) + ((heads_in,) if heads_in else ()))
            else:
                sites.append((w.detach(), w_eff, up.detach(), down.detach()) + ((heads_in,) if heads_in else ()))
            e["sites"] = sites
        self._launch_one(e)  # this forward's values; the step plan is rebuilt
        self._plans = None
        return e

    def _dither_key(self, module) -> int:
        """The site key the dithered rounding hashes (csrc/merge_step.hip): one per ADAPTER, handed out in the order the
        adapters first run — the host model's execution order, the same whether sites are grouped or not — and kept when an
        entry is rebuilt (new layout, new scale, re-bound factors), so that the same run gives the same bits (ADVICE r4:
        ``len(entries) + 1`` could collide after a rebuild and depended on the registration order of LAYOUTS)."""
        return self._keys.setdefault(id(module), len(self._keys) + 1)

    def _msite(self, e) -> dict:
        m = e["module"]
        ih, oh = e["in_heads"], e["out_heads"]
        return dict(w=e["w"], up=m.lora_up.weight.detach(), down=m.lora_down.weight.detach(), out=e["w_eff"],
                    out_t=e["w_eff_t"], row_heads=(oh[1], oh[2]) if oh else None, col_heads=(ih[1], ih[2]) if ih else None,
                    key=e["key"])

    def _launch_one(self, e) -> None:
        if e["step"]:
            _C.MergeStepPlan([self._msite(e)]).launch(e["scale"], MERGE_ROUNDING)
        else:
            _C.MergePlan(e["sites"]).launch(e["scale"], _C.ROUND_ONCE)

    def _add_transposed(self, e, out_t=None) -> None:
        """Second scratch weight of the entry: W_eff^T.  16-bit weights: written by the SAME tile of the in-step merge that
        writes W_eff (one read of W, identical values).  f32 weights: a ``transposed`` site of the collapse kernel on a
        frozen transposed copy of W."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("MergedWeights: a site's transposed scratch weight must exist before hipGraph capture")
        w, module, out_heads = e["w"], e["module"], e["out_heads"]
        e["w_eff_t"] = out_t if out_t is not None else torch.zeros((e["w_eff"].shape[1], e["w_eff"].shape[0]),
                                                                   dtype=w.dtype, device=w.device)  # pads stay zero
        if not e["step"]:
            e["w_t"] = w.t().contiguous()  # frozen: built once
            heads = (out_heads[1], out_heads[2]) if out_heads else None
            e["sites"].append((e["w_t"], e["w_eff_t"], module.lora_down.weight.detach(), module.lora_up.weight.detach(),
                               heads, True))
        self._launch_one(e)
        self._plans = None

    def refresh(self) -> None:
        """ONE merge launch per (weight dtype, scale) group — one in practice — over every registered site."""
        if self._state is not None:
            self._fresh_at = self._state.step_count
        if not self.entries:
            return
        if self._plans is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MergedWeights.refresh: the site table must exist before hipGraph capture (run two "
                                   "eager steps first: the first registers the sites, the second builds the table)")
            step_groups, old_groups = {}, {}
            for e in self.entries.values():
                if e["step"]:
                    step_groups.setdefault((e["w"].dtype, e["scale"]), []).append(self._msite(e))
                else:
                    for st in e["sites"]:
                        old_groups.setdefault((st[0].dtype, e["scale"]), []).append(st)
            self._plans = [(_C.MergeStepPlan(sites), alpha, MERGE_ROUNDING) for (_, alpha), sites in step_groups.items()]
            self._plans += [(_C.MergePlan(sites), alpha, _C.ROUND_ONCE) for (_, alpha), sites in old_groups.items()]
        for plan, alpha, rounding in self._plans:
            plan.launch(alpha, rounding)
        self.refreshes += 1

    def owe(self, g2, x2, down, up, up_part, down_part, scale, g_heads, x_heads, kind="self", plan=None, drop=None) -> None:
        """``drop`` = (p, seed, offset) of nn.Dropout on the site's branch (matrix-core pass only): G enters masked."""
        self._owed.append((g2, x2, down, up, up_part, down_part, scale, g_heads, x_heads, kind, plan, drop))

    def _upload(self, key, raw: bytes, device, capturing: bool) -> torch.Tensor:
        """A site table -> device memory on the launch stream.  The table changes every eager step (fresh G / X
        addresses) and never under hipGraph replay (the capture's private pool hands out the same addresses): it is
        written into a persistent pinned buffer and copied to a persistent device buffer, which is a memcpy node of the
        captured graph."""
        key = key + (len(raw),)
        slot = self._tables.get(key)
        if slot is None:
            if capturing:
                raise RuntimeError("MergedWeights.flush_factors: the site table's buffers must exist before hipGraph "
                                   "capture (run the step eagerly once first)")
            mk = lambda: (torch.empty(len(raw), dtype=torch.uint8).pin_memory(),  # noqa: E731
                          torch.empty(len(raw), dtype=torch.uint8, device=device))
            # [eager pair, copy-done event, pairs set aside for captures (pinned memory cannot be allocated inside one)]
            slot = self._tables[key] = [mk(), None, [mk() for _ in range(4)]]
        if capturing:
            if not slot[2]:
                raise RuntimeError("MergedWeights.flush_factors: more than 4 hipGraph captures of this step shape")
            host, dev = slot[2].pop()  # owned by this graph from now on: its memcpy node re-reads `host` every replay
            self._graph_keep.append((host, dev))
        else:
            host, dev = slot[0]
            if slot[1] is not None:
                slot[1].synchronize()  # the previous step's copy has read the pinned buffer
        host.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dev.copy_(host, non_blocking=True)
        if not capturing:
            slot[1] = torch.cuda.Event()
            slot[1].record()
        return dev

    def _packs_of(self, down, up, dt, plan):
        """Persistent MFMA fragment packs (csrc/factor_mfma.hip) of one adapter's factors in the activation dtype."""
        key = (down.data_ptr(), up.data_ptr(), dt)
        pk = self._packs.get(key)
        if pk is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MergedWeights: a site's fragment packs must exist before hipGraph capture")
            pk = self._packs[key] = (torch.empty(int(plan.pack_down_elems), dtype=dt, device=down.device),
                                     torch.empty(int(plan.pack_up_elems), dtype=dt, device=down.device), down, up)
            self._pack_tables.clear()
        return pk

    def _pack_factors(self, dt, packs) -> None:
        """ONE launch: this step's f32 factors -> hi / lo fragment packs of every site of the matrix-core pass."""
        sig = (dt,) + tuple(id(p[0]) for p in packs)
        tab = self._pack_tables.get(sig)
        if tab is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("MergedWeights: the pack table must exist before hipGraph capture")
            arr, total = _C.factor_pack_table([(p[2], p[3], p[0], p[1]) for p in packs])
            tab = self._pack_tables[sig] = (_C.table_to_device(arr, packs[0][0].device), len(packs), total)
        _C.factor_pack(tab[0], tab[1], tab[2], dt)

    def flush_factors(self) -> None:
        """ONE factor-gradient launch per (pass, activation dtype, rank tile, masked or not) — one in practice — for
        every site whose backward ran since the last flush: ``lora_amd_linear_bwd_factors_mfma_ragged`` (16-bit
        activations: G and X read once, matrix cores; preceded by one ``lora_amd_factor_pack`` launch) or
        ``lora_amd_linear_bwd_factors_self_ragged`` (f32 activations, shapes the former does not take)."""
        if not self._owed:
            return
        owed, self._owed = self._owed, []
        groups = {}
        # one launch per register class of a (dtype, rank tile, masked) group: class 1 (the narrower operand's block fits 6
        # resident pairs per wave: every M = 16384 site) runs three workgroups per CU, class 2 two (csrc/factor_mfma.hip)
        one_class = not FM_TWO_CLASSES
        for st in owed:
            r, kind, plan = st[2].shape[0], st[9], st[10]
            rt = 4 if r <= 4 else 8 if r <= 8 else 16
            masked = kind == "mfma" and st[11] is not None and st[11][0] > 0.0
            # (register class, masked, block height): class 1 = one height per table (64), class 2 = one launch for both
            # heights (0: the workgroup enters its site's instantiation)
            cls = 0
            if kind == "mfma":
                c_ = 2 if one_class else int(plan.lds_class)
                cls = (c_, masked, int(plan.rows_per_block) if c_ == 1 else 0)
            groups.setdefault((kind, st[0].dtype, rt, cls), []).append(st)
        capturing = torch.cuda.is_current_stream_capturing()
        packed = {}  # activation dtype -> packs of this flush's sites, each adapter once
        for (kind, dt, rt, cls), sites in groups.items():
            if kind == "mfma":
                for st in sites:
                    pk = self._packs_of(st[2], st[3], dt, st[10])
                    packed.setdefault(dt, {})[id(pk[0])] = pk
        for dt, pks in packed.items():
            self._pack_factors(dt, list(pks.values()))
        launches = []   # (workgroups, closure): the tables go up on the launch stream first, the kernels may then run side by side
        for (kind, dt, rt, cls), sites in groups.items():
            dev0 = sites[0][0].device
            if kind == "mfma":
                # longest row blocks first: a block of the widest sites lives 3-5x longer than one of the square ones; in backward
                # order some of them start late and run alone in the launch's tail (FM_LONGEST_FIRST above).  Stable: equal
                # blocks keep the backward order.
                if FM_LONGEST_FIRST and (cls[0] != 1 or FM_LONGEST_FIRST_CLASS1):
                    sites = sorted(sites, key=lambda st: -int(st[10].rows_per_block) * (st[0].shape[1] + st[1].shape[1]))
                rows = []
                for (g2, x2, down, up, up_part, down_part, scale, g_heads, x_heads, _, plan, drop) in sites:
                    pk = self._packs_of(down, up, dt, plan)
                    rows.append((g2, x2, pk[0], pk[1], up_part, down_part, scale, g_heads, x_heads, down.shape[0],
                                 plan, drop))
                arr, grid = _C.factors_mfma_table(rows, dt, cls[0])
                # the block -> site map rides behind the table (one upload): a workgroup finds its site with one scalar load
                raw, moff = _C.factors_mfma_table_bytes(arr, grid) if FM_BLOCK_MAP else (bytes(arr), 0)
                dev = self._upload((kind, dt, rt, cls), raw, dev0, capturing)
                launches.append((grid, lambda dev=dev, n=len(sites), grid=grid, cls=cls, dt=dt, moff=moff:
                                 _C.linear_bwd_factors_mfma_ragged(dev, n, grid, cls[0], dt, cls[1], cls[2], moff)))
            else:
                arr, grid = _C.factors_self_ragged_table([st[:9] for st in sites], dt)
                dev = self._upload((kind, dt, rt, cls), bytes(arr), dev0, capturing)
                launches.append((grid, lambda dev=dev, n=len(sites), grid=grid, r=sites[0][2].shape[0], dt=dt:
                                 _C.linear_bwd_factors_self_ragged(dev, n, grid, r, dt)))
        # The launches of one flush are independent (disjoint slabs): the largest goes out on the launch stream, the others on
        # side streams forked from it and joined behind it, so that the tail of one launch fills with the workgroups of the next
        # instead of three tails in a row (round 6: one launch per register class and block height = three launches in the
        # headline step).  Capturable: the fork / join become edges of the hipGraph.
        launches.sort(key=lambda t: -t[0])
        if len(launches) == 1 or not CONCURRENT_FACTOR_LAUNCHES:
            for _, fn in launches:
                fn()
            return
        main = torch.cuda.current_stream()
        side = self.__dict__.setdefault("_side_streams", [])
        while len(side) < len(launches) - 1:
            side.append(torch.cuda.Stream(device=main.device))
        for st_ in side[:len(launches) - 1]:
            st_.wait_stream(main)
        launches[0][1]()
        for (_, fn), st_ in zip(launches[1:], side):
            with torch.cuda.stream(st_):
                fn()
        for st_ in side[:len(launches) - 1]:
            main.wait_stream(st_)

    def invalidate(self) -> None:
        """Factor tensors were re-bound (new storage) or a scale changed: rebuild the entries on their next use."""
        self.entries.clear()
        self.groups.clear()
        self._plans = None

    @property
    def bytes_algorithmic(self) -> int:
        return sum(p[0].bytes_algorithmic for p in (self._plans or []))


def _merged_factor_grads(g2, x2, down, up, scale, sink, out_heads, in_heads, K, N, tag):
    """Both factor gradients of one merged-weight site: deferred to the step's one-launch pass (trainer state with
    ``MergedWeights.defer_factors``: the matrix-core pass for 16-bit activations, the VALU pass otherwise), or one
    ``linear_bwd_factors_self`` launch here.  Returns (d_down, d_up) — None when they leave through ``sink``."""
    M, r = x2.shape[0], down.shape[0]
    mw = getattr(getattr(sink, "owner", None), "merged", None)
    defer = mw is not None and mw.defer_factors
    plan, kind = None, "self"
    if defer and _C.FACTORS_MFMA_MODE == "all":
        plan = _C.factors_mfma_plan(M, K, N, r, g2.dtype)
        kind = "mfma" if plan.supported else "self"
    if kind == "self":
        plan = _C.factors_self_plan(M, K, N, r, _C.SELF_ROWS_DEFERRED if defer else 0)
    key = (kind, M, K, N, r, int(plan.nparts))
    if sink is not None:
        if sink.pending is not None:
            sink.flush()
        up_part, down_part = sink.self_workspace(key, plan, g2.device)
    else:
        up_part, down_part = (torch.empty(max(int(n), 1), dtype=torch.float32, device=g2.device)
                              for n in (plan.up_part_floats, plan.down_part_floats))
    if defer:
        mw.owe(g2, x2, down.contiguous(), up.contiguous(), up_part, down_part, scale, out_heads, in_heads, kind, plan)
        _log("bwd", f"{tag}_dx+factors_deferred_{kind}", M, K, N, r)
    else:
        _C.linear_bwd_factors_self(g2, x2, down.contiguous(), up.contiguous(), up_part, down_part, scale,
                                   g_heads=out_heads, x_heads=in_heads)
        _log("bwd", f"{tag}_dx+factors_self", M, K, N, r)
    if sink is not None:
        sink.pending = key
        return None, None
    d_up = torch.empty((N, r), dtype=torch.float32, device=g2.device)
    d_down = torch.empty((r, K), dtype=torch.float32, device=g2.device)
    rows = [(up_part, d_up, plan.nparts, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
            (down_part, d_down, plan.nparts, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    table, n, total = _C.make_reduce_table(rows, g2.device)
    _C.reduce_batched(table, n, total)
    return d_down.to(down.dtype), d_up.to(up.dtype)


class LoraLinearMergedFunction(torch.autograd.Function):
    """lora.py:53-58 (dropout not in effect) on the step's merged weight — see :class:`MergedWeights`.
    Saves X only.  The factor gradients leave as per-row-block partials for the batched reduce (``sink``) or are
    reduced here."""

    @staticmethod
    def forward(ctx, x, w_eff, b_eff, down, up, scale, sink, in_heads, out_heads, w_eff_t=None):
        _C.require()
        x2 = _rows2d(x, w_eff.shape[1])
        with _gemm_range():
            y = F.linear(x2, w_eff, b_eff)  # frozen dense GEMM (MFMA, hipBLASLt) on W + scale up down
        K = in_heads[0] * in_heads[1] if in_heads else w_eff.shape[1]
        N = out_heads[0] * out_heads[1] if out_heads else w_eff.shape[0]
        # down / up None: a FROZEN site in the merged path's layout (standin/frozen.py, bench.py's `frozen_only` leg)
        _log("fwd", "merged" + ("_heads" if (in_heads or out_heads) else ""), x2.shape[0], K, N,
             down.shape[0] if down is not None else 0)
        ctx.save_for_backward(x2, w_eff, down, up)
        ctx.w_eff_t = w_eff_t  # frozen for the step (never an autograd input)
        ctx.scale, ctx.sink, ctx.x_shape = float(scale), sink, x.shape
        ctx.in_heads, ctx.out_heads, ctx.dims = in_heads, out_heads, (K, N)
        return y.view(*x.shape[:-1], y.shape[1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x2, w_eff, down, up = ctx.saved_tensors
        K, N = ctx.dims
        g2 = _rows2d(g, w_eff.shape[0])
        need_x, _, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        sink = ctx.sink
        dx = None
        if need_x:
            with _gemm_range():  # frozen dense GEMM; pad columns of a head-padded dX come out zero
                dx = (F.linear(g2, ctx.w_eff_t) if ctx.w_eff_t is not None else g2 @ w_eff).view(ctx.x_shape)
        d_down = d_up = None
        if need_down or need_up:
            d_down, d_up = _merged_factor_grads(g2, x2, down, up, ctx.scale, sink, ctx.out_heads, ctx.in_heads, K, N,
                                                "merged")
        db = None
        if need_b:
            db = (unpack_heads(g2, ctx.out_heads) if ctx.out_heads else g2).sum(0)
            if ctx.out_heads:
                db = pack_heads(db, ctx.out_heads)
        return dx, None, db, d_down, d_up, None, None, None, None, None


class LoraLinearMergedGroupFunction(torch.autograd.Function):
    """Several merged-weight sites on ONE input (attn1's to_q / to_k / to_v, attn2's to_k / to_v) as one autograd node:
    the input gradients of the sites are accumulated by the GEMMs themselves (``addmm_``: beta = 1 in the library's
    epilogue) instead of n - 1 elementwise ``add`` launches over [M, K] that autograd would issue for n separate nodes.

    Inputs: x, n, cat, bias_cat, then per site (w_eff, b_eff, down, up, scale, sink, out_heads, w_eff_t).  ``cat`` (or
    None): the sites' scratch weights as row ranges of ONE buffer (``MergedWeights.lookup_group``) — the forward is then
    one GEMM whose output columns are the sites' outputs (views, no copies)."""

    @staticmethod
    def forward(ctx, x, n, cat, bias_cat, *args):
        _C.require()
        sites = [args[8 * i:8 * i + 8] for i in range(n)]
        K = sites[0][0].shape[1]
        x2 = _rows2d(x, K)
        outs = []
        with _gemm_range():
            ycat, pos = (F.linear(x2, cat, bias_cat) if cat is not None else None), 0
            for (w_eff, b_eff, down, up, scale, sink, out_heads, _) in sites:
                if ycat is not None:
                    y = ycat[:, pos:pos + w_eff.shape[0]]
                    pos += w_eff.shape[0]
                else:
                    y = F.linear(x2, w_eff, b_eff)
                outs.append(y.view(*x.shape[:-1], y.shape[1]))
                N = out_heads[0] * out_heads[1] if out_heads else w_eff.shape[0]
                _log("fwd", "merged_group" + ("_cat" if cat is not None else "") + ("_heads" if out_heads else ""),
                     x2.shape[0], K, N, down.shape[0] if down is not None else 0)
        ctx.save_for_backward(x2, *[t for st in sites for t in (st[0], st[2], st[3])])
        ctx.meta = [(float(st[4]), st[5], st[6], st[7]) for st in sites]
        ctx.n, ctx.x_shape = n, x.shape
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        x2 = saved[0]
        M, K = x2.shape
        need_x = ctx.needs_input_grad[0]
        dx = None
        grads = [None, None, None, None]
        for i in range(ctx.n):
            w_eff, down, up = saved[1 + 3 * i:4 + 3 * i]
            scale, sink, out_heads, w_eff_t = ctx.meta[i]
            g = gs[i]
            N = out_heads[0] * out_heads[1] if out_heads else w_eff.shape[0]
            d_down = d_up = None
            if g is not None:
                g2 = _rows2d(g, w_eff.shape[0])
                if need_x:
                    with _gemm_range():
                        wb = w_eff_t.t() if w_eff_t is not None else w_eff  # [N', K] operand; transposed storage: TN GEMM
                        dx = (g2 @ wb) if dx is None else dx.addmm_(g2, wb)
                if down is not None:   # None: a frozen site (standin/frozen.py)
                    d_down, d_up = _merged_factor_grads(g2, x2, down, up, scale, sink, out_heads, None, K, N, "merged_group")
            grads += [None, None, d_down, d_up, None, None, None, None]
        grads[0] = dx.view(ctx.x_shape) if dx is not None else None
        return tuple(grads)


def merged_ok(x: torch.Tensor, weight: torch.Tensor, down: torch.Tensor, up: torch.Tensor, sel, dropout_p: float,
              in_heads, out_heads, bias=None) -> bool:
    """Can this call take the merged-weight path?  (device, no dropout / selector, frozen weight AND bias, f32 factors, a
    shape and alignment the merge and the one-launch factor-gradient kernels cover, 16-bit or f32 activations matching
    the weight).  Mirrors the planners' preconditions so that a site they would refuse takes the per-site kernels
    instead of raising inside the forward."""
    if not x.is_cuda or dropout_p > 0.0 or sel is not None or weight.requires_grad:
        return False
    if bias is not None and bias.requires_grad:  # the merged path caches the bias and returns no gradient for it
        return False
    if (weight.data_ptr() | down.data_ptr() | up.data_ptr()) % 16 or not weight.is_contiguous():
        return False
    if weight.dtype == torch.float32 and (in_heads is not None or out_heads is not None) and \
            (weight.shape[0] % 32 or weight.shape[1] % 32):
        return False  # head layouts of f32 weights run on the collapse kernel's column-owner tiles
    if down.dtype != torch.float32 or up.dtype != torch.float32 or x.dtype != weight.dtype:
        return False
    N, K = weight.shape
    r = down.shape[0]
    kw = in_heads[0] * in_heads[2] if in_heads else K
    if x.shape[-1] != kw or x.numel() == 0:
        return False
    M = x.numel() // kw
    for hl in (in_heads, out_heads):
        if hl is not None and (hl[1] % 8 or hl[2] % 8 or hl[2] < hl[1]):
            return False
    return bool(_C.factors_self_plan(M, K, N, r).supported)


def pack_heads(y: torch.Tensor, lay) -> torch.Tensor:
    """[..., heads*d] -> [..., heads*D]: every head's d columns followed by D-d zeros (the layout the attention kernels
    want for head sizes 40 / 80)."""
    h, d, D = lay
    return F.pad(y.unflatten(-1, (h, d)), (0, D - d)).flatten(-2)


def unpack_heads(x: torch.Tensor, lay) -> torch.Tensor:
    """[..., heads*D] -> [..., heads*d] (a copy)."""
    h, d, D = lay
    return x.unflatten(-1, (h, D))[..., :d].flatten(-2)


class LoraLinearHeadsFunction(torch.autograd.Function):
    """:class:`LoraLinearFunction` for head-padded activations on the fused MFMA kernels: the X operand and / or the
    output (and, in the backward, G and / or dX) are read and written in the padded layout by the kernel itself, so the
    pad / slice copies around the attention core disappear.  Only entered when the forward tile for the shape is known;
    a backward whose fused tile is not known (yet) detours through dense copies and the regular backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, down, up, scale, sink, in_heads, out_heads, tile):
        _C.require()
        N, K = weight.shape
        x2 = _rows2d(x, _C.heads_width(K, in_heads))
        y, t = _C.linear_gemm_fwd(x2, weight, bias, down.contiguous(), up.contiguous(), scale, tile,
                                  x_heads=in_heads, y_heads=out_heads)
        _log("fwd", f"ring{tile}_heads", x2.shape[0], K, N, down.shape[0])
        ctx.save_for_backward(x2, weight, down, up, t)
        ctx.scale, ctx.has_bias, ctx.x_shape, ctx.sink = float(scale), bias is not None, x.shape, sink
        ctx.in_heads, ctx.out_heads = in_heads, out_heads
        return y.view(*x.shape[:-1], y.shape[1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x2, weight, down, up, t = ctx.saved_tensors
        N, K = weight.shape
        r, M = down.shape[0], x2.shape[0]
        g2 = _rows2d(g, _C.heads_width(N, ctx.out_heads))
        need_x, need_w, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        s, sink = ctx.scale, ctx.sink
        plan = _C.linear_plan(M, K, N, r)
        tile = _C.gemm_choice_bwd_cached(M, K, N, r, g2.dtype)
        if not (tile and plan.fused and _C._rows_ok(g2) and weight.is_contiguous() and _C.heads_tile_ok(ctx.in_heads)
                and not need_w):
            # dense detour: unpacked copies through the regular backward (which also times the fused candidates, so the
            # next backward of this shape stays in the padded layout)
            from types import SimpleNamespace

            x_log = unpack_heads(x2, ctx.in_heads) if ctx.in_heads else x2
            g_log = unpack_heads(g2, ctx.out_heads) if ctx.out_heads else g2
            fake = SimpleNamespace(saved_tensors=(x_log, weight, down, up, t, None),
                                   needs_input_grad=tuple(ctx.needs_input_grad[:5]) + (False,) * 4, scale=s, p=0.0,
                                   seed=0, off=0, sink=sink, fused=_C.fused_ok(x_log, N, r), has_bias=ctx.has_bias,
                                   x_shape=x_log.shape)
            dx, dw, db, d_down, d_up = LoraLinearFunction.backward(fake, g_log)[:5]
            if dx is not None:
                if ctx.in_heads:
                    dx = pack_heads(dx, ctx.in_heads)
                dx = dx.view(*ctx.x_shape[:-1], dx.shape[-1])
            return dx, dw, db, d_down, d_up, None, None, None, None, None
        key = (M, K, N, r)
        if sink is not None:
            if sink.pending is not None:
                sink.flush()
            _, up_part, down_part = sink.workspace(key, plan, g2.device)
        else:
            up_part, down_part = (torch.empty(max(int(n), 1), dtype=torch.float32, device=g2.device)
                                  for n in (plan.up_part_floats, plan.down_part_floats))
        down_c, up_c = down.contiguous(), up.contiguous()
        dx2, gt = _C.linear_gemm_dx(g2, _C.weight_t(weight), down_c, up_c, s, tile, g_heads=ctx.out_heads,
                                    dx_heads=ctx.in_heads)
        _log("bwd", f"ring{tile}_heads_dx+factors", M, K, N, r)
        _C.linear_bwd_factors(g2, t, up_part, x2, gt, down_part, r, s, None, g_heads=ctx.out_heads,
                              x_heads=ctx.in_heads)
        dx = dx2.view(*ctx.x_shape[:-1], dx2.shape[1]) if need_x else None
        d_up = d_down = None
        if sink is not None:
            sink.pending = key
        else:
            d_up = torch.empty((N, r), dtype=torch.float32, device=g2.device)
            d_down = torch.empty((r, K), dtype=torch.float32, device=g2.device)
            rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                    (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
            table, n, total = _C.make_reduce_table(rows, g2.device)
            _C.reduce_batched(table, n, total)
            d_up, d_down = d_up.to(up.dtype), d_down.to(down.dtype)
        db = None
        if ctx.has_bias and need_b:
            db = (unpack_heads(g2, ctx.out_heads) if ctx.out_heads else g2).sum(0)
        return dx, None, db, d_down, d_up, None, None, None, None, None


def ws_heads_route_ok(x, weight, down, up, sel, dropout_p, in_heads, out_heads) -> bool:
    """May a dropout site with head-padded input OR output run ``LoraLinearFunction`` in those layouts: the forward is the
    weight-stationary kernel's (the route ``WS_DROPOUT_WIDE`` takes for dense rows too), contraction and panel widths fit
    (``_C.ws_heads_ok``).  The backward decides for itself and can always detour through dense copies."""
    if not (WS_HEADS and WS_DROPOUT and WS_DROPOUT_WIDE and x.is_cuda and dropout_p > 0.0 and sel is None):
        return False
    if (in_heads is None) == (out_heads is None):
        return False
    N, K = weight.shape
    r = down.shape[0]
    if not (down.dtype == torch.float32 and up.dtype == torch.float32 and x.dtype in (torch.bfloat16, torch.float16)
            and weight.dtype == x.dtype and weight.is_contiguous() and N % 8 == 0 and not (K == 1280 and N >= 4 * K)):
        return False
    kw = _C.heads_width(K, in_heads)
    if x.shape[-1] != kw or x.numel() == 0:
        return False
    x2 = x.reshape(-1, kw)
    return bool(_C.ws_supported(x2, K, N, r) and _C.ws_heads_ok(K, N, in_heads, out_heads))


def lora_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], down: torch.Tensor,
                up: torch.Tensor, sel: Optional[torch.Tensor], scale: float, dropout_p: float,
                sink: Optional[GradSink] = None, in_heads=None, out_heads=None) -> torch.Tensor:
    """``in_heads`` / ``out_heads`` = (heads, d, D): x arrives / y leaves with every head's d columns padded to D."""
    if in_heads is None and out_heads is None:
        return LoraLinearFunction.apply(x, weight, bias, down, up, sel, float(scale), float(dropout_p), sink)
    N, K = weight.shape
    r = down.shape[0]
    M = x.numel() // _C.heads_width(K, in_heads)
    tile = 0
    if (x.is_cuda and dropout_p == 0.0 and sel is None and down.dtype == torch.float32 and up.dtype == torch.float32
            and x.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x.dtype and weight.is_contiguous()
            and _C.heads_tile_ok(out_heads) and (in_heads is None or in_heads[1] % 8 == 0)):
        tile = _C.gemm_choice_cached(M, K, N, r, x.dtype, bias is not None) or 0
    if tile:
        return LoraLinearHeadsFunction.apply(x, weight, bias, down, up, float(scale), sink, in_heads, out_heads, tile)
    if ws_heads_route_ok(x, weight, down, up, sel, dropout_p, in_heads, out_heads):
        # nn.Dropout on the branch (configs[3]): the weight-stationary kernel in the padded layouts, both directions
        return LoraLinearFunction.apply(x, weight, bias, down, up, None, float(scale), float(dropout_p), sink, in_heads,
                                        out_heads)
    # dense detour (CPU tensors, shapes the fused kernel does not take, or a shape not timed yet: the regular function
    # times it, so the next call stays in the padded layout)
    y = LoraLinearFunction.apply(unpack_heads(x, in_heads) if in_heads else x, weight, bias, down, up, sel,
                                 float(scale), float(dropout_p), sink)
    return pack_heads(y, out_heads) if out_heads else y


class LoraLinearGroupFunction(torch.autograd.Function):
    """Several LoraInjectedLinear sites applied to ONE input (attn1's to_q / to_k / to_v, attn2's to_k / to_v: the
    reference calls lora.py:53-58 once per site on the same tensor) as one weight-stationary launch forward
    (``_C.linear_ws``: the input is fetched from HBM once) and one autograd node backward, which also sums the sites'
    input gradients inside the kernels (accumulate flag) instead of leaving n - 1 ``add`` launches to autograd.

    Inputs: x, n, then per site (weight, bias, down, up, scale, sink).  Eligibility is the caller's job
    (``lora.lora_linear_group``): no dropout, no selector, f32 factors, 16-bit activations, supported K and N."""

    @staticmethod
    def forward(ctx, x, n, *args):
        _C.require()
        sites = [args[6 * i:6 * i + 6] for i in range(n)]
        K = sites[0][0].shape[1]
        x2 = _rows2d(x, K)
        descs = [dict(wp=_C.ws_pack(w), N=w.shape[0], bias=b, down=d.contiguous(), up=u.contiguous(), scale=float(sc))
                 for (w, b, d, u, sc, _) in sites]
        outs = _C.linear_ws(x2, descs)
        for d_ in descs:
            _log("fwd", f"ws_group{n}", x2.shape[0], K, d_["N"], d_["down"].shape[0])
        ctx.save_for_backward(x2, *[t for _, t in outs], *[a for s_ in sites for a in (s_[0], s_[2], s_[3])])
        ctx.n, ctx.x_shape = n, x.shape
        ctx.meta = [(float(sc), sink, b is not None) for (_, b, _, _, sc, sink) in sites]
        return tuple(y.view(*x.shape[:-1], y.shape[1]) for y, _ in outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        n = ctx.n
        saved = ctx.saved_tensors
        x2, ts, rest = saved[0], saved[1:1 + n], saved[1 + n:]
        M, K = x2.shape
        need_x = ctx.needs_input_grad[0]
        dx2 = None
        out = [None, None]
        per_site = []
        for i in range(n):
            weight, down, up = rest[3 * i:3 * i + 3]
            scale, sink, has_bias = ctx.meta[i]
            g = grads[i]
            base = 2 + 6 * i
            need_w, need_b, need_down, need_up = ctx.needs_input_grad[base:base + 4]
            d_down = d_up = dw = db = None
            if g is not None:
                N, r = weight.shape[0], down.shape[0]
                g2 = _rows2d(g, N)
                if not _C._rows_ok(g2):
                    g2 = g2.contiguous()
                down_c, up_c = down.contiguous(), up.contiguous()
                key = (M, K, N, r)
                plan = _C.linear_plan(*key)
                if sink is not None:
                    if sink.pending is not None:
                        sink.flush()
                    gt_part, up_part, down_part = sink.workspace(key, plan, g2.device)
                else:
                    gt_part, up_part, down_part = (torch.empty(max(int(k_), 1), dtype=torch.float32, device=g2.device)
                                                   for k_ in (plan.gt_part_floats, plan.up_part_floats,
                                                              plan.down_part_floats))
                # dX (+)= G W + s (G up) down and Gt = s G up: weight-stationary on W^T, accumulating across the sites
                first = dx2 is None
                if first:
                    dx2 = torch.empty((M, K), dtype=x2.dtype, device=x2.device)
                (_, gt), = _C.linear_ws(g2, [dict(wp=_C.ws_pack(weight, True), N=K, down=up_c, up=down_c, scale=scale,
                                                  t_scale=scale, flayout=3 if first else 7, y=dx2)])
                _C.linear_bwd_factors(g2, ts[i], up_part, x2, gt, down_part, r, scale)
                _log("bwd", "ws_dx+factors", M, K, N, r)
                if sink is not None:
                    sink.pending = key
                else:
                    d_up = torch.empty((N, r), dtype=torch.float32, device=g2.device)
                    d_down = torch.empty((r, K), dtype=torch.float32, device=g2.device)
                    rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                            (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
                    table, nn_, total = _C.make_reduce_table(rows, g2.device)
                    _C.reduce_batched(table, nn_, total)
                    d_up, d_down = d_up.to(up.dtype), d_down.to(down.dtype)
                if need_w:
                    dw = g2.t() @ x2
                if has_bias and need_b:
                    db = g2.sum(0)
            per_site += [dw, db, d_down, d_up, None, None]
        if need_x and dx2 is not None:
            out[0] = dx2.view(ctx.x_shape)
        return tuple(out + per_site)


def lora_linear_group(x: torch.Tensor, sites) -> tuple:
    """``sites``: [(weight, bias, down, up, scale, sink), ...] all applied to ``x``; returns one output per site."""
    flat = [a for s_ in sites for a in s_]
    return LoraLinearGroupFunction.apply(x, len(sites), *flat)


def linear_group_ok(x: torch.Tensor, shapes, r: int) -> bool:
    """Can ``lora_linear_group`` run sites of (N, K) ``shapes`` on ``x``: weight-stationary forward (contraction K) and
    input gradient (contraction N), plus the one-launch factor-gradient pass."""
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or not 1 <= len(shapes) <= _C.WS_MAX_SITES:
        return False
    K = shapes[0][1]
    x2 = x.reshape(-1, K) if x.shape[-1] == K else None
    if x2 is None or any(k != K for _, k in shapes):
        return False
    return all(_C.ws_supported(x2, K, N, r) and N in _C._WS_K and K % 4 == 0 and _C.fused_ok(x2, N, r)
               for N, _ in shapes)


class LoraConvUpFunction(torch.autograd.Function):
    """``y0 += scale * dropout(conv1x1(t; up))`` in NCHW, in place on the frozen conv's output
    (lora.py:116-123, 130-135).  Per sample, Y_b viewed as [C_out, H*W] gets the rank-r update
    ``up[C_out, r] @ T_b[r, H*W]`` from the same rank_update kernel as the Linear path (roles of the
    "row vector" and the "factor" swapped), so no [B, C_out, H, W] branch tensor is ever written.
    """

    @staticmethod
    def forward(ctx, y0, t, up, scale, dropout_p):
        _C.require()
        B, Co, H, W = y0.shape
        r = t.shape[1]
        if not y0.is_contiguous():
            raise ValueError("lora_conv: frozen conv output must be NCHW-contiguous")
        t = t.contiguous()
        up2 = up.reshape(Co, r).to(torch.float32).contiguous()
        streams = []
        for b in range(B):
            seed = off = 0
            if dropout_p > 0.0:
                seed, off = next_dropout_stream(y0.device)
            streams.append((seed, off))
            rank_update_any_(y0[b].view(Co, H * W), up2, t[b].view(r, H * W), _C.FACTOR_RK, scale, dropout_p,
                             seed, off)
        ctx.mark_dirty(y0)
        ctx.save_for_backward(t, up2)
        ctx.scale, ctx.p, ctx.streams = float(scale), float(dropout_p), streams
        ctx.up_shape, ctx.up_dtype = up.shape, up.dtype
        return y0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        t, up2 = ctx.saved_tensors
        B, Co, H, W = g.shape
        r = t.shape[1]
        g = g.contiguous()
        need_t, need_up = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dt = torch.empty((B, r, H * W), dtype=torch.float32, device=g.device) if need_t else None
        dup = None
        for b in range(B):
            seed, off = ctx.streams[b]
            gb = g[b].view(Co, H * W)
            if need_t:  # dT_b = scale * up^T @ (G_b .* mask)
                colreduce_any(gb, up2, _C.FACTOR_RK, ctx.scale, out=dt[b], beta=0.0, p=ctx.p, seed=seed, off=off)
            if need_up:  # dUp += scale * (G_b .* mask) @ T_b^T
                part = rowdot_any(gb, t[b].view(r, H * W), _C.FACTOR_RK, ctx.scale, None, False, ctx.p, seed, off)
                dup = part if dup is None else dup + part
        dt_out = dt.view(B, r, H, W).to(t.dtype) if need_t else None
        dup_out = dup.view(ctx.up_shape).to(ctx.up_dtype) if need_up else None
        return g, dt_out, dup_out, None, None


class LoraConvFunction(torch.autograd.Function):
    """``y = conv(x; W) + b + scale * dropout(conv1x1(S . conv_kxk(x; down); up))`` (lora.py:130-135) and its
    gradient for the native geometry (stride 1, "same" 1x1 / 3x3, groups 1, NCHW; ``_C.conv_plan(...).native``).

    The frozen convolution and its input gradient are MIOpen calls; everything low-rank is csrc/conv.hip:
    forward 2 launches (+1 tiny finalize when the channel loop is split), backward 2 (+1), the parameter
    gradients leave as per-group partials for the batched reduce.  Saves X and T [B,r,H,W] f32 only."""

    @staticmethod
    def forward(ctx, x, weight, bias, down, up, sel, ks, scale, dropout_p, sink):
        _C.require()
        B, Ci, H, W = x.shape
        Co, r = weight.shape[0], down.shape[0]
        pad = (ks - 1) // 2
        x = x.contiguous()
        y = F.conv2d(x, weight, bias, 1, pad)  # frozen dense conv (MIOpen, MFMA)
        if not y.is_contiguous():
            y = y.contiguous()
        plan = _C.conv_plan(B, Ci, Co, H, W, ks, r)
        key = ("conv", B, Ci, Co, H, W, ks, r)
        t_part = sink.conv_workspace(key, plan, x.device)[0] if sink is not None else \
            torch.empty(max(int(plan.t_part_floats), 1), dtype=torch.float32, device=x.device)
        t = torch.empty((B, r, H, W), dtype=torch.float32, device=x.device)
        down_c, up_c = down.float().contiguous(), up.float().contiguous()  # f32 masters: no-ops in training
        sel_c = sel.to(torch.float32).contiguous() if sel is not None else None
        _C.conv_down_fwd(x, down_c, sel_c, t_part, t, ks)
        seed = off = 0
        if dropout_p > 0.0:
            seed, off = next_dropout_stream(x.device)
        _C.conv_up_fwd_(y, t, up_c, scale, dropout_p, seed, off)
        ctx.save_for_backward(x, weight, down, up, t, sel_c)
        ctx.ks, ctx.scale, ctx.p, ctx.seed, ctx.off, ctx.sink = ks, float(scale), float(dropout_p), seed, off, sink
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, weight, down, up, t, sel = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co, r, ks = weight.shape[0], down.shape[0], ctx.ks
        pad = (ks - 1) // 2
        need_x, need_w, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        g = g.contiguous()
        plan = _C.conv_plan(B, Ci, Co, H, W, ks, r)
        key = ("conv", B, Ci, Co, H, W, ks, r)
        sink = ctx.sink
        if sink is not None:
            if sink.pending is not None:
                sink.flush()
            bufs = sink.conv_workspace(key, plan, g.device)
        else:
            bufs = conv_buffers(plan, B, r, H * W, g.device)
        _, gt_part, gt, up_part, down_part = bufs
        down_c, up_c = down.float().contiguous(), up.float().contiguous()
        _C.conv_bwd_g(g, t, up_c, sel, gt_part, gt, up_part, ctx.scale, ctx.p, ctx.seed, ctx.off)
        dx = None
        if need_x:  # frozen dense conv's input gradient (MIOpen), then the low-rank term is added in place
            dx = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
            if not dx.is_contiguous():
                dx = dx.contiguous()
        _C.conv_bwd_x(x, dx, gt, down_c, down_part, ks)
        d_down = d_up = None
        if sink is not None:
            sink.pending = key
        else:
            d_up = torch.empty(up.shape, dtype=torch.float32, device=g.device)
            d_down = torch.empty(down.shape, dtype=torch.float32, device=g.device)
            table, n, total = _C.make_reduce_table(conv_reduce_rows(bufs, plan, Ci, Co, ks, r, d_up, d_down, 0.0),
                                                   g.device)
            _C.reduce_batched(table, n, total)
            d_up, d_down = d_up.to(up.dtype), d_down.to(down.dtype)
        dw = db = None
        if need_w:
            dw = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        if ctx.has_bias and need_b:
            db = g.sum((0, 2, 3))
        return dx, dw, db, d_down, d_up, None, None, None, None, None


class ConvSitePacks:
    """Fragment packs (pf, pd, pu) and arrival counters of every channels-last 3x3 adapter site, refreshed by ONE
    ``lora_amd_conv3_nhwc_pack_batched`` launch per optimiser step — the factors change once per step, not per forward
    (rounds 3-5 re-packed ``down`` in every forward of every site: 44 launches a step at configs[3]).

    A site registers on its first forward (packed alone, on the spot).  From then on the first conv forward after an
    optimiser step (``owner.step_count`` moved: ``FlatLoraState``'s kernels write the factors through raw pointers, so
    tensor versions do not see them) launches the table of ALL registered sites; without an owner a site is re-packed
    alone when one of its factors' ``_version`` changed.  Tables and packs are persistent: hipGraph-replayable."""

    def __init__(self, owner=None):
        self.owner = owner
        self.sites = {}      # (down ptr, up ptr, dtype) -> record
        self._table = None   # (device table, n, total) per activation dtype
        self._fresh_at = None

    def _pack(self, recs, dt):
        arr, total = _C.conv3_nhwc_pack_table([(q["down"], q["up"], q["pf"], q["pd"], q["pu"]) for q in recs])
        return _C.table_to_device(arr, recs[0]["pf"].device), len(recs), total

    def refresh(self) -> None:
        if not self.sites:
            return
        if self._table is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ConvSitePacks: the site table must exist before hipGraph capture (run the step eagerly first)")
            by_dt = {}
            for q in self.sites.values():
                by_dt.setdefault(q["dt"], []).append(q)
            self._table = {dt: self._pack(recs, dt) for dt, recs in by_dt.items()}
        for dt, (tab, n, total) in self._table.items():
            _C.conv3_nhwc_pack_batched(tab, n, total, dt)
        self._fresh_at = self._stamp()

    def _stamp(self):
        return getattr(self.owner, "step_count", None)

    def get(self, down: torch.Tensor, up2: torch.Tensor, dt, plan, lplan_blocks: int):
        """(pf, pd, pu, fwd counters, bwd counters) of the site whose f32 factors are ``down`` [r, C_in, 3, 3], ``up2`` [C_out, r]."""
        key = (down.data_ptr(), up2.data_ptr(), dt)
        q = self.sites.get(key)
        if q is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ConvSitePacks: a new conv site appeared during hipGraph capture (run the step eagerly first)")
            dev = down.device
            q = self.sites[key] = dict(down=down, up=up2, dt=dt, ver=(down._version, up2._version),
                                       pf=torch.empty(int(plan.pf_elems), dtype=dt, device=dev),
                                       pd=torch.empty(int(plan.pd_elems), dtype=dt, device=dev),
                                       pu=torch.empty(up2.shape[0] * 32, dtype=dt, device=dev), cnt={})
            self._table = None
            tab, n, total = self._pack([q], dt)
            _C.conv3_nhwc_pack_batched(tab, n, total, dt)
        elif self.owner is not None:
            if self._fresh_at != self._stamp():
                self.refresh()
        elif q["ver"] != (down._version, up2._version):   # no trainer state: torch's own optimisers bump the version
            tab, n, total = self._pack([q], dt)
            _C.conv3_nhwc_pack_batched(tab, n, total, dt)
            q["ver"] = (down._version, up2._version)
        ck = (int(plan.fwd_tiles), int(lplan_blocks))
        cnt = q["cnt"].get(ck)
        if cnt is None:   # zeroed once: the last arriver of a tile / row block resets its counter
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ConvSitePacks: a site's counters must exist before hipGraph capture")
            cnt = q["cnt"][ck] = (torch.zeros(max(ck[0], 1), dtype=torch.int32, device=down.device),
                                  torch.zeros(max(ck[1], 1), dtype=torch.int32, device=down.device))
        return q["pf"], q["pd"], q["pu"], cnt[0], cnt[1]


def conv_pack_registries(*models):
    """The ConvSitePacks of the trainer states that own ``models``' conv adapters (found through the adapters' gradient
    sinks; cached on the first model once non-empty): ``trainer.forward_backward`` refreshes them at the top of a step, so
    that the ONE pack launch is the step's (and a captured graph's) first conv-adapter node."""
    if not models or models[0] is None:
        return ()
    cached = models[0].__dict__.get("_conv_pack_regs")
    if cached:
        return cached
    regs = []
    for m in models:
        if m is None:
            continue
        for mod in m.modules():
            sink = mod.__dict__.get("_grad_sink")
            reg = getattr(getattr(sink, "owner", None), "__dict__", {}).get("_conv_packs") if sink is not None else None
            if reg is not None and reg not in regs:
                regs.append(reg)
    if regs:
        models[0].__dict__["_conv_pack_regs"] = tuple(regs)
    return tuple(regs)


_CONV_PACKS_DEFAULT = ConvSitePacks()   # sites without a trainer state (inference through monkeypatch_*, plain torch optimisers)


def conv_site_packs(sink) -> ConvSitePacks:
    owner = getattr(sink, "owner", None)
    if owner is None:
        return _CONV_PACKS_DEFAULT
    reg = owner.__dict__.get("_conv_packs")
    if reg is None:
        reg = owner.__dict__["_conv_packs"] = ConvSitePacks(owner)
    return reg


class LoraConv3NhwcFunction(torch.autograd.Function):
    """The same site (lora.py:130-135, 3x3 / padding 1 / stride 1) for channels_last activations: csrc/conv_nhwc.hip.

    In [B, H, W, C] memory the up-projection, its gradient pass over G and dUp are the Linear adapter's kernels on the
    [B*H*W, C] rows (``rank_update``, ``linear_bwd_g``); the three 3x3 contractions (T, dX, dDown) are MFMA kernels that
    touch X / dX exactly once, with no partial buffers but the `nsplit` dDown partials.  The frozen convolution and its
    input gradient stay MIOpen (NHWC).  Saves X, T [B*H*W, r] f32 and the packed factor."""

    @staticmethod
    def forward(ctx, x, weight, bias, down, up, sel, scale, dropout_p, sink):
        _C.require()
        B, Ci, H, W = x.shape
        Co, r = weight.shape[0], down.shape[0]
        M = B * H * W
        y = F.conv2d(x, weight, bias, 1, 1)  # frozen dense conv (MIOpen, MFMA)
        if not y.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous(memory_format=torch.channels_last)
        plan = _C.conv3_nhwc_plan(B, Ci, H, W, r)
        down_c = down.float().contiguous()  # f32 masters: no-ops in training
        up2 = up.reshape(Co, r).float().contiguous()
        sel_c = sel.to(torch.float32).contiguous() if sel is not None else None
        seed = off = 0
        if dropout_p > 0.0:
            seed, off = next_dropout_stream(x.device)
        # ONE launch (round 6): packs from the per-step table, T, the fold of its channel shares, the up-projection, dropout
        # and the add; the selector form (set_lora_diag: T' = T S^T between the two products) keeps the launch sequence
        fused = (CONV3_FUSED and sel_c is None and _C.conv3_nhwc_fused_ok(x, Co, r) and down_c is down
                 and up2.data_ptr() == up.data_ptr())
        cnt_bwd = None
        if fused:
            packs = conv_site_packs(sink)
            pf, pd, pu, cnt_fwd, cnt_bwd = packs.get(down_c, up2, x.dtype, plan, _C.linear_bwd_g_blocks(M, Co, r))
            t_part = None
            if plan.ksplit > 1:
                t_part = packs.__dict__.setdefault("_t_part", {}).get((x.device, int(plan.t_part_floats)))
                if t_part is None:   # scratch of ONE launch (complete before the next conv site's forward starts)
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("ConvSitePacks: the T workspace must exist before hipGraph capture")
                    t_part = packs._t_part[(x.device, int(plan.t_part_floats))] = torch.empty(
                        int(plan.t_part_floats), dtype=torch.float32, device=x.device)
            t = _C.conv3_nhwc_fwd_fused_(x, pf, pu, y, r, scale, t_part, cnt_fwd, dropout_p, seed, off)
            _log("fwd", "conv3_nhwc_fused", M, Ci * 9, Co, r)
        else:
            pf, pd = _C.conv3_nhwc_pack(down_c, x.dtype, plan)
            t = _C.conv3_nhwc_down_fwd(x, pf, r)
            if sel_c is not None:  # rare (set_lora_diag): T' = T S^T on the [M, r] rows
                t = (t @ sel_c.t()).contiguous()
            _C.rank_update_(y.permute(0, 2, 3, 1).view(M, Co), t, up2, _C.FACTOR_KR, scale, dropout_p, seed, off)
            _log("fwd", "conv3_nhwc_pack+down+update", M, Ci * 9, Co, r)
        ctx.save_for_backward(x, weight, down, up, t, sel_c, pd)
        ctx.cnt_bwd = cnt_bwd
        ctx.scale, ctx.p, ctx.seed, ctx.off, ctx.sink = float(scale), float(dropout_p), seed, off, sink
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, weight, down, up, t, sel, pd = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co, r = weight.shape[0], down.shape[0]
        M = B * H * W
        need_x, need_w, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        g2 = g.permute(0, 2, 3, 1).view(M, Co)
        cplan = _C.conv3_nhwc_plan(B, Ci, H, W, r)
        lplan = _C.linear_plan(M, Ci, Co, r)
        key = ("conv3n", B, Ci, Co, H, W, r)
        sink = ctx.sink
        if sink is not None:
            if sink.pending is not None:
                sink.flush()
            bufs = sink.conv3_nhwc_workspace(key, lplan, cplan, g.device)
        else:
            bufs = conv3_nhwc_buffers(lplan, cplan, M, r, g.device)
        gt_part, up_part, gt_buf, down_part = bufs
        up2 = up.reshape(Co, r).float().contiguous()
        s, p, seed, off = ctx.scale, ctx.p, ctx.seed, ctx.off
        fused_g = bool(lplan.fused) and _C._rows_ok(g2)
        d_up_direct = None
        if fused_g and ctx.cnt_bwd is not None and _C.linear_bwd_g_folded_ok(g2, up2, r):
            # one pass over G AND the fold of the Gt column-tile partials in the same launch (round 6)
            gt = gt_buf[:M * r].view(M, r)
            _C.linear_bwd_g_folded(g2, t, up2, gt_part, gt, ctx.cnt_bwd, up_part, s, p, seed, off)
        elif fused_g:
            # one pass over G: Gt column-tile partials + dUp row-block partials; the partials of Gt are then folded
            _C.linear_bwd_g(g2, t, up2, gt_part, up_part, s, p, seed, off)
            gt = _C.sum_parts(gt_part, lplan.nct_g, M * r, out=gt_buf[:M * r]).view(M, r)
        else:  # C_out without 16-byte rows: the primitives (two passes over G)
            gt = rowdot_any(g2, up2, _C.FACTOR_KR, s, None, False, p, seed, off)
            d_up_direct = colreduce_any(g2, t, _C.FACTOR_KR, s, p=p, seed=seed, off=off)
        if sel is not None:
            gt = (gt @ sel).contiguous()
        dx = None
        if need_x:  # frozen dense conv's input gradient (MIOpen), then the low-rank term is added in place
            dx = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
            if not dx.is_contiguous(memory_format=torch.channels_last):
                dx = dx.contiguous(memory_format=torch.channels_last)
            _C.conv3_nhwc_bwd_dx_(dx, gt, pd)
        _C.conv3_nhwc_bwd_down(x, gt, down_part)
        d_down = d_up = None
        if sink is not None and fused_g:
            sink.pending = key  # both partial sets are folded by the trainer's batched reduce
        else:
            to_sink = sink is not None
            up_t = sink.up_grad if to_sink else torch.empty(up.shape, dtype=torch.float32, device=g.device)
            down_t = sink.down_grad if to_sink else torch.empty(down.shape, dtype=torch.float32, device=g.device)
            rows = conv3_nhwc_reduce_rows(bufs, lplan, cplan, Ci, Co, r, up_t, down_t, 1.0 if to_sink else 0.0)
            if d_up_direct is not None:
                rows = rows[1:]
                if to_sink:
                    up_t.add_(d_up_direct.view_as(up_t))
                else:
                    up_t = d_up_direct.view(up.shape)
            table, n, total = _C.make_reduce_table(rows, g.device)
            _C.reduce_batched(table, n, total)
            if not to_sink:
                d_up, d_down = up_t.to(up.dtype), down_t.to(down.dtype)
        dw = db = None
        if need_w:
            dw = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        if ctx.has_bias and need_b:
            db = g.sum((0, 2, 3))
        return dx, dw, db, d_down, d_up, None, None, None, None


def _channels_last_only(x: torch.Tensor) -> bool:
    """Memory is [B, H, W, C] and NOT also plain NCHW-contiguous (C == 1 or H == W == 1 are both)."""
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def conv_nhwc_ok(x: torch.Tensor, weight: torch.Tensor, r: int, stride, padding, dilation, groups) -> bool:
    """Channels-last activations at a site the NHWC forms cover: 1x1 (= the Linear adapter on the pixel rows) or
    3x3 / padding 1 with a geometry ``_C.conv3_nhwc_plan`` accepts; stride 1, 16-bit activations."""
    if not _channels_last_only(x) or groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1):
        return False
    kh, kw = weight.shape[2], weight.shape[3]
    if kh != kw or kh not in (1, 3) or tuple(padding) != ((kh - 1) // 2,) * 2:
        return False
    if x.dtype not in (torch.bfloat16, torch.float16) or weight.dtype != x.dtype:
        return False
    if kh == 1:
        return True
    B, Ci, H, W = x.shape
    return bool(_C.conv3_nhwc_plan(B, Ci, H, W, r).native)


def conv_native_ok(x: torch.Tensor, weight: torch.Tensor, r: int, stride, padding, dilation, groups) -> bool:
    if x.dim() != 4 or groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1):
        return False
    kh, kw = weight.shape[2], weight.shape[3]
    if kh != kw or kh not in (1, 3) or tuple(padding) != ((kh - 1) // 2,) * 2:
        return False
    B, Ci, H, W = x.shape
    return bool(_C.conv_plan(B, Ci, weight.shape[0], H, W, kh, r).native)


def lora_conv(x, weight, bias, down_w, up_w, sel, stride, padding, dilation, groups, scale, dropout_p, sink=None):
    """LoraInjectedConv2d forward on device tensors: the native kernels when the geometry qualifies, else the frozen
    conv + ``lora_conv_branch`` (library conv for the k x k down-projection, HIP kernel for the rest)."""
    r = down_w.shape[0]
    if down_w.dtype == up_w.dtype and conv_nhwc_ok(x, weight, r, stride, padding, dilation, groups):
        Co, Ci = weight.shape[0], weight.shape[1]
        if weight.shape[2] == 1:
            # a 1x1 convolution of channels-last pixels IS the Linear site on the [B*H*W, C] rows (zero-copy views)
            y = LoraLinearFunction.apply(x.permute(0, 2, 3, 1), weight.reshape(Co, Ci), bias, down_w.reshape(r, Ci),
                                         up_w.reshape(Co, r), sel, float(scale), float(dropout_p), sink)
            return y.permute(0, 3, 1, 2)
        return LoraConv3NhwcFunction.apply(x, weight, bias, down_w, up_w, sel, float(scale), float(dropout_p), sink)
    if down_w.dtype == up_w.dtype and conv_native_ok(x, weight, r, stride, padding, dilation, groups):
        return LoraConvFunction.apply(x, weight, bias, down_w, up_w, sel, int(weight.shape[2]), float(scale),
                                      float(dropout_p), sink)
    y0 = F.conv2d(x, weight, bias, stride, padding, dilation, groups)
    return lora_conv_branch(x, y0, down_w, up_w, sel, stride, padding, dilation, groups, scale, dropout_p)


def lora_conv_branch(x, y0, down_w, up_w, sel, stride, padding, dilation, groups, scale, dropout_p):
    """Low-rank branch of LoraInjectedConv2d added onto ``y0`` (the frozen conv's output) for geometries
    outside the native set (strided / dilated / grouped / odd spatial sizes).

    The k x k down-projection to r channels runs as a library conv; the 1x1 up-projection,
    dropout, scale and the add are one fused HIP kernel per sample."""
    if not y0.is_contiguous():  # channels_last host model: this (rare, non-native geometry) branch works in NCHW
        y0 = y0.contiguous()
    t = F.conv2d(x, down_w if down_w.dtype == x.dtype else down_w.to(x.dtype), None, stride, padding, dilation,
                 groups)
    if sel is not None:
        r = t.shape[1]
        t = F.conv2d(t, sel.reshape(r, r, 1, 1).to(t.dtype))
    return LoraConvUpFunction.apply(y0, t, up_w, float(scale), float(dropout_p))


def merge_sites(sites, alpha: float = 1.0, rounding: int = _C.ROUND_REFERENCE) -> None:
    """One launch per (weight dtype, factor dtype) group: w_out = w_in + alpha * up @ down."""
    groups = {}
    for s in sites:
        groups.setdefault((s[0].dtype, s[2].dtype, s[0].device), []).append(s)
    for grp in groups.values():
        _C.MergePlan(grp).launch(alpha, rounding)
