"""Autograd glue between the adapter modules and the HIP primitives.

Device tensors only: everything here ends in a call through the C-ABI
(``_C.require()`` raises if the library is missing).  The frozen dense
contractions (``X @ W^T``, ``G @ W``) are plain library GEMMs on MFMA
(hipBLASLt via ``F.linear``/``matmul``); the low-rank branch and its gradients
are the hand-written kernels of ``csrc/linear.hip``.
"""
from __future__ import annotations

import threading
from typing import Optional

import torch
import torch.nn.functional as F

from . import _C

_dropout_lock = threading.Lock()
_dropout_calls = 0


def next_dropout_stream() -> tuple:
    """(seed, offset) for one dropout mask: torch's seed + a per-process call counter.

    Deterministic under ``torch.manual_seed``; the mask is regenerated in backward
    from the same pair, so no [M, N] mask tensor ever exists in HBM.
    """
    global _dropout_calls
    with _dropout_lock:
        _dropout_calls += 1
        off = _dropout_calls
    return int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, off


class LoraLinearFunction(torch.autograd.Function):
    """y = x W^T + b + scale * dropout((x A^T) S^T B^T)   (lora.py:53-58) and its gradient.

    Saves X and the [M, r] projection T only — never an [M, N] tensor.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, down, up, sel, scale, dropout_p, grad_slots):
        _C.require()
        K = weight.shape[1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) != K):
            x2 = x2.contiguous()
        y = F.linear(x2, weight, bias)  # frozen dense GEMM (MFMA, hipBLASLt)
        t = _C.rowdot(x2, down, _C.FACTOR_RK, 1.0, sel, False)
        seed = off = 0
        if dropout_p > 0.0:
            seed, off = next_dropout_stream()
        _C.rank_update_(y, t, up, _C.FACTOR_KR, scale, dropout_p, seed, off)
        ctx.save_for_backward(x2, weight, down, up, t, sel)
        ctx.scale, ctx.p, ctx.seed, ctx.off = float(scale), float(dropout_p), seed, off
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        ctx.grad_slots = grad_slots
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x2, weight, down, up, t, sel = ctx.saved_tensors
        N, K = weight.shape
        g2 = g.reshape(-1, N)
        if g2.stride(-1) != 1 or (g2.shape[0] > 1 and g2.stride(0) != N):
            g2 = g2.contiguous()
        need_x, need_w, need_b, need_down, need_up = ctx.needs_input_grad[:5]
        s, p, seed, off = ctx.scale, ctx.p, ctx.seed, ctx.off

        gt = None
        if need_x or need_down:
            # dT = scale * (G .* mask) @ B (@ S)
            gt = _C.rowdot(g2, up, _C.FACTOR_KR, s, sel, True, p, seed, off)
        d_up = d_down = None
        slots = ctx.grad_slots
        if need_up:
            if slots is not None:  # accumulate straight into the trainer's flat grad buffer
                _C.colreduce(g2, t, _C.FACTOR_KR, s, out=slots[1], beta=1.0, dropout_p=p, seed=seed, offset=off)
            else:
                d_up = _C.colreduce(g2, t, _C.FACTOR_KR, s, dropout_p=p, seed=seed, offset=off).to(up.dtype)
        if need_down:
            if slots is not None:
                _C.colreduce(x2, gt, _C.FACTOR_RK, 1.0, out=slots[0], beta=1.0)
            else:
                d_down = _C.colreduce(x2, gt, _C.FACTOR_RK, 1.0).to(down.dtype)
        dx = None
        if need_x:
            dx2 = g2 @ weight  # frozen dense GEMM
            _C.rank_update_(dx2, gt, down, _C.FACTOR_RK, 1.0)
            dx = dx2.view(ctx.x_shape)
        dw = g2.t() @ x2 if need_w else None
        db = g2.sum(0) if (ctx.has_bias and need_b) else None
        return dx, dw, db, d_down, d_up, None, None, None, None


def lora_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], down: torch.Tensor,
                up: torch.Tensor, sel: Optional[torch.Tensor], scale: float, dropout_p: float,
                grad_slots=None) -> torch.Tensor:
    return LoraLinearFunction.apply(x, weight, bias, down, up, sel, float(scale), float(dropout_p), grad_slots)


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
                seed, off = next_dropout_stream()
            streams.append((seed, off))
            _C.rank_update_(y0[b].view(Co, H * W), up2, t[b].view(r, H * W), _C.FACTOR_RK, scale, dropout_p,
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
                _C.colreduce(gb, up2, _C.FACTOR_RK, ctx.scale, out=dt[b], beta=0.0, dropout_p=ctx.p, seed=seed,
                             offset=off)
            if need_up:  # dUp += scale * (G_b .* mask) @ T_b^T
                part = _C.rowdot(gb, t[b].view(r, H * W), _C.FACTOR_RK, ctx.scale, None, False, ctx.p, seed, off)
                dup = part if dup is None else dup + part
        dt_out = dt.view(B, r, H, W).to(t.dtype) if need_t else None
        dup_out = dup.view(ctx.up_shape).to(ctx.up_dtype) if need_up else None
        return g, dt_out, dup_out, None, None


def lora_conv_branch(x, y0, down_w, up_w, sel, stride, padding, dilation, groups, scale, dropout_p):
    """Low-rank branch of LoraInjectedConv2d added onto ``y0`` (the frozen conv's output).

    The k x k down-projection to r channels runs as a library conv for now (a HIP
    implicit-GEMM kernel replaces it in a later round); the 1x1 up-projection,
    dropout, scale and the add are one fused HIP kernel per sample.
    """
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
