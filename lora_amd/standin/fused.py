"""Fused normalisation / activation passes of the frozen stand-in UNet (``csrc/hostops.hip``).

The reference's training step runs diffusers' UNet through ATen (`train_lora_dreambooth.py:838-892`); between the
adapted sites that is GroupNorm -> SiLU (4 + 5 launches fwd/bwd, two extra saved tensors) and the GEGLU gate behind the
adapted ``proj`` (chunk, gelu, mul; gelu_backward, 2 mul, cat).  On a device with the HIP library these run as
2 + 2 and 1 + 1 launches that save only the block input (and [B, G] statistics).

The affine parameters of the host model are frozen in LoRA training; when they do require grad, when the tensor is
not NCHW-contiguous, or on CPU, the ATen sequence runs instead (same mathematics).  ``LORA_AMD_HOSTOPS=0`` forces
that path everywhere (A/B measurements).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _C

_ENABLED = os.environ.get("LORA_AMD_HOSTOPS", "1") != "0"
_DTYPES = (torch.bfloat16, torch.float16, torch.float32)


class _GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups: int, eps: float, act: bool):
        y, stats = _C.groupnorm_fwd(x, weight, bias, groups, eps, act)
        ctx.save_for_backward(x, weight, bias, stats)
        ctx.groups, ctx.act = groups, act
        return y

    @staticmethod
    def backward(ctx, gout):
        x, weight, bias, stats = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _C.groupnorm_bwd(x, gout.contiguous(), weight, bias, stats, ctx.groups, ctx.act)
        return dx, None, None, None, None, None


class _GroupNormActNHWC(torch.autograd.Function):
    """The same for channels_last activations (memory [B][HW][C]); saves x and the per-channel affine [B, 4, C]."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups: int, eps: float, act: bool, addend):
        add32 = None if addend is None else addend.detach().to(torch.float32).contiguous()
        y, aff = _C.groupnorm_nhwc_fwd(x, weight, bias, groups, eps, act, add32)
        ctx.save_for_backward(x, weight, aff)
        ctx.groups, ctx.act = groups, act
        ctx.add_dtype = None if addend is None else addend.dtype
        return y

    @staticmethod
    def backward(ctx, gout):
        x, weight, aff = ctx.saved_tensors
        dx = dadd = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[6]:
            gout = gout.contiguous(memory_format=torch.channels_last)
            dx = _C.groupnorm_nhwc_bwd(x, gout, weight, aff, ctx.groups, ctx.act)
            if ctx.needs_input_grad[6]:  # only when something upstream of the addend trains (extended injection)
                dadd = dx.sum(dim=(2, 3), dtype=torch.float32).to(ctx.add_dtype)
            if not ctx.needs_input_grad[0]:
                dx = None
        return dx, None, None, None, None, None, dadd


def _frozen_affine(x: torch.Tensor, norm: nn.GroupNorm) -> bool:
    w, b = norm.weight, norm.bias
    return not (w is None or b is None or w.requires_grad or b.requires_grad or w.dtype != x.dtype
                or b.dtype != x.dtype)


def _gn_native(x: torch.Tensor, norm: nn.GroupNorm) -> bool:
    if not (_ENABLED and x.is_cuda and x.dim() >= 3 and x.dtype in _DTYPES and x.is_contiguous()):
        return False
    if not _frozen_affine(x, norm):
        return False
    B, C = x.shape[0], x.shape[1]
    return _C.groupnorm_workspace(B, C, x.numel() // (B * C), norm.num_groups) > 0


def _gn_native_nhwc(x: torch.Tensor, norm: nn.GroupNorm) -> bool:
    if not (_ENABLED and x.is_cuda and x.dim() == 4 and x.dtype in _DTYPES and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 32 == 0):
        return False
    if not _frozen_affine(x, norm):
        return False
    B, C = x.shape[0], x.shape[1]
    return _C.groupnorm_nhwc_workspace(B, C, x.numel() // (B * C), norm.num_groups) > 0


def group_norm_act(x: torch.Tensor, norm: nn.GroupNorm, act: bool = True,
                   addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``silu(norm(x))`` (``act``) or ``norm(x)`` — ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm.
    ``addend`` [B, C] is added to x (broadcast over the pixels) before the normalisation; the channels_last kernels
    absorb it for free, every other path adds it first."""
    if _gn_native_nhwc(x, norm):
        return _GroupNormActNHWC.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act, addend)
    if addend is not None:
        x = x + addend[:, :, None, None].to(x.dtype)
    if _gn_native(x, norm):
        return _GroupNormAct.apply(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act)
    y = norm(x)
    return F.silu(y) if act else y


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps: float):
        y, stats = _C.layernorm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, stats)
        return y

    @staticmethod
    def backward(ctx, gout):
        x, weight, stats = ctx.saved_tensors
        dx = _C.layernorm_bwd(x, gout.contiguous(), weight, stats) if ctx.needs_input_grad[0] else None
        return dx, None, None, None


def layer_norm(x: torch.Tensor, norm: nn.LayerNorm) -> torch.Tensor:
    """``norm(x)`` over the last dimension (BasicTransformerBlock.norm1/2/3): one launch each way."""
    w, b = norm.weight, norm.bias
    if (_ENABLED and x.is_cuda and x.dtype in _DTYPES and x.is_contiguous() and x.numel() > 0
            and len(norm.normalized_shape) == 1 and w is not None and b is not None
            and not (w.requires_grad or b.requires_grad) and w.dtype == x.dtype and b.dtype == x.dtype
            and x.data_ptr() % 32 == 0 and _C.layernorm_supported(x.shape[-1])):
        return _LayerNorm.apply(x, w, b, norm.eps)
    return norm(x)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight, bias, eps: float):
        s, y, stats = _C.add_layernorm_fwd(a, b, weight, bias, eps)
        ctx.save_for_backward(s, weight, stats)
        return s, y

    @staticmethod
    def backward(ctx, gs, gy):
        s, weight, stats = ctx.saved_tensors
        if gy is None:  # only the sum was used downstream
            return gs, gs, None, None, None
        dx = _C.add_layernorm_bwd(s, gy.contiguous(), None if gs is None else gs.contiguous(), weight, stats)
        return dx, dx, None, None, None


def _ln_native(x: torch.Tensor, norm: nn.LayerNorm) -> bool:
    w, b = norm.weight, norm.bias
    return (_ENABLED and x.is_cuda and x.dtype in _DTYPES and x.is_contiguous() and x.numel() > 0
            and len(norm.normalized_shape) == 1 and w is not None and b is not None
            and not (w.requires_grad or b.requires_grad) and w.dtype == x.dtype and b.dtype == x.dtype
            and x.data_ptr() % 32 == 0 and _C.layernorm_supported(x.shape[-1]))


def add_layer_norm(a: torch.Tensor, b: torch.Tensor, norm: nn.LayerNorm):
    """``s = a + b; return s, norm(s)`` — the residual add of a transformer block and the norm that follows it, in one
    pass each way (the sum is rounded to the activation dtype exactly as the separate add would)."""
    if a.shape == b.shape and a.dtype == b.dtype and _ln_native(a, norm) and b.is_contiguous() and b.is_cuda \
            and b.data_ptr() % 32 == 0:
        return _AddLayerNorm.apply(a, b, norm.weight, norm.bias, norm.eps)
    s = a + b
    return s, layer_norm(s, norm)


class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        return _C.geglu_fwd(y)

    @staticmethod
    def backward(ctx, gout):
        (y,) = ctx.saved_tensors
        return _C.geglu_bwd(y, gout.contiguous())


def geglu(y: torch.Tensor) -> torch.Tensor:
    """``h * gelu(gate)`` for ``y = [h | gate]`` along the last dimension (GEGLU.forward after ``proj``)."""
    if (_ENABLED and y.is_cuda and y.dtype in _DTYPES and y.is_contiguous() and y.shape[-1] % 16 == 0
            and y.numel() > 0 and y.data_ptr() % 32 == 0):
        return _Geglu.apply(y)
    h, gate = y.chunk(2, dim=-1)
    return h * F.gelu(gate)
