"""SD1.5-shaped UNet used where ``diffusers`` is not installed (it is not in this image).

The reference takes its host model from ``diffusers.UNet2DConditionModel`` (ref:
training_scripts/train_lora_dreambooth.py:590-594); that package is un-vendored, so the trainer,
tests and bench build this stand-in instead: same architecture (block_out 320/640/1280/1280, 2 layers
per block, 8 heads, cross-attention dim 768 -> 859,520,964 parameters), same *class names* and the
same *registration order* as diffusers 0.11 — which is what the LoRA finder keys on and what fixes the
``unet:{i}`` index layout of saved files (attn1, ff, attn2 inside a block; down -> up -> mid across
the model; checked against the reference's shipped fixture in tests/test_standin.py).
Weights are random-init: there is no network to fetch checkpoints.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import attention
from .attention import sdpa
from .fused import add_layer_norm, geglu, group_norm_act, layer_norm
from torch.utils.checkpoint import checkpoint


class UNetOutput:
    """``.sample`` like diffusers' UNet2DConditionOutput; also indexable (``out[0]``)."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class Timesteps(nn.Module):
    def __init__(self, dim: int, flip_sin_to_cos: bool = True, shift: float = 0.0):
        super().__init__()
        self.dim, self.flip, self.shift = dim, flip_sin_to_cos, shift

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        half = self.dim // 2
        freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - self.shift))
        ang = t[:, None].float() * freq[None, :]
        sin, cos = torch.sin(ang), torch.cos(ang)
        return torch.cat([cos, sin] if self.flip else [sin, cos], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


def _linear_group(mods, x, out_heads=None):
    """Several projections on one tensor in one launch: LoRA adapters (``lora.lora_linear_group``) or their frozen twins
    (``standin/frozen.py``, the `frozen_only` measurement leg); None: call the modules one by one."""
    from ..lora import lora_linear_group
    from .frozen import FrozenSite, frozen_linear_group

    if isinstance(mods[0], FrozenSite):
        return frozen_linear_group(mods, x, out_heads)
    return lora_linear_group(mods, x, out_heads) if out_heads is not None else lora_linear_group(mods, x)


def project_qkv(attn, x, context=None):
    """to_q / to_k / to_v of an attention block.  When the projections are LoRA adapters that read the same tensor
    (self-attention: all three; cross-attention: to_k and to_v on the text states) they go out as ONE launch of the
    weight-stationary kernel (``lora.lora_linear_group``); otherwise module by module, as diffusers does."""
    ctx = x if context is None else context
    if x.is_cuda and os.environ.get("LORA_AMD_GROUP_QKV", "1") != "0":
        lora_linear_group = _linear_group

        if context is None:
            out = lora_linear_group([attn.to_q, attn.to_k, attn.to_v], x)
            if out is not None:
                return out
        else:
            kv = lora_linear_group([attn.to_k, attn.to_v], ctx)
            if kv is not None:
                return [attn.to_q(x)] + kv
    return [attn.to_q(x), attn.to_k(ctx), attn.to_v(ctx)]


class CrossAttention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0):
        super().__init__()
        inner = heads * dim_head
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(ctx, inner, bias=False)
        self.to_v = nn.Linear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, x, context=None):
        ctx = x if context is None else context
        B, T, _ = x.shape
        h = self.heads
        d = self.dim_head
        grad = torch.is_grad_enabled() and (x.requires_grad or ctx.requires_grad or _trains(self))
        # opt-in (LORA_AMD_HEAD_PAD=1; round-1 design, LDS-ring kernel): the projections write / read the padded head
        # layout of the chosen attention kernel.  Default: q/k/v as one weight-stationary launch (project_qkv)
        pad = attention.padded_choice(B, h, T, ctx.shape[1], d, x.dtype, grad) \
            if (os.environ.get("LORA_AMD_HEAD_PAD", "0") == "1" or attention.FORCE_PAD is not None) else None
        if pad is not None:
            # the chosen attention kernel wants head size D > d: the projections write / read that layout themselves
            # (fused GEMM epilogue and operand remap, ops.lora_linear) instead of pad + slice copies around the core
            backend, D = pad
            lay = (h, d, D)
            q = k = v = None
            if x.is_cuda and os.environ.get("LORA_AMD_GROUP_QKV", "1") != "0":
                lora_linear_group = _linear_group
                if context is None:
                    out = lora_linear_group([self.to_q, self.to_k, self.to_v], x, out_heads=lay)
                    if out is not None:
                        q, k, v = out
                else:
                    out = lora_linear_group([self.to_k, self.to_v], ctx, out_heads=lay)
                    if out is not None:
                        k, v = out
            q = _project(self.to_q, x, None, lay) if q is None else q
            k = _project(self.to_k, ctx, None, lay) if k is None else k
            v = _project(self.to_v, ctx, None, lay) if v is None else v
            q = q.view(B, T, h, D).transpose(1, 2)
            k = k.view(B, ctx.shape[1], h, D).transpose(1, 2)
            v = v.view(B, ctx.shape[1], h, D).transpose(1, 2)
            o = attention.sdpa_padded(q, k, v, d, backend).transpose(1, 2).reshape(B, T, h * D)
            return self.to_out[1](_project(self.to_out[0], o, lay, None))
        q, k, v = project_qkv(self, x, context)
        q = q.view(B, T, h, -1).transpose(1, 2)
        k = k.view(B, ctx.shape[1], h, -1).transpose(1, 2)
        v = v.view(B, ctx.shape[1], h, -1).transpose(1, 2)
        o = sdpa(q, k, v)  # dense contraction: library MFMA flash kernels, fastest variant per shape (attention.py)
        o = o.transpose(1, 2).reshape(B, T, -1)
        return self.to_out[1](self.to_out[0](o))


def _trains(module: nn.Module) -> bool:
    return any(p.requires_grad for p in module.parameters())


def _project(lin: nn.Module, x: torch.Tensor, in_heads, out_heads) -> torch.Tensor:
    """A q / k / v / out projection on head-padded activations: adapters handle the layout themselves
    (``forward_heads``), a plain Linear gets dense copies around it."""
    if hasattr(lin, "forward_heads"):
        return lin.forward_heads(x, in_heads, out_heads)
    from ..ops import pack_heads, unpack_heads

    y = lin(unpack_heads(x, in_heads) if in_heads else x)
    return pack_heads(y, out_heads) if out_heads else y


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return geglu(self.proj(x))  # h * gelu(gate): one launch each way behind the adapted proj (fused.py)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4, dropout: float = 0.0):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        # registration order attn1, ff, attn2, norm1..3 == the fixture's 9-per-block index pattern
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, x, context):
        # each residual add runs inside the LayerNorm pass that follows it (fused.add_layer_norm)
        x, n = add_layer_norm(self.attn1(layer_norm(x, self.norm1)), x, self.norm2)
        x, n = add_layer_norm(self.attn2(n, context), x, self.norm3)
        return self.ff(n) + x


class Transformer2DModel(nn.Module):
    def __init__(self, channels: int, heads: int, cross_attention_dim: int, groups: int = 32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, channels // heads, cross_attention_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, context):
        B, C, H, W = x.shape
        n = group_norm_act(x, self.norm, act=False)
        nhwc = n.is_contiguous(memory_format=torch.channels_last) and not n.is_contiguous()
        # A 1x1 convolution IS the token-major linear map: one library GEMM with the bias in its epilogue instead of a
        # MIOpen call (on NCHW activations wrapped in NCHW<->NHWC transposes) plus a strided bias add.  On channels_last
        # activations the token view is free in both directions (proj_in / proj_out are not adapter sites of the
        # reference's target classes; if they have been replaced, keep the module call).
        as_linear = (type(self.proj_in) is nn.Conv2d and type(self.proj_out) is nn.Conv2d
                     and self.proj_in.kernel_size == (1, 1) and self.proj_out.kernel_size == (1, 1))
        if as_linear:
            h = n.permute(0, 2, 3, 1).reshape(B, H * W, C)  # a view when the activations are NHWC, else one copy
            h = F.linear(h, self.proj_in.weight.view(C, C), self.proj_in.bias)
        else:
            h = self.proj_in(n).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if as_linear:
            h = F.linear(h, self.proj_out.weight.view(C, C), self.proj_out.bias)
            h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)  # channels_last strides
            return (h if nhwc else h.contiguous()) + x
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        if not nhwc:
            h = h.contiguous()
        return self.proj_out(h) + x


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int = 1280, groups: int = 32,
                 eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        n1 = group_norm_act(x, self.norm1)  # GroupNorm + SiLU as HIP passes (fused.py)
        t = self.time_emb_proj(self.nonlinearity(temb))  # [B, C_out]
        if type(self.conv1) is nn.Conv2d and self.conv1.bias is not None:
            # the convolution's bias and the time embedding are both per-(sample, channel) terms in front of norm2:
            # one small [B, C] add instead of two passes over the activation (none at all on channels_last)
            c1 = self.conv1
            h = F.conv2d(n1, c1.weight, None, c1.stride, c1.padding, c1.dilation, c1.groups)
            t = t + c1.bias
        else:
            h = self.conv1(n1)
        h = self.conv2(self.dropout(group_norm_act(h, self.norm2, addend=t)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Downsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _DownBase(nn.Module):
    has_attn = False

    def __init__(self, cin, cout, temb, layers, heads, ctx_dim, add_downsample):
        super().__init__()
        if self.has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim) for _ in range(layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, h, temb, context):
        outs = []
        for i, res in enumerate(self.resnets):
            h = res(h, temb)
            if self.has_attn:
                h = self.attentions[i](h, context)
            outs.append(h)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            outs.append(h)
        return h, outs


class CrossAttnDownBlock2D(_DownBase):
    has_attn = True


class DownBlock2D(_DownBase):
    has_attn = False


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, channels, temb, heads, ctx_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(channels, heads, ctx_dim)])
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb), ResnetBlock2D(channels, channels, temb)])

    def forward(self, h, temb, context):
        h = self.resnets[0](h, temb)
        for attn, res in zip(self.attentions, self.resnets[1:]):
            h = res(attn(h, context), temb)
        return h


class _UpBase(nn.Module):
    has_attn = False

    def __init__(self, cin, prev, cout, temb, layers, heads, ctx_dim, add_upsample):
        super().__init__()
        if self.has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, ctx_dim) for _ in range(layers)])
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            res.append(ResnetBlock2D((prev if i == 0 else cout) + skip, cout, temb))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, h, skips: List[torch.Tensor], temb, context):
        for i, res in enumerate(self.resnets):
            h = res(torch.cat([h, skips[-1 - i]], dim=1), temb)  # no mutation: checkpoint re-runs this
            if self.has_attn:
                h = self.attentions[i](h, context)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h)
        return h


class UpBlock2D(_UpBase):
    has_attn = False


class CrossAttnUpBlock2D(_UpBase):
    has_attn = True


class UNet2DConditionModel(nn.Module):
    """SD1.5 geometry by default; ``block_out_channels``/``layers_per_block`` shrink it for tests."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 attention_heads=8, cross_attention_dim=768, norm_num_groups=32):
        super().__init__()
        ch = list(block_out_channels)
        temb = ch[0] * 4
        self.config = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(ch),
                           layers_per_block=layers_per_block, attention_head_dim=attention_heads,
                           cross_attention_dim=cross_attention_dim, sample_size=64)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_proj = Timesteps(ch[0])
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        # down_blocks and up_blocks are registered before mid_block, as in diffusers: index order down -> up -> mid
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        out = ch[0]
        for i, c in enumerate(ch):
            cin, out = out, c
            last = i == len(ch) - 1
            cls = DownBlock2D if last else CrossAttnDownBlock2D
            self.down_blocks.append(cls(cin, out, temb, layers_per_block, attention_heads, cross_attention_dim, not last))
        self.mid_block = UNetMidBlock2DCrossAttn(ch[-1], temb, attention_heads, cross_attention_dim)
        rev = ch[::-1]
        out = rev[0]
        for i, c in enumerate(rev):
            prev, out = out, c
            cin = rev[min(i + 1, len(ch) - 1)]
            cls = UpBlock2D if i == 0 else CrossAttnUpBlock2D
            self.up_blocks.append(cls(cin, prev, out, temb, layers_per_block + 1, attention_heads, cross_attention_dim,
                                      i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self._grad_ckpt = False

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def enable_gradient_checkpointing(self):  # ref: train_lora_dreambooth.py:627-628
        self._grad_ckpt = True

    @property
    def gradient_checkpointing(self) -> bool:
        """The diffusers / transformers attribute name (what trainer._dropout_pool looks for: a step that recomputes
        activations must regenerate the SAME dropout masks, so it keeps the per-site RNG draws)."""
        return self._grad_ckpt

    def _run(self, blk, *a):
        if self._grad_ckpt and self.training and torch.is_grad_enabled():
            return checkpoint(blk, *a, use_reentrant=False)
        return blk(*a)

    def forward(self, sample, timestep, encoder_hidden_states) -> UNetOutput:
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        if timestep.dim() == 0:
            timestep = timestep[None]
        timestep = timestep.expand(sample.shape[0])
        temb = self.time_embedding(self.time_proj(timestep).to(self.time_embedding.linear_1.weight.dtype))
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = self._run(blk, h, temb, encoder_hidden_states)
            skips.extend(outs)
        h = self._run(self.mid_block, h, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            take, skips = skips[-n:], skips[:-n]
            h = self._run(blk, h, take, temb, encoder_hidden_states)
        return UNetOutput(self.conv_out(group_norm_act(h, self.conv_norm_out)))


def sd15_unet(**kw) -> UNet2DConditionModel:
    return UNet2DConditionModel(**kw)


def tiny_unet(cross_attention_dim: int = 32) -> UNet2DConditionModel:
    """4-level miniature (same topology, registration order and class names) for CPU tests."""
    return UNet2DConditionModel(block_out_channels=(32, 64, 64, 64), layers_per_block=1, attention_heads=2,
                                cross_attention_dim=cross_attention_dim, norm_num_groups=8)
