"""Glue for a real ``diffusers`` UNet (the host model of the reference: train_lora_dreambooth.py:590-594,
cli_lora_pti.py:116-120; un-vendored, absent from this image — exercised with a fake module in tests/).

``diffusers`` attention blocks (class ``Attention`` / ``CrossAttention``) call an *attention processor*; installing
:class:`LoraAmdAttnProcessor` makes their to_q / to_k / to_v projections — three ``LoraInjectedLinear`` reading one
tensor after ``inject_trainable_lora`` — go out as ONE weight-stationary launch (``lora.lora_linear_group``) exactly as
the stand-in ``CrossAttention`` does, and runs the dense softmax(QK^T)V on the library's fused kernel.  Anything the
processor does not understand (attention masks with unusual shapes, added-KV / norm_cross variants) is handed back to
the block's previous processor.

:func:`install_host_options` binds the remaining frozen-host passes of the stand-in (``standin/fused.py``: NHWC / NCHW
GroupNorm(+SiLU) with the time-embedding addend, residual-add + LayerNorm, the GEGLU gate; ``standin/attention.py``: the
per-shape attention-kernel choice with head-padded projections) into the corresponding ``diffusers`` blocks —
``ResnetBlock2D``, ``BasicTransformerBlock``, ``GEGLU``, and the attention processor above — by replacing the
instance's ``forward`` for the plain SD1.x configuration and keeping the original ``forward`` for everything else
(ada-norm, gated / positional variants, up/down-sampling ResNets, scale-shift time embeddings, masks, chunked
feed-forward ...).  Matched by class NAME and duck-typed attributes: ``diffusers`` is not importable in this image, the
tests drive these bindings with fakes that copy the attribute surface of diffusers 0.11-0.30."""
from __future__ import annotations

import os
import types

import torch
import torch.nn.functional as F

from .lora import lora_linear_group


class LoraAmdAttnProcessor:
    """Drop-in for ``diffusers.models.attention_processor.AttnProcessor2_0`` on SD1.x-style attention blocks."""

    def __init__(self, fallback=None, tuned: bool = False):
        self.fallback = fallback
        self.tuned = tuned  # install_host_options: per-shape attention-kernel choice + head-padded projections

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        plain = (hidden_states.dim() == 3 and attention_mask is None and getattr(attn, "norm_cross", None) is None
                 and getattr(attn, "group_norm", None) is None and getattr(attn, "spatial_norm", None) is None
                 and not getattr(attn, "residual_connection", False)
                 and getattr(attn, "added_kv_proj_dim", None) is None
                 # qk-norm, fp32-upcast attention / softmax and fused q|k|v projections change what the block computes:
                 # never approximate them, hand the call back
                 and getattr(attn, "norm_q", None) is None and getattr(attn, "norm_k", None) is None
                 and not getattr(attn, "upcast_attention", False) and not getattr(attn, "upcast_softmax", False)
                 and not getattr(attn, "fused_projections", False))
        if not plain:
            if self.fallback is None:
                raise NotImplementedError("LoraAmdAttnProcessor: attention variant outside the SD1.x pattern")
            return self.fallback(attn, hidden_states, encoder_hidden_states, attention_mask, temb, *args, **kwargs)
        x = hidden_states
        ctx = x if encoder_hidden_states is None else encoder_hidden_states
        o = self._padded(attn, x, ctx)
        if o is not None:
            return o
        if encoder_hidden_states is None:
            qkv = lora_linear_group([attn.to_q, attn.to_k, attn.to_v], x)
            q, k, v = qkv if qkv is not None else (attn.to_q(x), attn.to_k(x), attn.to_v(x))
        else:
            kv = lora_linear_group([attn.to_k, attn.to_v], ctx)
            k, v = kv if kv is not None else (attn.to_k(ctx), attn.to_v(ctx))
            q = attn.to_q(x)
        B, T, _ = x.shape
        h = attn.heads
        q, k, v = (t.view(B, t.shape[1], h, -1).transpose(1, 2) for t in (q, k, v))
        scale = getattr(attn, "scale", None)
        if self.tuned and (scale is None or abs(scale - q.shape[-1] ** -0.5) < 1e-6 * q.shape[-1] ** -0.5):
            from .standin.attention import sdpa

            o = sdpa(q, k, v)  # fastest library kernel per shape, timed once (flash / efficient, head dim padded or not)
        else:
            o = F.scaled_dot_product_attention(q, k, v, scale=scale)
        o = o.transpose(1, 2).reshape(B, T, -1).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        rescale = getattr(attn, "rescale_output_factor", 1.0)
        return o / rescale if rescale != 1.0 else o

    def _padded(self, attn, x, ctx):
        """The head-padded route of the stand-in's CrossAttention: when the attention kernel chosen for this shape runs
        on head size D > d, the four projections write / read that layout themselves (``forward_heads``), so no pad /
        slice copies surround the core.  None when it does not apply."""
        if not (self.tuned and x.is_cuda and os.environ.get("LORA_AMD_HEAD_PAD", "0") == "1"):
            return None
        from .standin import attention

        projs = (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0])
        if not all(hasattr(p, "forward_heads") for p in projs):
            return None
        h = attn.heads
        inner = attn.to_q.linear.out_features
        d = inner // h
        scale = getattr(attn, "scale", None)
        if scale is not None and abs(scale - d ** -0.5) > 1e-6 * d ** -0.5:
            return None
        B, T, _ = x.shape
        grad = torch.is_grad_enabled()
        pad = attention.padded_choice(B, h, T, ctx.shape[1], d, x.dtype, grad)
        if pad is None:
            return None
        backend, D = pad
        lay = (h, d, D)
        grouped = lora_linear_group([attn.to_q, attn.to_k, attn.to_v], x, out_heads=lay) if ctx is x else None
        kv = lora_linear_group([attn.to_k, attn.to_v], ctx, out_heads=lay) if ctx is not x else None
        q, k, v = grouped if grouped is not None else (None, None, None)
        if kv is not None:
            k, v = kv
        q = attn.to_q.forward_heads(x, None, lay) if q is None else q
        k = attn.to_k.forward_heads(ctx, None, lay) if k is None else k
        v = attn.to_v.forward_heads(ctx, None, lay) if v is None else v
        q = q.view(B, T, h, D).transpose(1, 2)
        k = k.view(B, ctx.shape[1], h, D).transpose(1, 2)
        v = v.view(B, ctx.shape[1], h, D).transpose(1, 2)
        o = attention.sdpa_padded(q, k, v, d, backend).transpose(1, 2).reshape(B, T, h * D)
        o = attn.to_out[1](attn.to_out[0].forward_heads(o, lay, None))
        rescale = getattr(attn, "rescale_output_factor", 1.0)
        return o / rescale if rescale != 1.0 else o


def install_attention_processor(unet) -> int:
    """Install :class:`LoraAmdAttnProcessor` on every attention block of a diffusers UNet (keeps each block's previous
    processor as the fallback).  Returns the number of blocks switched; 0 for models without the processor API."""
    n = 0
    for m in unet.modules():
        if hasattr(m, "set_processor") and hasattr(m, "to_q") and hasattr(m, "heads"):
            m.set_processor(LoraAmdAttnProcessor(getattr(m, "processor", None)))
            n += 1
    return n


# ----------------------------------------------------------------------------- frozen-host passes for diffusers blocks
def _geglu_forward(self, hidden_states, *args, **kwargs):
    """diffusers.models.activations.GEGLU.forward: ``h, gate = proj(x).chunk(2, -1); h * gelu(gate)``."""
    from .standin import fused

    return fused.geglu(self.proj(hidden_states))


def _plain_resnet(m) -> bool:
    return (getattr(m, "time_embedding_norm", "default") == "default" and not getattr(m, "up", False)
            and not getattr(m, "down", False) and getattr(m, "upsample", None) is None
            and getattr(m, "downsample", None) is None and isinstance(getattr(m, "norm1", None), torch.nn.GroupNorm)
            and isinstance(getattr(m, "norm2", None), torch.nn.GroupNorm)
            and isinstance(getattr(m, "nonlinearity", None), torch.nn.SiLU)
            and getattr(m, "time_emb_proj", None) is not None and not getattr(m, "skip_time_act", False))


def _resnet_forward(self, input_tensor, temb=None, *args, **kwargs):
    """diffusers.models.resnet.ResnetBlock2D.forward for the SD1.x configuration: norm1 -> SiLU -> conv1 -> (+ time
    embedding) -> norm2 -> SiLU -> dropout -> conv2 -> (+ shortcut) / output_scale_factor."""
    from .standin import fused

    if temb is None or not _plain_resnet(self):
        return self._lora_amd_forward(input_tensor, temb, *args, **kwargs)
    x = input_tensor
    n1 = fused.group_norm_act(x, self.norm1)
    t = self.time_emb_proj(self.nonlinearity(temb))  # [B, C_out]
    h = self.conv1(n1)
    h = self.conv2(self.dropout(fused.group_norm_act(h, self.norm2, addend=t)))
    sc = x if getattr(self, "conv_shortcut", None) is None else self.conv_shortcut(x)
    out = sc + h
    osf = getattr(self, "output_scale_factor", 1.0)
    return out / osf if osf != 1.0 else out


def _plain_block(m) -> bool:
    ln = torch.nn.LayerNorm
    return (isinstance(getattr(m, "norm1", None), ln) and isinstance(getattr(m, "norm2", None), ln)
            and isinstance(getattr(m, "norm3", None), ln) and getattr(m, "attn2", None) is not None
            and getattr(m, "norm_type", "layer_norm") == "layer_norm"
            and not getattr(m, "use_ada_layer_norm", False) and not getattr(m, "use_ada_layer_norm_zero", False)
            and not getattr(m, "use_ada_layer_norm_single", False) and not getattr(m, "only_cross_attention", False)
            and getattr(m, "pos_embed", None) is None and getattr(m, "_chunk_size", None) is None
            and getattr(m, "fuser", None) is None)


def _block_forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                   timestep=None, cross_attention_kwargs=None, class_labels=None, added_cond_kwargs=None, **kwargs):
    """diffusers.models.attention.BasicTransformerBlock.forward for the SD1.x configuration (LayerNorm, self-attention,
    cross-attention, GEGLU feed-forward, three residuals): every residual add runs inside the LayerNorm that follows it."""
    from .standin import fused

    if (attention_mask is not None or encoder_attention_mask is not None or cross_attention_kwargs or kwargs
            or class_labels is not None or added_cond_kwargs is not None or encoder_hidden_states is None
            or not _plain_block(self)):
        return self._lora_amd_forward(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask,
                                      timestep, cross_attention_kwargs, class_labels, added_cond_kwargs, **kwargs)
    x = hidden_states
    x, n = fused.add_layer_norm(self.attn1(fused.layer_norm(x, self.norm1)), x, self.norm2)
    x, n = fused.add_layer_norm(self.attn2(n, encoder_hidden_states=encoder_hidden_states), x, self.norm3)
    return self.ff(n) + x


def install_host_options(unet) -> dict:
    """Bind the stand-in's frozen-host passes into a diffusers UNet (see the module docstring).  Returns how many blocks
    of each kind were bound.  Idempotent; ``LORA_AMD_HOSTOPS=0`` makes the bound passes take their ATen sequences."""
    counts = {"attention": 0, "transformer_block": 0, "resnet": 0, "geglu": 0}
    for m in unet.modules():
        name = type(m).__name__
        if hasattr(m, "set_processor") and hasattr(m, "to_q") and hasattr(m, "heads"):
            prev = getattr(m, "processor", None)
            if isinstance(prev, LoraAmdAttnProcessor):
                prev.tuned = True
            else:
                m.set_processor(LoraAmdAttnProcessor(prev, tuned=True))
            counts["attention"] += 1
            continue
        if "_lora_amd_forward" in m.__dict__:
            continue
        if name == "GEGLU" and hasattr(m, "proj"):
            fwd, kind = _geglu_forward, "geglu"
        elif name == "ResnetBlock2D" and _plain_resnet(m):
            fwd, kind = _resnet_forward, "resnet"
        elif name == "BasicTransformerBlock" and _plain_block(m):
            fwd, kind = _block_forward, "transformer_block"
        else:
            continue
        m.__dict__["_lora_amd_forward"] = m.forward
        m.forward = types.MethodType(fwd, m)
        counts[kind] += 1
    return counts
