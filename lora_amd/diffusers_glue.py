"""Glue for a real ``diffusers`` UNet (the host model of the reference: train_lora_dreambooth.py:590-594,
cli_lora_pti.py:116-120; un-vendored, absent from this image — exercised with a fake module in tests/).

``diffusers`` attention blocks (class ``Attention`` / ``CrossAttention``) call an *attention processor*; installing
:class:`LoraAmdAttnProcessor` makes their to_q / to_k / to_v projections — three ``LoraInjectedLinear`` reading one
tensor after ``inject_trainable_lora`` — go out as ONE weight-stationary launch (``lora.lora_linear_group``) exactly as
the stand-in ``CrossAttention`` does, and runs the dense softmax(QK^T)V on the library's fused kernel.  Anything the
processor does not understand (attention masks with unusual shapes, added-KV / norm_cross variants) is handed back to
the block's previous processor."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .lora import lora_linear_group


class LoraAmdAttnProcessor:
    """Drop-in for ``diffusers.models.attention_processor.AttnProcessor2_0`` on SD1.x-style attention blocks."""

    def __init__(self, fallback=None):
        self.fallback = fallback

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
        o = F.scaled_dot_product_attention(q, k, v, scale=getattr(attn, "scale", None))
        o = o.transpose(1, 2).reshape(B, T, -1).to(q.dtype)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
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
