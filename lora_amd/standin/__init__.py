"""Stand-ins for the reference's un-vendored host-model dependencies (diffusers is not installed here)."""
from __future__ import annotations

from typing import List, Tuple

import torch

from .scheduler import DDPMScheduler  # noqa: F401
from .unet import UNet2DConditionModel, sd15_unet, tiny_unet  # noqa: F401


def clip_text_model(hidden: int = 768, layers: int = 12, heads: int = 12, vocab: int = 49408, max_pos: int = 77):
    """Random-init ``transformers.CLIPTextModel`` with SD1.5's text-encoder geometry (no hub access here)."""
    from transformers import CLIPTextConfig, CLIPTextModel

    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=hidden * 4, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=max_pos, hidden_act="quick_gelu",
                         projection_dim=hidden)
    return CLIPTextModel(cfg)


def sd15_lora_site_shapes(extended: bool = False) -> List[Tuple[int, int]]:
    """(N, K) of every frozen weight the default (or extended) UNet injection touches, in index order.
    Conv sites are flattened to [C_out, C_in*kh*kw].  Built on the meta device: no memory."""
    from .. import lora as L

    with torch.device("meta"):
        unet = sd15_unet()
    targets = L.UNET_EXTENDED_TARGET_REPLACE if extended else L.UNET_DEFAULT_TARGET_REPLACE
    kinds = [torch.nn.Linear, torch.nn.Conv2d] if extended else [torch.nn.Linear]
    out = []
    for _, _, m in L._find_modules(unet, targets, search_class=kinds):
        w = m.weight
        out.append((w.shape[0], w.numel() // w.shape[0]))
    return out
