"""Drop-in for ``lora_diffusion/lora_manager.py``: rank-concatenation of several LoRA files (``lora_join``) and the
``LoRAManager`` that patches the joined LoRA into a pipeline and re-weights the members at inference time through the
diagonal selector between ``lora_down`` and ``lora_up``.  On device the selector is the ``sel`` operand of
the HIP adapter kernels (``lora_amd_linear_fwd`` / ``lora_amd_conv_down_fwd``) — no extra launch per member LoRA."""
from __future__ import annotations

from typing import List

import torch

from .lora import (apply_learned_embed_in_clip, monkeypatch_or_replace_safeloras, parse_safeloras_embeds, set_lora_diag)

try:
    from safetensors import safe_open
except ImportError:  # pragma: no cover
    from .safe_open import safe_open


def lora_join(lora_safetenors: list):
    """ref :13-71 — concatenate N LoRAs along the rank axis (``down`` dim 0, ``up`` dim 1); every member must use one
    rank throughout; learned tokens are renamed ``<s{file}-{k}>``.  Returns (tensors, metadata, ranks, token counts)."""
    metadatas = [dict(s.metadata()) for s in lora_safetenors]
    merged_meta, total_metadata, total_tensor = {}, {}, {}
    ranklist: List[int] = []
    for md in metadatas:
        ranks = {int(v) for k, v in md.items() if k.endswith("rank")}
        assert len(ranks) <= 1, "Rank should be the same per model"
        ranklist.append(ranks.pop() if ranks else 0)
        merged_meta.update(md)
    total_rank = sum(ranklist)
    for k, v in merged_meta.items():
        if v != "<embed>":
            total_metadata[k] = v
    keys = set()
    for s in lora_safetenors:
        keys.update(s.keys())
    for key in keys:
        if key.startswith("text_encoder") or key.startswith("unet"):
            parts = [s.get_tensor(key) for s in lora_safetenors]
            dim = 0 if key.endswith("down") else 1
            joined = torch.cat(parts, dim=dim)
            assert joined.shape[dim] == total_rank
            total_tensor[key] = joined
            total_metadata[":".join(key.split(":")[:-1]) + ":rank"] = str(total_rank)
    token_size_list = []
    for idx, s in enumerate(lora_safetenors):
        tokens = [k for k, v in s.metadata().items() if v == "<embed>"]
        for jdx, token in enumerate(sorted(tokens)):
            total_tensor[f"<s{idx}-{jdx}>"] = s.get_tensor(token)
            total_metadata[f"<s{idx}-{jdx}>"] = "<embed>"
            print(f"Embedding {token} replaced to <s{idx}-{jdx}>")
        token_size_list.append(len(tokens))
    return total_tensor, total_metadata, ranklist, token_size_list


class DummySafeTensorObject:  # ref :74-86
    def __init__(self, tensor: dict, metadata):
        self.tensor = tensor
        self._metadata = metadata

    def keys(self):
        return self.tensor.keys()

    def metadata(self):
        return self._metadata

    def get_tensor(self, key):
        return self.tensor[key]


class LoRAManager:
    """ref :89-144.  ``pipe`` needs ``.unet``, ``.text_encoder`` and (for learned tokens) ``.tokenizer``."""

    def __init__(self, lora_paths_list: List[str], pipe):
        self.lora_paths_list = lora_paths_list
        self.pipe = pipe
        self._setup()

    def _setup(self):
        self._lora_safetenors = [safe_open(path, framework="pt", device="cpu") for path in self.lora_paths_list]
        total_tensor, total_metadata, self.ranklist, self.token_size_list = lora_join(self._lora_safetenors)
        self.total_safelora = DummySafeTensorObject(total_tensor, total_metadata)
        monkeypatch_or_replace_safeloras(self.pipe, self.total_safelora)
        tok_dict = parse_safeloras_embeds(self.total_safelora)
        if tok_dict:
            apply_learned_embed_in_clip(tok_dict, self.pipe.text_encoder, self.pipe.tokenizer, token=None,
                                        idempotent=True)

    def tune(self, scales):
        assert len(scales) == len(self.ranklist), "Scale list should be the same length as ranklist"
        diags = []
        for scale, rank in zip(scales, self.ranklist):
            diags = diags + [scale] * rank
        set_lora_diag(self.pipe.unet, torch.tensor(diags))

    def prompt(self, prompt):
        if prompt is not None:
            for idx, tok_size in enumerate(self.token_size_list):
                prompt = prompt.replace(f"<{idx + 1}>", "".join([f"<s{idx}-{jdx}>" for jdx in range(tok_size)]))
        return prompt
