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


EMBED = "<embed>"


def _uniform_rank(metadata: dict) -> int:
    """The single rank a LoRA file uses (0 if it records none); mixed ranks inside one file are refused (ref :22-27)."""
    ranks = {int(value) for name, value in metadata.items() if name.endswith("rank")}
    assert len(ranks) <= 1, "Rank should be the same per model"
    return ranks.pop() if ranks else 0


def lora_join(lora_safetenors: list):
    """Rank-concatenate N LoRA files (ref :13-71): ``down`` factors stack along dim 0, ``up`` factors along dim 1, so
    member ``i`` owns the rank slice ``[sum(ranks[:i]), sum(ranks[:i+1]))`` of every site — the slice its entry of
    ``LoRAManager.tune`` scales through the diagonal selector.  Learned tokens are renamed ``<s{file}-{k}>``.
    Returns ``(tensors, metadata, ranks, token counts)``."""
    ranklist: List[int] = [_uniform_rank(dict(f.metadata())) for f in lora_safetenors]
    total_rank = sum(ranklist)

    total_metadata = {}
    for f in lora_safetenors:  # later files win on equal keys, as dict.update does in the reference
        total_metadata.update({name: value for name, value in dict(f.metadata()).items() if value != EMBED})

    total_tensor = {}
    factor_keys = {key for f in lora_safetenors for key in f.keys() if key.startswith(("text_encoder", "unet"))}
    for key in factor_keys:
        axis = 0 if key.endswith("down") else 1
        stacked = torch.cat([f.get_tensor(key) for f in lora_safetenors], dim=axis)
        assert stacked.shape[axis] == total_rank
        total_tensor[key] = stacked
        total_metadata[key.rsplit(":", 1)[0] + ":rank"] = str(total_rank)

    token_size_list = []
    for file_idx, f in enumerate(lora_safetenors):
        learned = sorted(name for name, value in f.metadata().items() if value == EMBED)
        for k, token in enumerate(learned):
            alias = f"<s{file_idx}-{k}>"
            total_tensor[alias] = f.get_tensor(token)
            total_metadata[alias] = EMBED
            print(f"Embedding {token} replaced to {alias}")
        token_size_list.append(len(learned))
    return total_tensor, total_metadata, ranklist, token_size_list


class DummySafeTensorObject:
    """In-memory stand-in for a ``safe_open`` handle (ref :74-86): just what ``parse_safeloras*`` touches."""

    def __init__(self, tensor: dict, metadata):
        self.tensor, self._metadata = tensor, metadata

    def keys(self):
        return self.tensor.keys()

    def metadata(self):
        return self._metadata

    def get_tensor(self, key):
        return self.tensor[key]


class LoRAManager:
    """Several LoRA files patched into one pipeline as ONE rank-concatenated adapter per site; per-member weights are
    the diagonal selector of that adapter (ref :89-144).  ``pipe`` needs ``.unet``, ``.text_encoder`` and (for learned
    tokens) ``.tokenizer``."""

    def __init__(self, lora_paths_list: List[str], pipe):
        self.lora_paths_list, self.pipe = lora_paths_list, pipe
        self._setup()

    def _setup(self):
        self._lora_safetenors = [safe_open(path, framework="pt", device="cpu") for path in self.lora_paths_list]
        tensors, metadata, self.ranklist, self.token_size_list = lora_join(self._lora_safetenors)
        self.total_safelora = DummySafeTensorObject(tensors, metadata)
        monkeypatch_or_replace_safeloras(self.pipe, self.total_safelora)
        learned = parse_safeloras_embeds(self.total_safelora)
        if learned:
            apply_learned_embed_in_clip(learned, self.pipe.text_encoder, self.pipe.tokenizer, token=None,
                                        idempotent=True)

    def tune(self, scales):
        """One weight per member file, repeated over that member's rank slice."""
        assert len(scales) == len(self.ranklist), "Scale list should be the same length as ranklist"
        diag = torch.repeat_interleave(torch.as_tensor(scales, dtype=torch.float32), torch.as_tensor(self.ranklist))
        set_lora_diag(self.pipe.unet, diag)

    def prompt(self, prompt):
        """``<1>``, ``<2>``, ... expand to the renamed learned tokens of member 1, 2, ..."""
        if prompt is None:
            return None
        for idx, n_tokens in enumerate(self.token_size_list):
            prompt = prompt.replace(f"<{idx + 1}>", "".join(f"<s{idx}-{k}>" for k in range(n_tokens)))
        return prompt
