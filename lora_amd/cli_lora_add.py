"""Drop-in for the reference's ``lora_add`` console script (``lora_diffusion/cli_lora_add.py:24-183``).

* ``lpl``  LoRA + LoRA -> LoRA: ``alpha_1 * A + alpha_2 * B`` over the two files (``.pt`` lists or ``.safetensors``);
* ``upl``  model + LoRA -> model: patch, ``collapse_lora`` (ONE launch of the fused ``W + alpha*up@down`` HIP kernel per
           model when the weights are on the device), strip the adapters, save;
* ``ljl``  LoRA join LoRA: rank concatenation (``lora_manager.lora_join``);
* ``upl-ckpt-v2`` needs the CompVis checkpoint converter (out of scope here).
"""
from __future__ import annotations

import os
import sys

import torch

from .lora import (_text_lora_path, collapse_lora, monkeypatch_remove_lora, patch_pipe)
from .lora_manager import lora_join

try:
    from safetensors import safe_open
    from safetensors.torch import save_file
except ImportError:  # pragma: no cover
    safe_open = save_file = None


def _load_pipeline(path: str, device: str):
    if path.startswith("standin"):
        import types

        from .standin import clip_text_model, tiny_unet
        from .standin.io import StandinTokenizer

        seed = int(path.split(":", 1)[1]) if ":" in path else 0
        torch.manual_seed(seed)
        pipe = types.SimpleNamespace(unet=tiny_unet().to(device),
                                     text_encoder=clip_text_model(hidden=32, layers=2, heads=2).to(device),
                                     tokenizer=StandinTokenizer())
        pipe.save_pretrained = lambda out: (os.makedirs(out, exist_ok=True),
                                            torch.save(pipe.unet.state_dict(), os.path.join(out, "unet.pt")),
                                            torch.save(pipe.text_encoder.state_dict(),
                                                       os.path.join(out, "text_encoder.pt")))
        return pipe
    from diffusers import StableDiffusionPipeline

    return StableDiffusionPipeline.from_pretrained(path).to(device)


def add(path_1: str, path_2: str, output_path: str, alpha_1: float = 0.5, alpha_2: float = 0.5, mode: str = "lpl",
        with_text_lora: bool = False, device: str = "cpu"):
    """ref :24-183 (``device`` is an addition: ``cuda:0`` runs the ``upl`` merge on the HIP kernel)."""
    print("Lora Add, mode " + mode)
    if mode == "lpl":
        if path_1.endswith(".pt") and path_2.endswith(".pt"):
            jobs = [(path_1, path_2, output_path, "unet")]
            if with_text_lora:
                jobs.append((_text_lora_path(path_1), _text_lora_path(path_2), _text_lora_path(output_path),
                             "text_encoder"))
            for p1, p2, out, opt in jobs:
                if opt == "text_encoder" and not (os.path.exists(p1) and os.path.exists(p2)):
                    print(f"No text encoder found in {p1} / {p2}, skipping...")
                    continue
                print("Loading", p1, p2)
                l1, l2 = torch.load(p1), torch.load(p2)
                out_list = [alpha_1 * a.data + alpha_2 * b.data for a, b in zip(l1, l2)]
                print(f"Saving merged {opt} to", out)
                torch.save(out_list, out)
        elif path_1.endswith(".safetensors") and path_2.endswith(".safetensors"):
            s1 = safe_open(path_1, framework="pt", device="cpu")
            s2 = safe_open(path_2, framework="pt", device="cpu")
            metadata = dict(s1.metadata())
            metadata.update(dict(s2.metadata()))
            ret = {}
            for key in set(list(s1.keys()) + list(s2.keys())):
                if key.startswith("text_encoder") or key.startswith("unet"):
                    ret[key] = alpha_1 * s1.get_tensor(key) + alpha_2 * s2.get_tensor(key)
                else:
                    ret[key] = s1.get_tensor(key) if key in s1.keys() else s2.get_tensor(key)
            save_file(ret, output_path, metadata)
        else:
            raise ValueError("lpl needs two .pt or two .safetensors files")
    elif mode == "upl":
        print(f"Merging UNET/CLIP from {path_1} with LoRA from {path_2} to {output_path}. Merging ratio : {alpha_1}.")
        pipe = _load_pipeline(path_1, device)
        patch_pipe(pipe, path_2)
        collapse_lora(pipe.unet, alpha_1)
        collapse_lora(pipe.text_encoder, alpha_1)
        monkeypatch_remove_lora(pipe.unet)
        monkeypatch_remove_lora(pipe.text_encoder)
        pipe.save_pretrained(output_path)
    elif mode == "upl-ckpt-v2":
        raise NotImplementedError("upl-ckpt-v2 needs the diffusers -> CompVis .ckpt converter (to_ckpt_v2), out of scope")
    elif mode == "ljl":
        print("Using Join mode : alpha will not have an effect here.")
        assert path_1.endswith(".safetensors") and path_2.endswith(".safetensors"), \
            "Only .safetensors files are supported"
        s1 = safe_open(path_1, framework="pt", device="cpu")
        s2 = safe_open(path_2, framework="pt", device="cpu")
        total_tensor, total_metadata, _, _ = lora_join([s1, s2])
        save_file(total_tensor, output_path, total_metadata)
    else:
        print("Unknown mode", mode)
        raise ValueError(f"Unknown mode {mode}")


def main():
    from .cli_lora_pti import _parse_cli

    add(**_parse_cli(sys.argv[1:]))


if __name__ == "__main__":
    main()
