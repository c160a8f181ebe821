"""Stand-ins for the host-side pieces the CLIs take from un-vendored packages: the VAE encoder
(``diffusers.AutoencoderKL``), the CLIP tokenizer vocabulary (hub files) and the image dataset
(``torchvision.transforms``).  They exist so that ``training_scripts/train_lora_dreambooth.py`` and
``lora_amd.cli_lora_pti`` run end to end in an image without ``diffusers``/``torchvision`` and without
network access; when the real packages and a real checkpoint directory are present the CLIs use those.

None of this is on the graded hot path: it produces tensors of the right shape, dtype and scale
(latents ``[B,4,H/8,W/8]`` * 0.18215, token ids ``[B,77]``) for the LoRA step to consume.
"""
from __future__ import annotations

import hashlib
import os
import types
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn


class StandinVAE(nn.Module):
    """Fixed (seeded, frozen) 8x-downsampling encoder: 8x8 average pooling, then a 3->4 channel mix and a small
    learned-looking texture term.  ``encode(x).latent_dist.sample()`` like AutoencoderKL (ref
    train_lora_dreambooth.py:818-821); latents have roughly unit variance before the 0.18215 scaling."""

    def __init__(self, seed: int = 1234):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("mix", torch.randn(4, 3, generator=g) * 1.2)
        self.register_buffer("tex", torch.randn(4, 3, 8, 8, generator=g) * 0.35)
        self.config = types.SimpleNamespace(scaling_factor=0.18215)

    class _Dist:
        def __init__(self, mean: torch.Tensor):
            self.mean = mean

        def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
            return self.mean + 0.05 * torch.randn(self.mean.shape, device=self.mean.device, dtype=self.mean.dtype,
                                                  generator=generator)

        def mode(self):
            return self.mean

    @torch.no_grad()
    def encode(self, pixels: torch.Tensor):
        x = pixels.float()
        pooled = torch.nn.functional.avg_pool2d(x, 8)
        mean = torch.einsum("oc,bchw->bohw", self.mix, pooled) + torch.nn.functional.conv2d(x, self.tex, stride=8)
        return types.SimpleNamespace(latent_dist=self._Dist(mean.to(pixels.dtype)))


class StandinTokenizer:
    """Deterministic word-hash tokenizer with CLIP's framing: ``<|startoftext|>`` 49406, ``<|endoftext|>`` 49407,
    77 positions, pad = eot.  Supports ``add_tokens`` / ``convert_tokens_to_ids`` / ``encode`` the way the TI code
    uses them (ref lora.py:899-942, cli_lora_pti.py:49-128)."""

    bos_token_id, eos_token_id, model_max_length = 49406, 49407, 77

    def __init__(self, vocab_size: int = 49408):
        self.base_vocab = vocab_size
        self.added: dict = {}

    def __len__(self):
        return self.base_vocab + len(self.added)

    def _word_id(self, w: str) -> int:
        if w in self.added:
            return self.added[w]
        h = int.from_bytes(hashlib.sha1(w.lower().encode()).digest()[:4], "little")
        return 1000 + h % (self.bos_token_id - 1000)

    def add_tokens(self, tokens) -> int:
        if isinstance(tokens, str):
            tokens = [tokens]
        n = 0
        for t in tokens:
            if t not in self.added:
                self.added[t] = self.base_vocab + len(self.added)
                n += 1
        return n

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self._word_id(tokens)
        return [self._word_id(t) for t in tokens]

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        if self.added:  # added tokens are matched verbatim wherever they occur, as HF tokenizers do
            import re

            pat = "(" + "|".join(re.escape(t) for t in sorted(self.added, key=len, reverse=True)) + ")"
            text = " ".join(p for p in re.split(pat, text) if p)
        ids = [self._word_id(w) for w in text.replace(",", " , ").split()]
        return [self.bos_token_id] + ids + [self.eos_token_id] if add_special_tokens else ids

    def __call__(self, text, padding="do_not_pad", truncation=True, max_length=None, return_tensors=None):
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)
            if truncation and len(ids) > L:
                ids = ids[: L - 1] + [self.eos_token_id]
            if padding == "max_length":
                ids = ids + [self.eos_token_id] * (L - len(ids))
            rows.append(ids)
        if return_tensors == "pt":
            width = max(len(r) for r in rows)
            rows = [r + [self.eos_token_id] * (width - len(r)) for r in rows]
            return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
        return types.SimpleNamespace(input_ids=rows[0] if single else rows)

    def pad(self, encoded, padding=True, max_length=None, return_tensors="pt"):
        rows = encoded["input_ids"]
        width = max(len(r) for r in rows)
        if padding == "max_length":
            width = max(width, max_length or self.model_max_length)
        rows = [list(r) + [self.eos_token_id] * (width - len(r)) for r in rows]
        return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))


IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


def list_images(root: str) -> List[str]:
    return sorted(os.path.join(root, f) for f in os.listdir(root) if f.lower().endswith(IMG_EXT))


def load_image(path: str, size: int, center_crop: bool, rng: np.random.Generator, resize: bool = True) -> torch.Tensor:
    """RGB image -> ``[3, size, size]`` in [-1, 1] (what torchvision's Resize / Crop / ToTensor / Normalize([0.5],[0.5])
    chain of the reference's dataset produces, ref train_lora_dreambooth.py:96-113)."""
    from PIL import Image

    img = Image.open(path)
    if img.mode != "RGB":
        img = img.convert("RGB")
    if resize:
        w, h = img.size
        s = size / min(w, h)
        img = img.resize((max(size, round(w * s)), max(size, round(h * s))), Image.BILINEAR)
    w, h = img.size
    if w < size or h < size:
        img = img.resize((max(w, size), max(h, size)), Image.BILINEAR)
        w, h = img.size
    if center_crop:
        x0, y0 = (w - size) // 2, (h - size) // 2
    else:
        x0, y0 = int(rng.integers(0, w - size + 1)), int(rng.integers(0, h - size + 1))
    arr = np.asarray(img.crop((x0, y0, x0 + size, y0 + size)), dtype=np.float32) / 127.5 - 1.0
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous()


def synthetic_images(n: int, size: int, seed: int = 0) -> List[torch.Tensor]:
    """``n`` uniform[-1,1] images (BASELINE configs[0]: "4 synthetic 512x512 images")."""
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(3, size, size, generator=g) * 2 - 1 for _ in range(n)]


class DreamBoothDataset(torch.utils.data.Dataset):
    """Instance (+ class) images with their prompts (ref train_lora_dreambooth.py:51-145).  ``instance_data_root`` may
    be ``synthetic:N`` for N generated images; likewise ``class_data_root``."""

    def __init__(self, instance_data_root: str, instance_prompt: str, tokenizer, class_data_root: Optional[str] = None,
                 class_prompt: Optional[str] = None, size: int = 512, center_crop: bool = False, resize: bool = True,
                 seed: int = 0, content_seed: Optional[int] = None):
        """``seed``: this process's augmentation stream (random crops; per rank).  ``content_seed``: what ``synthetic:N``
        images are generated from — the SAME on every rank of a data-parallel run, so that a DistributedSampler partitions
        one dataset (default: ``seed``)."""
        self.size, self.center_crop, self.resize, self.tokenizer = size, center_crop, resize, tokenizer
        self.rng = np.random.default_rng(seed)
        seed = seed if content_seed is None else content_seed
        self.instance = self._source(instance_data_root, seed)
        if len(self.instance) == 0:
            raise ValueError("Instance images root doesn't exists.")
        self.instance_prompt = instance_prompt
        self._length = len(self.instance)
        self.klass = None
        if class_data_root is not None:
            self.klass = self._source(class_data_root, seed + 1)
            self._length = max(len(self.klass), len(self.instance))
            self.class_prompt = class_prompt

    def _source(self, root: str, seed: int) -> Sequence:
        if root.startswith("synthetic:"):
            return synthetic_images(int(root.split(":", 1)[1]), self.size, seed)
        if not os.path.isdir(root):
            raise ValueError("Instance images root doesn't exists.")
        return list_images(root)

    def __len__(self):
        return self._length

    def _image(self, item) -> torch.Tensor:
        if torch.is_tensor(item):
            return item
        return load_image(item, self.size, self.center_crop, self.rng, self.resize)

    def _ids(self, prompt: str):
        return self.tokenizer(prompt, padding="do_not_pad", truncation=True,
                              max_length=self.tokenizer.model_max_length).input_ids

    def __getitem__(self, index):
        ex = {"instance_images": self._image(self.instance[index % len(self.instance)]),
              "instance_prompt_ids": self._ids(self.instance_prompt)}
        if self.klass is not None:
            ex["class_images"] = self._image(self.klass[index % len(self.klass)])
            ex["class_prompt_ids"] = self._ids(self.class_prompt)
        return ex


def collate(examples, tokenizer, with_prior_preservation: bool):
    """Instance and class examples are concatenated into ONE batch (ref :693-719): one forward for both."""
    ids = [e["instance_prompt_ids"] for e in examples]
    px = [e["instance_images"] for e in examples]
    if with_prior_preservation:
        ids += [e["class_prompt_ids"] for e in examples]
        px += [e["class_images"] for e in examples]
    px = torch.stack(px).contiguous().float()
    ids = tokenizer.pad({"input_ids": ids}, padding="max_length", max_length=tokenizer.model_max_length,
                        return_tensors="pt").input_ids
    return {"input_ids": ids, "pixel_values": px}


def load_host_models(path: str, vae_path: Optional[str], revision: Optional[str], tokenizer_name: Optional[str], device,
                     standin: str = "sd15", seed: Optional[int] = None):
    """(tokenizer, text_encoder, vae, unet, noise_scheduler, description).

    The real ``diffusers`` / ``transformers`` modules when ``path`` is a checkpoint directory and ``diffusers`` is
    importable (ref train_lora_dreambooth.py:566-594, 678-680; cli_lora_pti.py:49-128); otherwise the stand-ins of this
    package: SD1.5-shaped (or tiny) random-init UNet, random-init CLIP text encoder from ``transformers``' config
    class, the fixed stand-in VAE encoder and the hash tokenizer."""
    try:
        import diffusers  # noqa: F401
        real = os.path.isdir(path)
    except ImportError:
        real = False
    if real:
        from diffusers import AutoencoderKL, DDPMScheduler, UNet2DConditionModel
        from transformers import CLIPTextModel, CLIPTokenizer

        tok = CLIPTokenizer.from_pretrained(tokenizer_name or path, subfolder=None if tokenizer_name else "tokenizer",
                                            revision=revision)
        te = CLIPTextModel.from_pretrained(path, subfolder="text_encoder", revision=revision)
        vae = AutoencoderKL.from_pretrained(vae_path or path, subfolder=None if vae_path else "vae",
                                            revision=None if vae_path else revision)
        unet = UNet2DConditionModel.from_pretrained(path, subfolder="unet", revision=revision)
        sched = DDPMScheduler.from_config(path, subfolder="scheduler")
        from ..diffusers_glue import install_attention_processor, install_host_options

        n_attn = install_attention_processor(unet)  # q/k/v of a block in one launch once the adapters are injected
        bound = install_host_options(unet)  # GroupNorm(+SiLU), add+LayerNorm, GEGLU, attention-kernel choice / head padding
        return tok, te, vae, unet, sched, (f"diffusers checkpoint {path} ({n_attn} attention blocks on LoraAmdAttnProcessor; "
                                           f"host passes bound: {bound})")
    from . import DDPMScheduler, clip_text_model, sd15_unet, tiny_unet

    if seed is None:
        torch.manual_seed(0)
    tok = StandinTokenizer()
    if standin == "tiny":
        te = clip_text_model(hidden=32, layers=2, heads=2)
        unet = tiny_unet(cross_attention_dim=32)
    else:
        te = clip_text_model()
        with torch.device("meta"):
            unet = sd15_unet()
        unet.to_empty(device=device)
        g = torch.Generator(device=device).manual_seed(0 if seed is None else seed)
        with torch.no_grad():
            for name, prm in unet.named_parameters():
                if prm.dim() > 1:
                    prm.normal_(0.0, 0.02, generator=g)
                elif name.endswith("weight"):
                    prm.fill_(1.0)
                else:
                    prm.zero_()
    return tok, te, StandinVAE(), unet, DDPMScheduler(), f"stand-in models ({standin}; random init, no checkpoint)"


# ----------------------------------------------------------------------------- pivotal-tuning dataset
# A few caption templates per mode (the reference carries the textual-inversion template lists, dataset.py:12-70).
TEMPLATES = {
    "object": ["a photo of a {}", "a rendering of a {}", "a cropped photo of the {}", "a close-up photo of a {}",
               "a bright photo of the {}", "a good photo of a {}", "a photo of the small {}", "a photo of one {}"],
    "style": ["a painting in the style of {}", "a rendering in the style of {}", "a picture in the style of {}",
              "a close-up painting in the style of {}", "a bright painting in the style of {}"],
    "null": ["{}"],
}


class PivotalTuningDataset(torch.utils.data.Dataset):
    """Images + captions for pivotal tuning (role of dataset.py:119-311).  Caption = a template around the joined
    placeholder tokens (``use_template``), or the file stem with ``token_map`` substitutions; optional
    ``{idx}.mask.png`` loss masks with ``use_mask_captioned_data``.  ``instance_data_root`` may be ``synthetic:N``."""

    def __init__(self, instance_data_root: str, tokenizer, token_map: Optional[dict] = None,
                 use_template: Optional[str] = None, size: int = 512, h_flip: bool = True, resize: bool = True,
                 use_mask_captioned_data: bool = False, seed: int = 0, content_seed: Optional[int] = None):
        # seed: this process's augmentation stream (flips; per rank); content_seed: the synthetic images (same on every rank)
        assert not (use_mask_captioned_data and use_template), "Can't use both mask caption data and template."
        self.size, self.tokenizer, self.resize, self.h_flip = size, tokenizer, resize, h_flip
        self.token_map, self.use_template = token_map or {}, use_template
        self.rng = np.random.default_rng(seed)
        self.masks: List[Optional[str]] = []
        if instance_data_root.startswith("synthetic:"):
            n = int(instance_data_root.split(":", 1)[1])
            self.items = synthetic_images(n, size, seed if content_seed is None else content_seed)
            self.captions = [f"synthetic image {i} of DUMMY" for i in range(n)]
            self.masks = [None] * n
        else:
            if not os.path.isdir(instance_data_root):
                raise ValueError("Instance images root doesn't exists.")
            if use_mask_captioned_data:
                srcs = sorted(f for f in list_images(instance_data_root) if f.endswith("src.jpg"))
                self.items, self.masks = [], []
                for f in srcs:
                    idx = int(os.path.basename(f).split(".")[0])
                    m = os.path.join(instance_data_root, f"{idx}.mask.png")
                    if os.path.exists(m):
                        self.items.append(f)
                        self.masks.append(m)
                    else:
                        print(f"Mask not found for {f}")
                self.captions = [ln.strip() for ln in open(os.path.join(instance_data_root, "caption.txt"))]
            else:
                self.items = [f for f in list_images(instance_data_root) if not f.endswith("mask.png")]
                self.captions = [os.path.basename(f).split(".")[0] for f in self.items]
                self.masks = [None] * len(self.items)
            assert len(self.items) > 0, "No images found in the instance data root."

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        i = index % len(self.items)
        item = self.items[i]
        img = item if torch.is_tensor(item) else load_image(item, self.size, True, self.rng, self.resize)
        if self.use_template:
            assert self.token_map, "token_map should be specified when using template"
            text = TEMPLATES[self.use_template][int(self.rng.integers(len(TEMPLATES[self.use_template])))].format(
                "".join(self.token_map.values()))
        else:
            text = self.captions[i % len(self.captions)].strip()
            for tok, val in self.token_map.items():
                text = text.replace(tok, val)
        ex = {"instance_images": img}
        if self.masks[i] is not None:
            from PIL import Image

            m = Image.open(self.masks[i]).convert("L").resize((self.size, self.size))
            ex["mask"] = torch.from_numpy(np.asarray(m, dtype=np.float32) / 255.0)[None]
        if self.h_flip and self.rng.random() > 0.5:
            ex["instance_images"] = torch.flip(ex["instance_images"], dims=[2])
            if "mask" in ex:
                ex["mask"] = torch.flip(ex["mask"], dims=[2])
        ex["instance_prompt_ids"] = self.tokenizer(text, padding="do_not_pad", truncation=True,
                                                   max_length=self.tokenizer.model_max_length).input_ids
        return ex
