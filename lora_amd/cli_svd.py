"""Drop-in for the reference's ``lora_distill`` console script (``lora_diffusion/cli_svd.py``): distil the difference
between a fine-tuned model and its base into rank-r LoRA factors, site by site — BASELINE configs[4] / SURVEY §8f-1.

Reference recipe per site (cli_svd.py:30-53 Linear, :55-92 Conv2d flattened from dim 1): full ``torch.linalg.svd`` of
``dW = W_tuned - W_base`` in f32, keep the top r triplets, ``up = U_r diag(S_r)``, ``down = Vh_r``, clamp both at the
0.99-quantile of their joint value distribution.

MI355X path (device tensors): only r of min(N, K) singular triplets are wanted, so a full SVD (O(N K min(N,K)) flops,
10240 x 1280 at the largest site) is replaced by randomized subspace iteration (Halko-Martinsson-Tropp) whose heavy
steps are passes over dW with a skinny factor — exactly the HBM-streaming primitives of ``csrc/linear.hip``:

    Y = dW @ Omega^T          ``lora_amd_rowdot``    [N, l]   l = r + oversample
    Z = Q^T @ dW              ``lora_amd_colreduce`` [l, K]

with thin QR factorisations ([N, l], [K, l]) and one small SVD ([l, K]) in between.  With ``n_iter`` power iterations
the captured subspace error decays like (s_{l+1}/s_r)^(2 n_iter + 1); the defaults reproduce ``up @ down`` of the
reference to ~1e-4 relative on distillation-like spectra.  The signs of singular vectors are arbitrary (LAPACK's are
too, and the reference's clamp threshold - a quantile of SIGNED entries - inherits that arbitrariness); the device
path fixes them by making the largest-magnitude entry of every ``down`` row positive, so results are reproducible -
the singular values and the un-clamped product ``up @ down`` are what device parity is checked on.

CPU tensors take the reference's exact full-SVD path (plumbing).
"""
from __future__ import annotations

import sys
from typing import Optional, Tuple

import torch

from . import _C
from .lora import (LoraInjectedConv2d, LoraInjectedLinear, inject_trainable_lora, inject_trainable_lora_extended,
                   save_all)


def _iter_lora(model):  # ref :16-21
    for module in model.modules():
        if isinstance(module, (LoraInjectedConv2d, LoraInjectedLinear)):
            yield module


def _fix_signs(U: torch.Tensor, Vh: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    idx = Vh.abs().argmax(dim=1)
    sgn = torch.sign(Vh.gather(1, idx[:, None]).squeeze(1))
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
    return U * sgn[None, :], Vh * sgn[:, None]


def topr_svd(delta: torch.Tensor, rank: int, oversample: int = 8, n_iter: int = 4,
             generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Top-``rank`` singular triplets (U [N,r], S [r], Vh [r,K]) of a 2-D f32 matrix.

    Device tensors: randomized subspace iteration on the HIP primitives (see module docstring).
    CPU tensors: exact ``torch.linalg.svd`` (what the reference runs)."""
    N, K = delta.shape
    if not delta.is_cuda:
        U, S, Vh = torch.linalg.svd(delta.float(), full_matrices=False)
        return U[:, :rank], S[:rank], Vh[:rank]  # LAPACK's signs, as in the reference
    _C.require()
    delta = delta.float().contiguous()
    l = min(rank + oversample, N, K, _C.MAX_RANK)
    if l < rank:
        raise ValueError(f"rank {rank} exceeds what the device path supports for a {N}x{K} matrix")
    omega = torch.randn(l, K, device=delta.device, dtype=torch.float32, generator=generator)
    y = _C.rowdot(delta, omega, _C.FACTOR_RK)                       # [N, l] = dW @ Omega^T
    q, _ = torch.linalg.qr(y)
    for _ in range(n_iter):
        z = _C.colreduce(delta, q.contiguous(), _C.FACTOR_RK)       # [l, K] = Q^T dW
        qz, _ = torch.linalg.qr(z.t())                              # [K, l]
        y = _C.rowdot(delta, qz.t().contiguous(), _C.FACTOR_RK)     # [N, l] = dW @ Qz
        q, _ = torch.linalg.qr(y)
    b = _C.colreduce(delta, q.contiguous(), _C.FACTOR_RK)           # [l, K] = Q^T dW
    ub, s, vh = torch.linalg.svd(b, full_matrices=False)            # small: l x K
    u = q @ ub
    U, Vh = _fix_signs(u[:, :rank], vh[:rank])
    return U, s[:rank], Vh


def distill_pair(w_tuned: torch.Tensor, w_base: torch.Tensor, rank: int, clamp_quantile: float = 0.99,
                 generator: Optional[torch.Generator] = None, **svd_kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """(up [N, r], down [r, K]) of one site from its tuned / base weights flattened to 2-D (ref :30-47, :57-74)."""
    residual = (w_tuned.float() - w_base.float()).flatten(start_dim=1)
    U, S, Vh = topr_svd(residual, rank, generator=generator, **svd_kw)
    U = U @ torch.diag(S)
    dist = torch.cat([U.flatten(), Vh.flatten()])
    hi = torch.quantile(dist, clamp_quantile)
    return U.clamp(-hi, hi), Vh.clamp(-hi, hi)


def overwrite_base(base_model, tuned_model, rank, clamp_quantile, seed: int = 0, **svd_kw):
    """ref :24-92 — walk both models' adapters pairwise and overwrite the BASE model's LoRA factors with the rank-r
    distillation of (tuned - base) frozen weights."""
    gens = {}
    for lor_base, lor_tune in zip(_iter_lora(base_model), _iter_lora(tuned_model)):
        fb, ft = lor_base._frozen(), lor_tune._frozen()
        dev, dtype = fb.weight.device, fb.weight.dtype
        if dev.type == "cuda" and dev not in gens:
            gens[dev] = torch.Generator(device=dev).manual_seed(seed)
        print(("Distill Linear shape " if isinstance(lor_base, LoraInjectedLinear) else "Distill Conv shape "),
              tuple(fb.weight.shape))
        up, down = distill_pair(ft.weight.data, fb.weight.data, rank, clamp_quantile, gens.get(dev), **svd_kw)
        if isinstance(lor_base, LoraInjectedConv2d):
            up = up.reshape(up.shape[0], up.shape[1], 1, 1)
            down = down.reshape(down.shape[0], fb.in_channels, fb.kernel_size[0], fb.kernel_size[1])
        assert lor_base.lora_up.weight.shape == up.shape
        assert lor_base.lora_down.weight.shape == down.shape
        lor_base.lora_up.weight.data = up.to(device=dev, dtype=dtype)
        lor_base.lora_down.weight.data = down.to(device=dev, dtype=dtype)


def _load_pipe(path: str, device: str):
    """(unet, text_encoder) of a checkpoint; ``standin[:seed]`` builds random-init stand-ins (no diffusers here)."""
    if path.startswith("standin"):
        from .standin import clip_text_model, tiny_unet

        seed = int(path.split(":", 1)[1]) if ":" in path else 0
        torch.manual_seed(seed)
        return tiny_unet().to(device), clip_text_model(hidden=32, layers=2, heads=2).to(device)
    from diffusers import StableDiffusionPipeline

    pipe = StableDiffusionPipeline.from_pretrained(path, torch_dtype=torch.float16).to(device)
    return pipe.unet, pipe.text_encoder


def svd_distill(target_model: str, base_model: str, rank: int = 4, clamp_quantile: float = 0.99,
                device: str = "cuda:0", save_path: str = "svd_distill.safetensors"):
    """ref :95-142."""
    unet_b, te_b = _load_pipe(base_model, device)
    unet_t, te_t = _load_pipe(target_model, device)
    inject_trainable_lora_extended(unet_b, r=rank)
    inject_trainable_lora_extended(unet_t, r=rank)
    overwrite_base(unet_b, unet_t, rank=rank, clamp_quantile=clamp_quantile)
    inject_trainable_lora(te_b, r=rank, target_replace_module={"CLIPAttention"})
    inject_trainable_lora(te_t, r=rank, target_replace_module={"CLIPAttention"})
    overwrite_base(te_b, te_t, rank=rank, clamp_quantile=clamp_quantile)
    save_all(unet=unet_b, text_encoder=te_b, placeholder_token_ids=None, placeholder_tokens=None, save_path=save_path,
             save_lora=True, save_ti=False)


def main():
    from .cli_lora_pti import _parse_cli

    svd_distill(**_parse_cli(sys.argv[1:]))


if __name__ == "__main__":
    main()
