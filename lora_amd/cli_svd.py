"""Drop-in for the reference's ``lora_distill`` console script (``lora_diffusion/cli_svd.py``): distil the difference
between a fine-tuned model and its base into rank-r LoRA factors, site by site — BASELINE configs[4] / SURVEY §8f-1.

Reference recipe per site (cli_svd.py:30-53 Linear, :55-92 Conv2d flattened from dim 1): full ``torch.linalg.svd`` of
``dW = W_tuned - W_base`` in f32, keep the top r triplets, ``up = U_r diag(S_r)``, ``down = Vh_r``, clamp both at the
0.99-quantile of their joint value distribution.

MI355X path (device tensors): only r of min(N, K) singular triplets are wanted, so the full SVD (O(N K min(N,K)) flops,
one LAPACK-style call per site, 224 of them) is replaced by randomized subspace iteration (Halko-Martinsson-Tropp)
BATCHED over every same-shape site of the model (``distill_group``): the residuals of a group are stacked [B, N, K] and
each step of the iteration is ONE launch for the whole stack —

    Y  = dW Omega            ``lora_amd_colreduce_batched`` over the transposed stack   [B, N, l]
    Z  = dW^T Q              ``lora_amd_colreduce_batched`` over the stack              [B, K, l]   (l = 2r rounded up to 8)
    Q  = orth(Y)             shifted CholeskyQR3 on the device: Gram = Y^T Y (colreduce_batched of Y with itself),
                             L^{-1} of the small Gram (``lora_amd_chol_inverse_batched``, one wave per matrix),
                             Q = Y L^{-T} (rowdot_batched); shift on the first pass, two clean passes after it

— all HBM-streaming passes over dW with a skinny factor (``csrc/linear.hip``), plus one batched l x l SVD (torch) of
the triangular-factor-sized core at the end.  With ``n_iter`` power iterations the captured subspace error decays like
(s_{l+1}/s_r)^(2 n_iter + 1); the defaults reproduce ``up @ down`` of the reference to ~1e-4 relative on
distillation-like spectra.  The signs of singular vectors are arbitrary (LAPACK's are too, and the reference's clamp
threshold - a quantile of SIGNED entries - inherits that arbitrariness); the device path fixes them by making the
largest-magnitude entry of every ``down`` row positive, so results are reproducible.  Device parity
(tests/test_gpu_parity_r2.py) compares, after aligning each pair's sign to the reference's, the factors, the clamp
threshold and the clamped factors with the reference recipe.

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


def _orth(y: torch.Tensor) -> torch.Tensor:
    """Columns of every matrix of the stack ``y`` [B, M, l] orthonormalised on the device: shifted CholeskyQR3
    (Fukaya et al.): the first pass adds a shift to the Gram matrix so that the f32 Cholesky cannot break down on an
    ill-conditioned block (power iteration drives the columns towards each other), the two clean passes restore
    orthonormality to f32 precision.  9 small batched launches; no host round trip."""
    for shift in (1e-4, 0.0, 0.0):
        gram = _C.colreduce_batched(y, y, _C.FACTOR_RK)                  # [B, l, l] = Y^T Y
        linv = _C.chol_inverse_batched(gram, shift)                       # L^{-1}
        y = _C.rowdot_batched(y, linv, _C.FACTOR_RK)                      # Y L^{-T}
    return y


def _sketch_width(rank: int, oversample: int, N: int, K: int) -> int:
    """Sketch columns: rank + max(oversample, rank) rounded up to the kernels' 8-column granule, at most 32."""
    l = -(-(rank + max(oversample, rank)) // 8) * 8
    return min(l, 32)


def group_supported(N: int, K: int, rank: int, oversample: int = 8) -> bool:
    """Shapes the batched device path takes: 16-byte-friendly rows and a sketch that fits the small dense kernels."""
    l = _sketch_width(rank, oversample, N, K)
    return K % 8 == 0 and rank <= l <= min(N, K)


def topr_svd_batched(delta: torch.Tensor, rank: int, oversample: int = 8, n_iter: int = 4,
                     generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Top-``rank`` singular triplets (U [B,N,r], S [B,r], Vh [B,r,K]) of a stack of f32 matrices [B, N, K] on the
    device; every step is one launch for the whole stack (see module docstring)."""
    _C.require()
    B, N, K = delta.shape
    l = _sketch_width(rank, oversample, N, K)
    if not group_supported(N, K, rank, oversample):
        raise ValueError(f"topr_svd_batched: shape {N}x{K} rank {rank} is outside the batched device path")
    delta = delta.float().contiguous()
    # both products of the iteration as column-reduction passes (the faster primitive: ~2 TB/s on f32 rows, the row-dot
    # form re-stages its factor tile per few rows of a tall-and-wide weight): dW^T Q streams dW, dW Qz streams the
    # resident transposed stack (a second f32 layout of the residuals; 288 GB of HBM)
    delta_t = delta.transpose(1, 2).contiguous()
    omega = torch.randn(B, K, l, device=delta.device, dtype=torch.float32, generator=generator)
    q = _orth(_C.colreduce_batched(delta_t, omega, _C.FACTOR_KR))        # [B, N, l] = orth(dW Omega)
    for _ in range(n_iter):
        qz = _orth(_C.colreduce_batched(delta, q, _C.FACTOR_KR))         # [B, K, l] = orth(dW^T Q)
        q = _orth(_C.colreduce_batched(delta_t, qz, _C.FACTOR_KR))       # [B, N, l] = orth(dW Qz)
    b = _C.colreduce_batched(delta, q, _C.FACTOR_RK)                     # [B, l, K] = Q^T dW
    # small SVD of b [l, K]: b^T = Qb Rb with Qb = orth(b^T) (the same CholeskyQR3), SVD of the l x l factor Rb^T = b Qb
    # (batched, tiny), b = Ub S (Qb Vb)^T.  (A batched gesvd on [B, l, K] is 10-20x slower here; the Gram/eigh route
    # loses the orthonormality of the weak directions.)
    qb = _orth(b.transpose(1, 2).contiguous())                           # [B, K, l]
    ub, s, vbh = torch.linalg.svd(torch.bmm(b, qb), full_matrices=False)  # [B, l, l]
    u = torch.bmm(q, ub[:, :, :rank])
    vh = torch.bmm(vbh[:, :rank], qb.transpose(1, 2))
    s = s[:, :rank]
    idx = vh.abs().argmax(dim=2, keepdim=True)
    sgn = torch.sign(vh.gather(2, idx)).squeeze(2)
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
    return u * sgn[:, None, :], s, vh * sgn[:, :, None]


def topr_svd(delta: torch.Tensor, rank: int, oversample: int = 8, n_iter: int = 4,
             generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Top-``rank`` singular triplets (U [N,r], S [r], Vh [r,K]) of a 2-D f32 matrix.

    Device tensors: the batched randomized path with a stack of one (shapes it does not cover: exact SVD on the
    device).  CPU tensors: exact ``torch.linalg.svd`` (what the reference runs)."""
    N, K = delta.shape
    if not delta.is_cuda or not group_supported(N, K, rank, oversample):
        U, S, Vh = torch.linalg.svd(delta.float(), full_matrices=False)
        U, S, Vh = U[:, :rank], S[:rank], Vh[:rank]
        if delta.is_cuda:  # same deterministic sign rule as the batched path
            U, Vh = _fix_signs(U, Vh)
        return U, S, Vh  # CPU: LAPACK's signs, as in the reference
    U, S, Vh = topr_svd_batched(delta[None], rank, oversample, n_iter, generator)
    return U[0], S[0], Vh[0]


def _clamp_pairs(U: torch.Tensor, Vh: torch.Tensor, clamp_quantile: float):
    """ref :39-47 for a stack: per site, clamp both factors at the quantile of their joint (signed) values."""
    dist = torch.cat([U.flatten(1), Vh.flatten(1)], dim=1)
    hi = torch.quantile(dist, clamp_quantile, dim=1)
    return torch.minimum(torch.maximum(U, -hi[:, None, None]), hi[:, None, None]), \
        torch.minimum(torch.maximum(Vh, -hi[:, None, None]), hi[:, None, None])


def distill_group(w_tuned, w_base, rank: int, clamp_quantile: float = 0.99,
                  generator: Optional[torch.Generator] = None, **svd_kw):
    """(up [B, N, r], down [B, r, K]) of a group of same-shape sites: lists of tuned / base weights (any trailing
    dims are flattened from dim 1, ref :57-60).  Device tensors of supported shapes: ONE batched iteration."""
    res = torch.stack([(t.float() - b.float()).flatten(start_dim=1) for t, b in zip(w_tuned, w_base)])
    B, N, K = res.shape
    if res.is_cuda and group_supported(N, K, rank, svd_kw.get("oversample", 8)):
        U, S, Vh = topr_svd_batched(res, rank, generator=generator, **svd_kw)
    else:
        trip = [topr_svd(res[i], rank, generator=generator, **svd_kw) for i in range(B)]
        U, S, Vh = (torch.stack([t[k] for t in trip]) for k in range(3))
    return _clamp_pairs(U * S[:, None, :], Vh, clamp_quantile)


def distill_pair(w_tuned: torch.Tensor, w_base: torch.Tensor, rank: int, clamp_quantile: float = 0.99,
                 generator: Optional[torch.Generator] = None, **svd_kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """(up [N, r], down [r, K]) of one site from its tuned / base weights flattened to 2-D (ref :30-47, :57-74)."""
    up, down = distill_group([w_tuned], [w_base], rank, clamp_quantile, generator, **svd_kw)
    return up[0], down[0]


def overwrite_base(base_model, tuned_model, rank, clamp_quantile, seed: int = 0, **svd_kw):
    """ref :24-92 — walk both models' adapters pairwise and overwrite the BASE model's LoRA factors with the rank-r
    distillation of (tuned - base) frozen weights.  Sites are grouped by (kind, weight shape): one batched device
    iteration per group (SD1.5 extended: 224 sites in ~20 groups) instead of 224 LAPACK calls."""
    groups, gens = {}, {}
    for lor_base, lor_tune in zip(_iter_lora(base_model), _iter_lora(tuned_model)):
        fb, ft = lor_base._frozen(), lor_tune._frozen()
        print(("Distill Linear shape " if isinstance(lor_base, LoraInjectedLinear) else "Distill Conv shape "),
              tuple(fb.weight.shape))
        key = (type(lor_base).__name__, tuple(fb.weight.shape), fb.weight.device, fb.weight.dtype)
        groups.setdefault(key, []).append((lor_base, fb, ft))
    for (kind, shape, dev, dtype), sites in groups.items():
        if dev.type == "cuda" and dev not in gens:
            gens[dev] = torch.Generator(device=dev).manual_seed(seed)
        ups, downs = distill_group([ft.weight.data for _, _, ft in sites], [fb.weight.data for _, fb, _ in sites],
                                   rank, clamp_quantile, gens.get(dev), **svd_kw)
        for (lor_base, fb, _), up, down in zip(sites, ups, downs):
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
