"""Plain-PyTorch (CPU, fp32) restatement of the reference adapters and training step.
TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.

The reference executes its hot path as stock ATen ops; this file restates that op sequence
functionally (one launch per reference op, nothing fused) so that (a) whole-model parity of the
HIP path can be checked and (b) ``bench.py`` can time the reference's algorithm on the host cores
of the GPU box, where ``/root/reference`` does not exist.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def linear_adapter_forward(x, W, b, down, up, scale, selector=None, p=0.0, training=False, mask=None):
    """ref: lora_diffusion/lora.py:53-58 — addmm, mm, (selector mm), mm, dropout, mul, add.
    ``mask``: nn.Dropout's random draw GIVEN (the multiplier tensor 0 | 1/(1-p), broadcastable to the branch) instead of
    drawn — the way pti_loss_step below takes its noise and timesteps; then ``p`` / ``training`` are not consulted."""
    t = F.linear(x, down)
    if selector is not None:
        t = F.linear(t, selector)
    branch = F.linear(t, up)
    branch = branch * mask.reshape(branch.shape) if mask is not None else F.dropout(branch, p, training)
    return F.linear(x, W, b) + branch * scale


def conv_adapter_forward(x, W, b, down, up, scale, stride, padding, dilation, groups, selector=None, p=0.0,
                         training=False, mask=None):
    """ref: lora_diffusion/lora.py:130-135 — conv2d, conv2d (k x k -> r), conv2d 1x1, dropout, mul, add.
    ``mask``: as in linear_adapter_forward ([B, C_out, H, W])."""
    t = F.conv2d(x, down, None, stride, padding, dilation, groups)
    if selector is not None:
        t = F.conv2d(t, selector)
    branch = F.conv2d(t, up)
    branch = branch * mask if mask is not None else F.dropout(branch, p, training)
    return F.conv2d(x, W, b, stride, padding, dilation, groups) + branch * scale


class RefLinearSite(nn.Module):
    """Holds the frozen Linear (aliased, ref:290-292) + fresh factors (init ref:50-51)."""

    def __init__(self, frozen: nn.Linear, r: int, dropout_p: float, scale: float):
        super().__init__()
        if r > min(frozen.in_features, frozen.out_features):  # ref:38-41
            raise ValueError(f"LoRA rank {r} must be less or equal than {min(frozen.in_features, frozen.out_features)}")
        self.frozen, self.r, self.p, self.scale = frozen, r, dropout_p, scale
        self.down = nn.Parameter(torch.randn(r, frozen.in_features) / r)
        self.up = nn.Parameter(torch.zeros(frozen.out_features, r))
        self.mask = None  # a given dropout draw for the next forward (tests: the device's mask)

    def forward(self, x):
        return linear_adapter_forward(x, self.frozen.weight, self.frozen.bias, self.down, self.up, self.scale,
                                      None, self.p, self.training, self.mask)


class RefConvSite(nn.Module):
    def __init__(self, frozen: nn.Conv2d, r: int, dropout_p: float, scale: float):
        super().__init__()
        if r > min(frozen.in_channels, frozen.out_channels):  # ref:89-92
            raise ValueError(f"LoRA rank {r} must be less or equal than {min(frozen.in_channels, frozen.out_channels)}")
        self.frozen, self.r, self.p, self.scale = frozen, r, dropout_p, scale
        kh, kw = frozen.kernel_size
        self.down = nn.Parameter(torch.randn(r, frozen.in_channels // frozen.groups, kh, kw) / r)
        self.up = nn.Parameter(torch.zeros(frozen.out_channels, r, 1, 1))
        self.mask = None

    def forward(self, x):
        f = self.frozen
        return conv_adapter_forward(x, f.weight, f.bias, self.down, self.up, self.scale, f.stride, f.padding,
                                    f.dilation, f.groups, None, self.p, self.training, self.mask)


def _sites(model: nn.Module, targets: Iterable[str], kinds: Tuple[type, ...]):
    """ref: lora_diffusion/lora.py:207-232 — (parent, name, module) in the reference's yield order."""
    targets = set(targets)
    for anc in (m for m in model.modules() if type(m).__name__ in targets):
        for dotted, mod in anc.named_modules():
            if isinstance(mod, kinds):
                head, _, leaf = dotted.rpartition(".")
                parent = anc.get_submodule(head) if head else anc
                if isinstance(parent, (RefLinearSite, RefConvSite)):
                    continue
                yield parent, leaf, mod


def inject(model: nn.Module, targets: Iterable[str], r: int = 4, dropout_p: float = 0.0, scale: float = 1.0,
           conv: bool = False) -> List[nn.Parameter]:
    """ref: lora_diffusion/lora.py:255-309 / :312-380 — returns the trainable [up, down, up, down, ...]."""
    kinds = (nn.Linear, nn.Conv2d) if conv else (nn.Linear,)
    out = []
    for parent, name, mod in _sites(model, targets, kinds):
        site = RefLinearSite(mod, r, dropout_p, scale) if isinstance(mod, nn.Linear) else RefConvSite(mod, r, dropout_p, scale)
        site.to(mod.weight.device)
        parent._modules[name] = site
        out += [site.up, site.down]
    return out


def sites_of(model: nn.Module):
    return [m for m in model.modules() if isinstance(m, (RefLinearSite, RefConvSite))]


def collapse(W: torch.Tensor, up: torch.Tensor, down: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """``collapse_lora`` for one Linear site (lora_diffusion/lora.py:646-655): materialise ``up @ down`` in the factor
    dtype, cast to W's dtype, scale, add — three passes over an [N, K] tensor plus the new Parameter."""
    return ((up @ down).type(W.dtype) * alpha + W).to(W.dtype)


def dreambooth_step(unet, params, optimizer, latents, noise, timesteps, ehs, alphas_cumprod, max_grad_norm=1.0,
                    with_prior_preservation=False, prior_loss_weight=1.0):
    """ref: training_scripts/train_lora_dreambooth.py:824-888 with VAE/text-encoder outputs given:
    add_noise -> unet -> MSE (+prior) -> backward -> clip_grad_norm_ -> AdamW.step -> zero_grad."""
    a = alphas_cumprod[timesteps].to(latents.dtype)
    noisy = a.sqrt().view(-1, 1, 1, 1) * latents + (1 - a).sqrt().view(-1, 1, 1, 1) * noise
    pred = unet(noisy, timesteps, ehs)
    if with_prior_preservation:
        pred, pred_prior = torch.chunk(pred, 2, dim=0)
        target, target_prior = torch.chunk(noise, 2, dim=0)
        loss = F.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
        loss = loss + prior_loss_weight * F.mse_loss(pred_prior.float(), target_prior.float(), reduction="mean")
    else:
        loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()


def pti_loss_step(unet, latents, noise, timesteps, ehs, alphas_cumprod, mask=None, mask_temperature=1.0,
                  prediction_type="epsilon"):
    """ref: lora_diffusion/cli_lora_pti.py:260-370 (``loss_step``) with the random draws (noise :296, timesteps
    :299-305) and the text-encoder output (:317-319 / :327-329) given, VAE skipped as under ``cached_latents``
    (:288-289): add_noise (:307) -> unet (:331) -> target by prediction type (:333-338) -> optional mask: reshape to
    the image grid (:342-349), nearest resize to the prediction's grid (:351-355), ``(mask + 0.01) ** temperature``
    (:357), divide by its max (:359), multiply prediction AND target (:361-363) -> per-sample MSE, mean over the
    batch (:365-369).  ``unet(noisy, t, ehs)`` returns the prediction tensor."""
    a = alphas_cumprod[timesteps].to(latents.dtype)
    sa, sb = a.sqrt().view(-1, 1, 1, 1), (1 - a).sqrt().view(-1, 1, 1, 1)
    noisy = sa * latents + sb * noise
    pred = unet(noisy, timesteps, ehs)
    if prediction_type == "epsilon":
        target = noise
    elif prediction_type == "v_prediction":
        target = sa * noise - sb * latents
    else:
        raise ValueError(f"Unknown prediction type {prediction_type}")
    if mask is not None:
        m = mask.to(pred.device).reshape(pred.shape[0], 1, pred.shape[2] * 8, pred.shape[3] * 8)
        m = F.interpolate(m.float(), size=pred.shape[-2:], mode="nearest")
        m = (m + 0.01).pow(mask_temperature)
        m = m / m.max()
        pred, target = pred * m, target * m
    return F.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()
