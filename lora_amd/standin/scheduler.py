"""DDPM forward-noising of SD1.5 (scaled-linear betas 0.00085..0.012, 1000 steps).

Stand-in for ``diffusers.DDPMScheduler`` (un-vendored; ref call sites:
training_scripts/train_lora_dreambooth.py:678-680, 837; lora_diffusion/cli_lora_pti.py:305)."""
from __future__ import annotations

import types

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 prediction_type: str = "epsilon"):
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                            beta_end=beta_end, beta_schedule="scaled_linear",
                                            prediction_type=prediction_type)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(torch.float32)
        self._dev_tables = {}

    def _table(self, device) -> torch.Tensor:
        # resident per device: no host->device copy inside the step (illegal during hipGraph capture)
        t = self._dev_tables.get(device)
        if t is None:
            t = self._dev_tables[device] = self.alphas_cumprod.to(device=device)
        return t

    def _coef(self, like: torch.Tensor, timesteps: torch.Tensor):
        a = self._table(like.device)[timesteps].to(like.dtype)
        shape = (-1,) + (1,) * (like.dim() - 1)
        return a.sqrt().view(shape), (1 - a).sqrt().view(shape)

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        sa, sb = self._coef(original, timesteps)
        return sa * original + sb * noise

    def get_velocity(self, sample: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        sa, sb = self._coef(sample, timesteps)
        return sa * noise - sb * sample
