"""Drop-in for the reference's ``lora_pti`` console script (``lora_diffusion/cli_lora_pti.py``): pivotal tuning =
textual inversion of placeholder tokens, then LoRA tuning of the UNet (+ CLIP text encoder), optionally with the
extended (ResnetBlock2D Conv2d) adapters — BASELINE configs[3].

``train(**kwargs)`` keeps the reference's keyword surface (ref :696-752) and outputs (``step_inv_{n}.safetensors``,
``step_{n}.safetensors``, ``{out_name}.safetensors`` via ``save_all``).  MI355X-first differences:

* phase 2 (ref ``perform_tuning`` :545-693) runs the adapters on the HIP kernels, keeps the frozen weights resident
  in bf16 (the reference autocasts fp32 weights every step, :315-324), and replaces ``clip_grad_norm_`` over every
  UNet + CLIP parameter plus per-tensor AdamW by the fused flat-buffer clip + AdamW;
* phase 1 (ref ``train_inversion`` :373-542) updates ONLY the placeholder rows of the token-embedding matrix (a
  ``[n_tokens, hidden]`` AdamW state) instead of running AdamW over the whole 49k x 768 table and copying all other
  rows back afterwards (:477-479) — same result, ~38 M fewer parameters touched per step;
* it may be launched with one process per GPU (``python -m torch.distributed.run -m lora_amd.cli_lora_pti ...``): the
  image set is sharded by rank and the flat LoRA gradient (and the placeholder-row gradient) is all-reduced over RCCL.
  The reference CLI is single-device only (``device="cuda:0"``, :743).

Not supported without the real packages: ``train_inpainting`` (needs the 9-channel inpainting UNet),
``use_face_segmentation_condition`` (mediapipe), ``log_wandb``.
"""
from __future__ import annotations

import itertools
import math
import os
import re
import sys
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import trainer as T
from .lora import (UNET_EXTENDED_TARGET_REPLACE, inject_trainable_lora, inject_trainable_lora_extended, inspect_lora,
                   save_all)
from .standin import io as SIO


def get_models(pretrained_model_name_or_path, pretrained_vae_name_or_path, revision, placeholder_tokens: List[str],
               initializer_tokens: List[str], device="cuda:0", standin: str = "sd15", seed: Optional[int] = None):
    """ref :49-128 — load (or stand in for) the models, add the placeholder tokens and initialise their embeddings
    (``<rand-sigma>``, ``<zero>`` or a copy of a single existing token)."""
    tokenizer, text_encoder, vae, unet, scheduler, what = SIO.load_host_models(
        pretrained_model_name_or_path, pretrained_vae_name_or_path, revision, None, torch.device(device), standin, seed)
    print("PTI : models:", what)
    ids = []
    for token, init_tok in zip(placeholder_tokens, initializer_tokens):
        if tokenizer.add_tokens(token) == 0:
            raise ValueError(f"The tokenizer already contains the token {token}. Please pass a different"
                             " `placeholder_token` that is not already in the tokenizer.")
        tid = tokenizer.convert_tokens_to_ids(token)
        ids.append(tid)
        text_encoder.resize_token_embeddings(len(tokenizer))
        emb = text_encoder.get_input_embeddings().weight.data
        if init_tok.startswith("<rand"):
            sigma = float(re.findall(r"<rand-(.*)>", init_tok)[0])
            emb[tid] = torch.randn_like(emb[0]) * sigma
            print(f"Initialized {token} with random noise (sigma={sigma}); norm {emb[tid].norm():.4f}")
        elif init_tok == "<zero>":
            emb[tid] = torch.zeros_like(emb[0])
        else:
            tids = tokenizer.encode(init_tok, add_special_tokens=False)
            if len(tids) > 1:
                raise ValueError("The initializer token must be a single token.")
            emb[tid] = emb[tids[0]]
    return text_encoder.to(device), vae.to(device), unet.to(device), tokenizer, ids, scheduler


def text2img_dataloader(dataset, batch_size, tokenizer, vae, cached_latents: bool, device, rank: int = 0, world: int = 1,
                        seed: int = 0):
    """ref :131-195 — with ``cached_latents`` the VAE runs ONCE over the (rank's share of the) image set and the loader
    yields latents; afterwards the VAE is not needed any more."""
    if cached_latents:
        cached = []
        with torch.no_grad():
            for i in range(rank, len(dataset), world):
                ex = dataset[i]
                ex["instance_images"] = (vae.encode(ex["instance_images"][None].to(device)).latent_dist.sample()[0]
                                         * 0.18215).cpu()
                cached.append(ex)
        if not cached:
            raise ValueError(f"rank {rank} of {world} got no images: use at least {world} images")
        source, sampler = cached, None
    else:
        source = dataset
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True, seed=seed) \
            if world > 1 else None

    def collate(examples):
        ids = [e["instance_prompt_ids"] for e in examples]
        ids = tokenizer.pad({"input_ids": ids}, padding="max_length", max_length=tokenizer.model_max_length,
                            return_tensors="pt").input_ids
        batch = {"input_ids": ids, "pixel_values": torch.stack([e["instance_images"] for e in examples]).contiguous()}
        if examples[0].get("mask") is not None:
            batch["mask"] = torch.stack([e["mask"] for e in examples])
        return batch

    return torch.utils.data.DataLoader(source, batch_size=batch_size, shuffle=sampler is None, sampler=sampler,
                                       collate_fn=collate)


def loss_step(batch, unet, vae, text_encoder, scheduler, train_inpainting=False, t_mutliplier=1.0, mixed_precision=False,
              mask_temperature=1.0, cached_latents: bool = False):
    """ref :260-370 — noise, DDPM forward process with timesteps < 1000 * t_mutliplier, text encoder, UNet, (masked)
    per-sample MSE.  ``mixed_precision`` needs no autocast here: the frozen weights are already resident in the
    compute dtype and the adapters read the f32 LoRA masters directly."""
    if train_inpainting:
        raise NotImplementedError("train_inpainting needs the 9-channel inpainting UNet of a real checkpoint")
    dev, dt = unet.device, unet.dtype
    if cached_latents:
        latents = batch["pixel_values"].to(dev, dt)
    else:
        with torch.no_grad():
            latents = (vae.encode(batch["pixel_values"].to(dev)).latent_dist.sample() * 0.18215).to(dt)
    noise = torch.randn_like(latents)
    timesteps = torch.randint(0, int(scheduler.config.num_train_timesteps * t_mutliplier), (latents.shape[0],),
                              device=dev).long()
    # formed in f32, rounded once (trainer.forward_backward says why: a 16-bit alpha_bar_t rounds to 1 at t = 0)
    noisy = scheduler.add_noise(latents.float(), noise.float(), timesteps).to(dt)
    ehs = text_encoder(batch["input_ids"].to(dev))[0]
    pred = unet(noisy, timesteps, ehs.to(dt)).sample
    ptype = getattr(scheduler.config, "prediction_type", "epsilon")
    if ptype == "epsilon":
        target = noise
    elif ptype == "v_prediction":
        target = scheduler.get_velocity(latents.float(), noise.float(), timesteps)
    else:
        raise ValueError(f"Unknown prediction type {ptype}")
    if batch.get("mask", None) is not None:
        mask = batch["mask"].to(dev).reshape(pred.shape[0], 1, pred.shape[2] * 8, pred.shape[3] * 8)
        mask = F.interpolate(mask.float(), size=pred.shape[-2:], mode="nearest")
        mask = (mask + 0.01).pow(mask_temperature)
        mask = mask / mask.max()
        pred, target = pred * mask, target * mask
    return F.mse_loss(pred.float(), target.float(), reduction="none").mean([1, 2, 3]).mean()


class PlaceholderRows:
    """AdamW + norm decay over the placeholder rows of the token-embedding table only (see module docstring).  On the
    device the whole update (gather the rows' gradient, AdamW on f32 master rows, decay toward norm 0.4, scatter back)
    is ONE launch of ``lora_amd_ti_rows_step``; on CPU the same maths in torch."""

    def __init__(self, text_encoder, token_ids: List[int], lr: float, weight_decay: float):
        self.emb = text_encoder.get_input_embeddings().weight
        self.ids = torch.tensor(token_ids, dtype=torch.long, device=self.emb.device)
        self.rows = self.emb.data[self.ids].float().clone().contiguous()
        self.m, self.v = torch.zeros_like(self.rows), torch.zeros_like(self.rows)
        self.weight_decay, self.t = float(weight_decay), 0

    def step(self, lr: float, world: int = 1, clip_ti_decay: bool = False, inv_scale=None):
        """One optimiser step on the rows from ``emb.grad`` (averaged over ``world`` ranks), optional norm decay with
        lambda = min(1, 100 lr) (ref :451-469), rows written back; every other row of the table is untouched."""
        self.t += 1
        lam = min(1.0, 100 * lr) if clip_ti_decay and len(self.ids) else -1.0
        grad = self.emb.grad
        if inv_scale is not None:  # fp16 loss scaling: the table gradient carries the scale of this backward (the
            # caller skips the whole step when the batch overflowed: PlaceholderRows.skip)
            grad = torch.nan_to_num(grad.float() * inv_scale, nan=0.0, posinf=0.0, neginf=0.0).to(grad.dtype)
            self.emb.grad = grad
        if world > 1:
            g = grad[self.ids].contiguous()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            grad[self.ids] = g
        if self.emb.is_cuda:
            from . import _C

            _C.ti_rows_step(self.emb.data, grad.contiguous(), self.ids, self.rows, self.m, self.v, lr, self.t,
                            weight_decay=self.weight_decay, grad_scale=1.0 / world, decay_lambda=lam)
        else:
            g = grad[self.ids].float() / world
            b1, b2, eps = 0.9, 0.999, 1e-8
            self.rows.mul_(1 - lr * self.weight_decay)
            self.m.lerp_(g, 1 - b1)
            self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = self.v.sqrt() / math.sqrt(1 - b2 ** self.t) + eps
            self.rows.addcdiv_(self.m, denom, value=-lr / (1 - b1 ** self.t))
            if lam >= 0:
                pre = self.rows.norm(dim=-1, keepdim=True)
                self.rows.copy_(F.normalize(self.rows, dim=-1) * (pre + lam * (0.4 - pre)))
            self.emb.data[self.ids] = self.rows.to(self.emb.dtype)
        self.emb.grad = None
        return self.rows.norm(dim=-1)

    def skip(self) -> None:
        """An overflowed fp16 batch: drop the table gradient, leave rows, moments and the step count untouched."""
        self.emb.grad = None


def _freeze_all_but_token_embedding(text_encoder):
    # ref :879-885 freezes encoder, final_layer_norm and position embeddings through ``.text_model`` (absent in recent
    # transformers); by name it is the same set
    for name, p in text_encoder.named_parameters():
        p.requires_grad = "token_embedding" in name


def train_inversion(unet, vae, text_encoder, dataloader, num_steps: int, scheduler, rows: PlaceholderRows, base_lr: float,
                    lr_lambda, save_steps: int, placeholder_token_ids, placeholder_tokens, save_path: str,
                    cached_latents: bool, accum_iter: int = 1, clip_ti_decay: bool = True, world: int = 1,
                    is_main: bool = True):
    """ref :373-542."""
    global_step = 0
    for _ in range(math.ceil(num_steps / max(1, len(dataloader)))):
        unet.eval()
        text_encoder.train()
        for batch in dataloader:
            lr = base_lr * lr_lambda(global_step + 1)  # the reference steps its scheduler before the optimiser
            loss = loss_step(batch, unet, vae, text_encoder, scheduler, cached_latents=cached_latents) / accum_iter
            loss.backward()
            if global_step % accum_iter == 0:  # (sic) ref :433 — also fires on the very first micro-batch
                norm = rows.step(lr, world, clip_ti_decay)
                if is_main and global_step % (10 * accum_iter) == 0:
                    print(f"TI step {global_step} loss {loss.item() * accum_iter:.5f} lr {lr:.3e} norm {norm.tolist()}")
            global_step += 1
            if global_step % save_steps == 0 and is_main:
                save_all(unet=unet, text_encoder=text_encoder, placeholder_token_ids=placeholder_token_ids,
                         placeholder_tokens=placeholder_tokens,
                         save_path=os.path.join(save_path, f"step_inv_{global_step}.safetensors"), save_lora=False)
            if global_step >= num_steps:
                return


def perform_tuning(unet, vae, text_encoder, dataloader, num_steps, scheduler, state: T.FlatLoraState, base_lrs,
                   lr_lambda, save_steps: int, placeholder_token_ids, placeholder_tokens, save_path,
                   lora_unet_target_modules, lora_clip_target_modules, mask_temperature, out_name: str,
                   cached_latents: bool, rows: Optional[PlaceholderRows] = None, rows_lr: float = 0.0, world: int = 1,
                   is_main: bool = True):
    """ref :545-693 — timesteps < 800, clip(1.0) over the trainable set, AdamW, periodic ``step_{n}.safetensors``."""
    global_step = 0
    unet.train()
    text_encoder.train()
    done = False
    for _ in range(math.ceil(num_steps / max(1, len(dataloader)))):
        for batch in dataloader:
            mult = lr_lambda(global_step + 1)
            state.set_lrs([b * mult for b in base_lrs])
            loss = loss_step(batch, unet, vae, text_encoder, scheduler, t_mutliplier=0.8, mixed_precision=True,
                             mask_temperature=mask_temperature, cached_latents=cached_latents)
            (loss * state.loss_scale if state.loss_scale is not None else loss).backward()
            state.step(state.all_reduce())
            applied = True
            if rows is not None:  # continue_inversion
                # GradScaler.step skips EVERY parameter of the optimiser on an overflowed batch (the placeholder rows
                # sit in the same AdamW as the LoRA factors, ref :960-997): no moment update, no weight decay
                applied = state.scaler is None or bool(state.scaler[3].item() > 0)
                if applied:
                    rows.step(rows_lr * mult, world, False, inv_scale=state.scaler[2] if state.scaler is not None else None)
                else:
                    rows.skip()
            global_step += 1
            if is_main and global_step % 10 == 0:
                print(f"tuning step {global_step}/{num_steps} loss {loss.item():.5f} lr {state.lrs[0]:.3e}")
            if global_step % save_steps == 0 and is_main:
                save_all(unet, text_encoder, placeholder_token_ids=placeholder_token_ids,
                         placeholder_tokens=placeholder_tokens,
                         save_path=os.path.join(save_path, f"step_{global_step}.safetensors"),
                         target_replace_module_text=lora_clip_target_modules,
                         target_replace_module_unet=lora_unet_target_modules)
                for nm, model in (("Unet", unet), ("CLIP", text_encoder)):
                    vals = list(itertools.chain(*inspect_lora(model).values()))
                    if vals:
                        print(f"LORA {nm} Moved", sum(vals) / len(vals))
            if global_step >= num_steps:
                done = True
                break
        if done:
            break
    if is_main:
        save_all(unet, text_encoder, placeholder_token_ids=placeholder_token_ids, placeholder_tokens=placeholder_tokens,
                 save_path=os.path.join(save_path, f"{out_name}.safetensors"),
                 target_replace_module_text=lora_clip_target_modules,
                 target_replace_module_unet=lora_unet_target_modules)


def train(instance_data_dir: str, pretrained_model_name_or_path: str, output_dir: str, train_text_encoder: bool = True,
          pretrained_vae_name_or_path: str = None, revision: Optional[str] = None, perform_inversion: bool = True,
          use_template=None, train_inpainting: bool = False, placeholder_tokens: str = "",
          placeholder_token_at_data: Optional[str] = None, initializer_tokens: Optional[str] = None, seed: int = 42,
          resolution: int = 512, color_jitter: bool = True, train_batch_size: int = 1, sample_batch_size: int = 1,
          max_train_steps_tuning: int = 1000, max_train_steps_ti: int = 1000, save_steps: int = 100,
          gradient_accumulation_steps: int = 4, gradient_checkpointing: bool = False, lora_rank: int = 4,
          lora_unet_target_modules={"CrossAttention", "Attention", "GEGLU"}, lora_clip_target_modules={"CLIPAttention"},
          lora_dropout_p: float = 0.0, lora_scale: float = 1.0, use_extended_lora: bool = False,
          clip_ti_decay: bool = True, learning_rate_unet: float = 1e-4, learning_rate_text: float = 1e-5,
          learning_rate_ti: float = 5e-4, continue_inversion: bool = False, continue_inversion_lr: Optional[float] = None,
          use_face_segmentation_condition: bool = False, cached_latents: bool = True,
          use_mask_captioned_data: bool = False, mask_temperature: float = 1.0, scale_lr: bool = False,
          lr_scheduler: str = "linear", lr_warmup_steps: int = 0, lr_scheduler_lora: str = "linear",
          lr_warmup_steps_lora: int = 0, weight_decay_ti: float = 0.00, weight_decay_lora: float = 0.001,
          use_8bit_adam: bool = False, device="cuda:0", extra_args: Optional[dict] = None, log_wandb: bool = False,
          wandb_log_prompt_cnt: int = 10, wandb_project_name: str = "new_pti_project",
          wandb_entity: str = "new_pti_entity", proxy_token: str = "person",
          enable_xformers_memory_efficient_attention: bool = False, out_name: str = "final_lora",
          standin: str = "sd15", mixed_precision: str = "bf16"):
    """ref :696-1036.  ``standin`` / ``mixed_precision`` are additions (stand-in model size; resident compute dtype)."""
    torch.manual_seed(seed)
    if log_wandb:
        raise NotImplementedError("log_wandb: wandb / the CLIP evaluation models are not available offline")
    if use_face_segmentation_condition:
        raise NotImplementedError("use_face_segmentation_condition needs mediapipe")
    dev = torch.device(device)
    rank, local, world = T.init_distributed(dev.type)
    if dev.type == "cuda" and world > 1:
        dev = torch.device("cuda", local)
    is_main = rank == 0
    if output_dir is not None and is_main:
        os.makedirs(output_dir, exist_ok=True)
    if len(placeholder_tokens) == 0:
        placeholder_tokens = []
        print("PTI : Placeholder Tokens not given, using null token")
    else:
        placeholder_tokens = placeholder_tokens.split("|")
        assert sorted(placeholder_tokens) == placeholder_tokens, \
            f"Placeholder tokens should be sorted. Use something like {'|'.join(sorted(placeholder_tokens))}'"
    if initializer_tokens is None:
        print("PTI : Initializer Tokens not given, doing random inits")
        initializer_tokens = ["<rand-0.017>"] * len(placeholder_tokens)
    else:
        initializer_tokens = initializer_tokens.split("|")
    assert len(initializer_tokens) == len(placeholder_tokens), "Unequal Initializer token for Placeholder tokens."
    if placeholder_token_at_data is not None:
        tok, pat = placeholder_token_at_data.split("|")
        token_map = {tok: pat}
    else:
        token_map = {"DUMMY": "".join(placeholder_tokens)}
    print("PTI : Placeholder Tokens", placeholder_tokens)
    print("PTI : Initializer Tokens", initializer_tokens)

    text_encoder, vae, unet, tokenizer, placeholder_token_ids, noise_scheduler = get_models(
        pretrained_model_name_or_path, pretrained_vae_name_or_path, revision, placeholder_tokens, initializer_tokens,
        device=str(dev), standin=standin, seed=seed)
    if gradient_checkpointing:
        unet.enable_gradient_checkpointing()
    mult = gradient_accumulation_steps * train_batch_size if scale_lr else 1
    unet_lr, text_encoder_lr, ti_lr = learning_rate_unet * mult, learning_rate_text * mult, learning_rate_ti * mult

    dataset = SIO.PivotalTuningDataset(instance_data_dir, tokenizer, token_map, use_template, resolution,
                                       use_mask_captioned_data=use_mask_captioned_data, seed=seed * 1000 + rank,
                                       content_seed=seed * 1000)
    if color_jitter and is_main:
        print("PTI : color_jitter needs torchvision (not installed); ignored")
    dataloader = text2img_dataloader(dataset, train_batch_size, tokenizer, vae, cached_latents, dev, rank, world, seed)

    unet.requires_grad_(False)
    vae.requires_grad_(False)
    _freeze_all_but_token_embedding(text_encoder)
    if cached_latents:
        vae = None

    # STEP 1 : inversion (f32 models, as in the reference: mixed_precision=False, ref :923)
    if perform_inversion and placeholder_token_ids:
        rows = PlaceholderRows(text_encoder, placeholder_token_ids, ti_lr, weight_decay_ti)
        train_inversion(unet, vae, text_encoder, dataloader, max_train_steps_ti, noise_scheduler, rows, ti_lr,
                        T.get_lr_lambda(lr_scheduler, lr_warmup_steps, max_train_steps_ti, lr_init=ti_lr), save_steps,
                        placeholder_token_ids, placeholder_tokens, output_dir, cached_latents,
                        accum_iter=gradient_accumulation_steps, clip_ti_decay=clip_ti_decay, world=world,
                        is_main=is_main)
        del rows

    # STEP 2 : LoRA tuning
    wdt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(mixed_precision, torch.float32)
    if dev.type == "cpu":
        wdt = torch.float32
    unet.to(wdt)
    if not use_extended_lora:
        unet_lora_params, _ = inject_trainable_lora(unet, r=lora_rank, target_replace_module=lora_unet_target_modules,
                                                    dropout_p=lora_dropout_p, scale=lora_scale)
    else:
        print("PTI : USING EXTENDED UNET!!!")
        lora_unet_target_modules = set(lora_unet_target_modules) | UNET_EXTENDED_TARGET_REPLACE
        print("PTI : Will replace modules: ", lora_unet_target_modules)
        unet_lora_params, _ = inject_trainable_lora_extended(unet, r=lora_rank,
                                                             target_replace_module=lora_unet_target_modules)
    print(f"PTI : has {len(unet_lora_params)} lora")
    T.promote_lora_to_fp32(unet)
    groups = [{"params": T.lora_params(unet), "lr": unet_lr, "weight_decay": weight_decay_lora}]
    text_encoder.requires_grad_(False)
    rows, rows_lr = None, 0.0
    if continue_inversion and placeholder_token_ids:
        _freeze_all_but_token_embedding(text_encoder)
        rows_lr = continue_inversion_lr if continue_inversion_lr is not None else ti_lr
        rows = PlaceholderRows(text_encoder, placeholder_token_ids, rows_lr, weight_decay_lora)
    if train_text_encoder:
        inject_trainable_lora(text_encoder, target_replace_module=lora_clip_target_modules, r=lora_rank)
        T.promote_lora_to_fp32(text_encoder)
        groups.append({"params": T.lora_params(text_encoder), "lr": text_encoder_lr, "weight_decay": weight_decay_lora})
    state = T.FlatLoraState(groups, max_grad_norm=1.0, device=dev)
    if dev.type == "cuda":
        state.attach_direct_grads(unet, *([text_encoder] if train_text_encoder else []))
        if lora_dropout_p == 0.0:
            # maskless adapters: the step's merged weight; the eager loop needs no explicit refresh — the first
            # adapter forward after an optimiser step re-merges (MergedWeights.lookup)
            state.enable_merged_weights(unet, *([text_encoder] if train_text_encoder else []))
    if wdt == torch.float16:
        state.enable_loss_scaling()
    perform_tuning(unet, vae, text_encoder, dataloader, max_train_steps_tuning, noise_scheduler, state, list(state.lrs),
                   # the reference builds optim.AdamW(params_to_optimize, weight_decay=...) WITHOUT an lr (ref :997), so
                   # diffusers' polynomial schedule reads lr_init = optimizer.defaults["lr"] = AdamW's 1e-3, not unet_lr
                   T.get_lr_lambda(lr_scheduler_lora, lr_warmup_steps_lora, max_train_steps_tuning, lr_init=1e-3), save_steps,
                   placeholder_token_ids, placeholder_tokens, output_dir, lora_unet_target_modules,
                   lora_clip_target_modules, mask_temperature, out_name, cached_latents, rows, rows_lr, world, is_main)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _parse_cli(argv: List[str]) -> dict:
    """The part of ``fire.Fire(train)`` the reference's shell examples use: ``--key=value``, ``--key value``, bare
    ``--flag`` (True), ``--noflag`` (False); values go through ``ast.literal_eval`` when they parse."""
    import ast

    def conv(v: str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    out, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected positional argument {a!r}; use --name=value")
        key = a[2:]
        if "=" in key:
            key, val = key.split("=", 1)
            out[key.replace("-", "_")] = conv(val)
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            out[key.replace("-", "_")] = conv(argv[i + 1])
            i += 1
        elif key.startswith("no") and len(key) > 2:
            out[key[2:].replace("-", "_")] = False
        else:
            out[key.replace("-", "_")] = True
        i += 1
    return out


def main():
    train(**_parse_cli(sys.argv[1:]))


if __name__ == "__main__":
    main()
