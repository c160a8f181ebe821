#!/usr/bin/env python
"""Drop-in for the reference's ``training_scripts/train_lora_dreambooth.py`` on MI355X.

Same command line (every flag of ref :168-483, same defaults), same outputs (``lora_weight.pt``,
``lora_weight.text_encoder.pt``, ``lora_weight.safetensors``, periodic ``lora_weight_e{E}_s{S}.pt``), same step
(ref :816-888).  What differs is how the step runs:

* one process per GPU under ``python -m torch.distributed.run`` (RANK/LOCAL_RANK/WORLD_SIZE), RCCL all-reduce of ONE
  flat f32 gradient buffer instead of ``accelerate``'s DDP buckets (the reference's only parallelism, ref :489-494);
* the LoRA adapters run on the gfx950 HIP kernels (``lora_amd``), clip + AdamW are one fused pass over the flat
  state, and with ``--mixed_precision bf16|fp16`` the frozen weights are kept resident in that dtype (f32 LoRA
  masters) instead of being re-cast by autocast every step;
* ``diffusers`` / a checkpoint directory are optional: when they are missing (this image has no network) the script
  trains the SD1.5-shaped stand-in UNet, a random-init ``transformers`` CLIP text encoder, a fixed stand-in VAE
  encoder and a hash tokenizer (``lora_amd/standin``), and ``--instance_data_dir synthetic:N`` generates N images.

Extra flags (not in the reference): ``--standin {sd15,tiny}``, ``--device``, ``--hip_graph``, ``--channels_last``,
``--merged_weights``.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from lora_amd import (extract_lora_ups_down, inject_trainable_lora, safetensors_available, save_lora_weight,  # noqa: E402
                      save_safeloras)
from lora_amd import trainer as T  # noqa: E402
from lora_amd.standin import io as SIO  # noqa: E402


def parse_args(input_args=None):
    p = argparse.ArgumentParser(description="LoRA DreamBooth fine-tuning (MI355X-native drop-in).")
    a = p.add_argument
    a("--pretrained_model_name_or_path", type=str, default=None, required=True,
      help="Checkpoint directory (diffusers layout) or hub id; 'standin' trains the built-in SD1.5-shaped stand-in.")
    a("--pretrained_vae_name_or_path", type=str, default=None, help="Optional separate VAE checkpoint.")
    a("--revision", type=str, default=None, required=False, help="Checkpoint revision.")
    a("--tokenizer_name", type=str, default=None, help="Tokenizer path if it differs from the model path.")
    a("--instance_data_dir", type=str, default=None, required=True,
      help="Folder with the instance images, or synthetic:N.")
    a("--class_data_dir", type=str, default=None, required=False, help="Folder with class (prior) images.")
    a("--instance_prompt", type=str, default=None, required=True, help="Prompt with the instance identifier.")
    a("--class_prompt", type=str, default=None, help="Prompt of the class images.")
    a("--with_prior_preservation", default=False, action="store_true", help="Add the prior-preservation loss.")
    a("--prior_loss_weight", type=float, default=1.0, help="Weight of the prior-preservation loss.")
    a("--num_class_images", type=int, default=100, help="Minimum number of class images.")
    a("--output_dir", type=str, default="text-inversion-model", help="Where the LoRA weights are written.")
    a("--output_format", type=str, choices=["pt", "safe", "both"], default="both", help="Output file format(s).")
    a("--seed", type=int, default=None, help="Seed for reproducible training.")
    a("--resolution", type=int, default=512, help="Training resolution (images are resized/cropped to it).")
    a("--center_crop", action="store_true", help="Center crop instead of random crop.")
    a("--color_jitter", action="store_true", help="Colour jitter augmentation.")
    a("--train_text_encoder", action="store_true", help="Also inject and train LoRA in the CLIP text encoder.")
    a("--train_batch_size", type=int, default=4, help="Batch size per device.")
    a("--sample_batch_size", type=int, default=4, help="Batch size for sampling class images.")
    a("--num_train_epochs", type=int, default=1)
    a("--max_train_steps", type=int, default=None, help="Total optimiser steps; overrides num_train_epochs.")
    a("--save_steps", type=int, default=500, help="Save LoRA weights every N steps.")
    a("--gradient_accumulation_steps", type=int, default=1, help="Accepted for compatibility (see note in main).")
    a("--gradient_checkpointing", action="store_true", help="Recompute block activations in backward.")
    a("--lora_rank", type=int, default=4, help="Rank of the LoRA factors.")
    a("--learning_rate", type=float, default=None, help="UNet LoRA learning rate.")
    a("--learning_rate_text", type=float, default=5e-6, help="Text-encoder LoRA learning rate.")
    a("--scale_lr", action="store_true", default=False, help="Scale lr by accumulation x batch x processes.")
    a("--lr_scheduler", type=str, default="constant",
      help='One of "linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup".')
    a("--lr_warmup_steps", type=int, default=500, help="Warm-up steps of the lr scheduler.")
    a("--use_8bit_adam", action="store_true", help="Accepted; the fused f32 AdamW over the flat LoRA state is used.")
    a("--adam_beta1", type=float, default=0.9)
    a("--adam_beta2", type=float, default=0.999)
    a("--adam_weight_decay", type=float, default=1e-2)
    a("--adam_epsilon", type=float, default=1e-08)
    a("--max_grad_norm", default=1.0, type=float, help="Gradient-norm clip.")
    a("--push_to_hub", action="store_true", help="Accepted; ignored (no network).")
    a("--hub_token", type=str, default=None)
    a("--logging_dir", type=str, default="logs", help="Log directory under output_dir (loss/lr as JSON lines).")
    a("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"], help="Compute precision.")
    a("--local_rank", type=int, default=-1, help="For distributed launchers.")
    a("--resume_unet", type=str, default=None, help="'.pt' LoRA list to resume the UNet adapters from.")
    a("--resume_text_encoder", type=str, default=None, help="'.pt' LoRA list to resume the text-encoder adapters from.")
    a("--resize", type=bool, default=True, required=False, help="Resize images before cropping.")
    a("--use_xformers", action="store_true", help="Accepted; ROCm uses PyTorch SDPA flash kernels.")
    # --- not in the reference
    a("--standin", type=str, default="sd15", choices=["sd15", "tiny"], help="Stand-in UNet size when no checkpoint.")
    a("--device", type=str, default=None, help="cuda | cpu (default: cuda if available).")
    a("--hip_graph", type=int, default=0, help="Capture forward+backward into a hipGraph (fixed batch shape).")
    a("--merged_weights", type=int, default=1, help="Run the maskless Linear adapters (--lora_dropout 0, the reference's "
      "default) on the step's merged weight W + scale*up@down: one merge launch per step, frozen GEMMs forward / "
      "input gradient, one launch for every site's factor gradients (DESIGN 9.1).  0: one fused kernel per site.")
    a("--channels_last", type=int, default=0, help="Run the UNet in NHWC (the layout MIOpen's convolutions use on "
      "MI355X; +5 %% steps/s on the SD1.5 stand-in).  Linear sites run as they are; Conv2d adapter sites take the "
      "channels-last MFMA kernels of csrc/conv_nhwc.hip (3x3) or the Linear kernels on the pixel rows (1x1).")
    args = p.parse_args(input_args) if input_args is not None else p.parse_args()

    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    if args.with_prior_preservation:
        if args.class_data_dir is None:
            raise ValueError("You must specify a data directory for class images.")
        if args.class_prompt is None:
            raise ValueError("You must specify prompt for class images.")
    else:
        if args.class_data_dir is not None:
            print("warning: --class_data_dir is unused without --with_prior_preservation.")
        if args.class_prompt is not None:
            print("warning: --class_prompt is unused without --with_prior_preservation.")
    if not safetensors_available:
        if args.output_format == "both":
            print("Safetensors is not available - changing output format to just output PyTorch files")
            args.output_format = "pt"
        elif args.output_format == "safe":
            raise ValueError("Safetensors is not available - either install it, or change output_format.")
    return args


def load_models(args, device):
    """(tokenizer, text_encoder, vae, unet, noise_scheduler, what): the real diffusers/transformers modules when a
    checkpoint directory and ``diffusers`` are available (ref :566-594, 678-680), the stand-ins otherwise."""
    return SIO.load_host_models(args.pretrained_model_name_or_path, args.pretrained_vae_name_or_path, args.revision,
                                args.tokenizer_name, device, args.standin, args.seed)


def main(args):
    dev_type = args.device or ("cuda" if torch.cuda.is_available() else "cpu")
    rank, local, world = T.init_distributed(dev_type)
    device = torch.device("cuda", local) if dev_type == "cuda" else torch.device("cpu")
    is_main = rank == 0
    logging_dir = Path(args.output_dir, args.logging_dir)
    if args.train_text_encoder and args.gradient_accumulation_steps > 1 and world > 1:
        raise ValueError("Gradient accumulation is not supported when training the text encoder in distributed "
                         "training. Please set gradient_accumulation_steps to 1.")
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if args.with_prior_preservation and not str(args.class_data_dir).startswith("synthetic:"):
        cdir = Path(args.class_data_dir)
        cdir.mkdir(parents=True, exist_ok=True)
        if len(SIO.list_images(str(cdir))) < args.num_class_images:
            # the reference samples the missing class images with the full pipeline (ref :512-558); without a
            # checkpoint there is nothing to sample from
            raise ValueError(f"{cdir} holds fewer than --num_class_images={args.num_class_images} images and no "
                             "diffusers pipeline is available to sample more; provide them or use synthetic:N")
    if is_main and args.output_dir is not None:
        os.makedirs(args.output_dir, exist_ok=True)
        os.makedirs(logging_dir, exist_ok=True)

    tokenizer, text_encoder, vae, unet, noise_scheduler, what = load_models(args, device)
    if is_main:
        print("models:", what)
    weight_dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(args.mixed_precision or "no", torch.float32)
    if device.type == "cpu":
        weight_dtype = torch.float32

    unet.requires_grad_(False)
    unet.to(device=device, dtype=weight_dtype)
    if args.channels_last and device.type == "cuda":
        unet.to(memory_format=torch.channels_last)
    unet_lora_params, _ = inject_trainable_lora(unet, r=args.lora_rank, loras=args.resume_unet)  # ref :596-598
    vae.requires_grad_(False)
    text_encoder.requires_grad_(False)
    vae.to(device)
    text_encoder.to(device=device, dtype=weight_dtype)
    if args.train_text_encoder:  # ref :608-621
        inject_trainable_lora(text_encoder, target_replace_module=["CLIPAttention"], r=args.lora_rank,
                              loras=args.resume_text_encoder)
    T.promote_lora_to_fp32(unet)
    T.promote_lora_to_fp32(text_encoder)
    if args.gradient_checkpointing:
        unet.enable_gradient_checkpointing()
        if args.train_text_encoder and hasattr(text_encoder, "gradient_checkpointing_enable"):
            text_encoder.gradient_checkpointing_enable()
    if args.learning_rate is None:
        raise ValueError("--learning_rate is required")
    if args.scale_lr:  # ref :632-638
        args.learning_rate = args.learning_rate * args.gradient_accumulation_steps * args.train_batch_size * world
    text_lr = args.learning_rate if args.learning_rate_text is None else args.learning_rate_text
    groups = [{"params": T.lora_params(unet), "lr": args.learning_rate, "weight_decay": args.adam_weight_decay}]
    if args.train_text_encoder:
        groups.append({"params": T.lora_params(text_encoder), "lr": text_lr, "weight_decay": args.adam_weight_decay})
    state = T.FlatLoraState(groups, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_epsilon,
                            max_grad_norm=args.max_grad_norm, device=device)
    if device.type == "cuda":
        state.attach_direct_grads(unet, *([text_encoder] if args.train_text_encoder else []))
    merged = None
    if device.type == "cuda" and args.merged_weights:
        merged = state.enable_merged_weights(unet, *([text_encoder] if args.train_text_encoder else []))
    if weight_dtype == torch.float16:  # fp16 activation gradients underflow without it (accelerate's GradScaler)
        state.enable_loss_scaling()
    base_lrs = list(state.lrs)

    dataset = SIO.DreamBoothDataset(args.instance_data_dir, args.instance_prompt, tokenizer,
                                    args.class_data_dir if args.with_prior_preservation else None, args.class_prompt,
                                    args.resolution, args.center_crop, args.resize,
                                    seed=(args.seed or 0) * 1000 + rank, content_seed=(args.seed or 0) * 1000)
    if args.color_jitter and is_main:
        print("warning: --color_jitter needs torchvision (not installed); ignored")
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True,
                                                              seed=args.seed or 0) if world > 1 else None
    loader = torch.utils.data.DataLoader(dataset, batch_size=args.train_batch_size, shuffle=sampler is None,
                                         sampler=sampler, num_workers=0,
                                         collate_fn=lambda ex: SIO.collate(ex, tokenizer, args.with_prior_preservation))

    steps_per_epoch = math.ceil(len(loader) / args.gradient_accumulation_steps)
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * steps_per_epoch
    args.num_train_epochs = math.ceil(args.max_train_steps / steps_per_epoch)
    lr_lambda = T.get_lr_lambda(args.lr_scheduler, args.lr_warmup_steps * args.gradient_accumulation_steps,
                                args.max_train_steps * args.gradient_accumulation_steps, lr_init=args.learning_rate)
    cfg = T.StepConfig(with_prior_preservation=args.with_prior_preservation, prior_loss_weight=args.prior_loss_weight,
                       num_train_timesteps=noise_scheduler.config.num_train_timesteps,
                       prediction_type=getattr(noise_scheduler.config, "prediction_type", "epsilon"))
    if is_main:
        print("***** Running training *****")
        print(f"  Num examples = {len(dataset)}")
        print(f"  Num batches each epoch = {len(loader)}")
        print(f"  Num Epochs = {args.num_train_epochs}")
        print(f"  Instantaneous batch size per device = {args.train_batch_size}")
        print(f"  Total train batch size (w. parallel, distributed & accumulation) = "
              f"{args.train_batch_size * world * args.gradient_accumulation_steps}")
        print(f"  Gradient Accumulation steps = {args.gradient_accumulation_steps}")
        print(f"  Total optimization steps = {args.max_train_steps}")
        print(f"  Trainable LoRA parameters = {state.n} ({state.payload_bytes} B all-reduce payload)")
    log_f = open(logging_dir / "dreambooth.jsonl", "a") if is_main else None

    def encode(batch):
        with torch.no_grad():  # frozen VAE (ref :818-821 leaves autograd on; nothing upstream needs a gradient)
            lat = vae.encode(batch["pixel_values"].to(device)).latent_dist.sample() * 0.18215
        return lat.to(weight_dtype), batch["input_ids"].to(device)

    te_arg = text_encoder  # the reference always runs the text encoder inside the step (ref :840)
    fwd_bwd = lambda lat, ids: T.forward_backward(unet, noise_scheduler, lat, ids, cfg, text_encoder=te_arg,  # noqa: E731
                                                  loss_scale=state.loss_scale, merged=merged)
    graphed = None
    global_step, last_save, t0 = 0, 0, time.perf_counter()
    done = False
    for epoch in range(args.num_train_epochs):
        unet.train()
        if args.train_text_encoder:
            text_encoder.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch in loader:
            lat, ids = encode(batch)
            if args.hip_graph and device.type == "cuda":
                if graphed is None or graphed.latents.shape != lat.shape:
                    graphed = T.GraphedForwardBackward(fwd_bwd, lat, ids, state)
                loss = graphed(lat, ids)
            else:
                loss = fwd_bwd(lat, ids)
            # NB the reference never enters accelerator.accumulate(): every batch is one optimiser step
            state.set_lrs([b * lr_lambda(global_step) for b in base_lrs])
            state.step(state.all_reduce())
            global_step += 1
            if args.save_steps and global_step - last_save >= args.save_steps and is_main:
                f_unet = f"{args.output_dir}/lora_weight_e{epoch}_s{global_step}.pt"
                f_text = f"{args.output_dir}/lora_weight_e{epoch}_s{global_step}.text_encoder.pt"
                print(f"save weights {f_unet}, {f_text}")
                save_lora_weight(unet, f_unet)
                if args.train_text_encoder:
                    save_lora_weight(text_encoder, f_text, target_replace_module=["CLIPAttention"])
                last_save = global_step
            if is_main and (global_step % 10 == 0 or global_step == args.max_train_steps):
                lv = float(loss.item())  # one host sync per 10 steps (the reference syncs every step, ref :959)
                dt = time.perf_counter() - t0
                print(f"step {global_step}/{args.max_train_steps} loss {lv:.5f} lr {state.lrs[0]:.3e} "
                      f"{global_step / dt:.2f} steps/s", flush=True)
                log_f.write('{"step": %d, "loss": %.6f, "lr": %.6e}\n' % (global_step, lv, state.lrs[0]))
                log_f.flush()
            if global_step >= args.max_train_steps:
                done = True
                break
        if done:
            break
    if world > 1:
        dist.barrier()
    if is_main:
        print("\n\nLora TRAINING DONE!\n\n")
        if args.output_format in ("pt", "both"):
            save_lora_weight(unet, args.output_dir + "/lora_weight.pt")
            if args.train_text_encoder:
                save_lora_weight(text_encoder, args.output_dir + "/lora_weight.text_encoder.pt",
                                 target_replace_module=["CLIPAttention"])
        if args.output_format in ("safe", "both"):
            loras = {"unet": (unet, {"CrossAttention", "Attention", "GEGLU"})}
            if args.train_text_encoder:
                loras["text_encoder"] = (text_encoder, {"CLIPAttention"})
            save_safeloras(loras, args.output_dir + "/lora_weight.safetensors")
        for up, down in extract_lora_ups_down(unet):
            print("First Unet Layer's Up Weight is now : ", up.weight.data.flatten()[:4])
            break
        log_f.close()
    if world > 1:
        dist.destroy_process_group()
    return global_step


if __name__ == "__main__":
    main(parse_args())
