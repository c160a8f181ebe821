"""The drop-in DreamBooth CLI on one MI355X: bf16-resident frozen weights, f32 LoRA masters, UNet + CLIP text-encoder
adapters on the HIP kernels (BASELINE configs[2] geometry at toy size), eager and hipGraph-replayed."""
import os
import sys

import pytest
import torch

import lora_amd as L
from tests.helpers import REPO

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(REPO, "training_scripts"))
import train_lora_dreambooth as cli  # noqa: E402

BASE = ["--pretrained_model_name_or_path", "standin", "--standin", "tiny", "--instance_data_dir", "synthetic:4",
        "--instance_prompt", "a photo of sks dog", "--resolution", "128", "--train_batch_size", "2", "--learning_rate",
        "1e-3", "--lr_warmup_steps", "0", "--device", "cuda", "--seed", "3", "--mixed_precision", "bf16",
        "--train_text_encoder", "--lora_rank", "8", "--output_format", "safe"]


@pytest.mark.parametrize("graph", [0, 1])
def test_cli_on_device(tmp_path, graph):
    out = str(tmp_path / f"g{graph}")
    steps = cli.main(cli.parse_args(BASE + ["--output_dir", out, "--max_train_steps", "4", "--hip_graph", str(graph)]))
    assert steps == 4
    loras = L.load_safeloras(os.path.join(out, "lora_weight.safetensors"))
    assert set(loras) == {"unet", "text_encoder"}
    ups = loras["unet"][0][0::2]
    assert all(torch.isfinite(u).all() for u in ups) and max(float(u.abs().max()) for u in ups) > 0
    te_ups = loras["text_encoder"][0][0::2]
    assert max(float(u.abs().max()) for u in te_ups) > 0  # the text-encoder adapters trained too


def test_pti_extended_on_device(tmp_path):
    """cli_lora_pti with --use_extended_lora at rank 16 (BASELINE configs[3] geometry at toy size): TI phase, then LoRA
    tuning with Linear + Conv2d adapters on the HIP kernels (dropout 0.1 on the conv adapters, as the reference's
    extended injection leaves it)."""
    from lora_amd import cli_lora_pti as pti

    out = str(tmp_path / "pti")
    pti.train(instance_data_dir="synthetic:4", pretrained_model_name_or_path="standin", output_dir=out, standin="tiny",
              placeholder_tokens="<s1>", use_template="object", resolution=256, train_batch_size=2,
              max_train_steps_ti=2, max_train_steps_tuning=4, save_steps=4, gradient_accumulation_steps=1,
              lora_rank=16, use_extended_lora=True, device="cuda:0", out_name="final")
    loras, embeds = L.load_safeloras_both(os.path.join(out, "final.safetensors"))
    ups = loras["unet"][0][0::2]
    assert any(u.dim() == 4 for u in ups) and all(torch.isfinite(u).all() for u in ups)
    conv_ups = [u for u in ups if u.dim() == 4]
    assert max(float(u.abs().max()) for u in conv_ups) > 0 and set(embeds) == {"<s1>"}
