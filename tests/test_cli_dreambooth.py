"""The drop-in DreamBooth CLI end to end on CPU (BASELINE configs[0]: plumbing, no GPU): flag surface, outputs,
file layout, resume, and a 2-process gloo launch."""
import json
import os
import subprocess
import sys

import pytest
import torch

import lora_amd as L
from tests.helpers import REPO

sys.path.insert(0, os.path.join(REPO, "training_scripts"))
import train_lora_dreambooth as cli  # noqa: E402

BASE = ["--pretrained_model_name_or_path", "standin", "--standin", "tiny", "--instance_data_dir", "synthetic:4",
        "--instance_prompt", "a photo of sks dog", "--resolution", "64", "--train_batch_size", "2", "--learning_rate",
        "1e-3", "--lr_scheduler", "constant", "--lr_warmup_steps", "0", "--device", "cpu", "--seed", "3"]


def test_flag_surface_matches_reference_names():
    """Every flag of the reference's parser (ref training_scripts/train_lora_dreambooth.py:168-483) is accepted."""
    ref_flags = """pretrained_model_name_or_path pretrained_vae_name_or_path revision tokenizer_name instance_data_dir
    class_data_dir instance_prompt class_prompt with_prior_preservation prior_loss_weight num_class_images output_dir
    output_format seed resolution center_crop color_jitter train_text_encoder train_batch_size sample_batch_size
    num_train_epochs max_train_steps save_steps gradient_accumulation_steps gradient_checkpointing lora_rank
    learning_rate learning_rate_text scale_lr lr_scheduler lr_warmup_steps use_8bit_adam adam_beta1 adam_beta2
    adam_weight_decay adam_epsilon max_grad_norm push_to_hub hub_token logging_dir mixed_precision local_rank resume_unet
    resume_text_encoder resize use_xformers""".split()
    assert len(ref_flags) == 46
    args = cli.parse_args(BASE)
    for f in ref_flags:
        assert hasattr(args, f), f
    assert (args.lora_rank, args.train_batch_size, args.save_steps, args.output_format) == (4, 2, 500, "both")
    assert (args.learning_rate_text, args.adam_weight_decay, args.max_grad_norm, args.prior_loss_weight) == (5e-6, 1e-2, 1.0, 1.0)
    with pytest.raises(ValueError):
        cli.parse_args(BASE + ["--with_prior_preservation"])  # class dir / prompt missing


def test_cli_trains_and_writes_reference_layout(tmp_path):
    out = str(tmp_path / "out")
    args = cli.parse_args(BASE + ["--output_dir", out, "--max_train_steps", "3", "--save_steps", "2",
                                  "--train_text_encoder", "--with_prior_preservation", "--class_data_dir", "synthetic:4",
                                  "--class_prompt", "a photo of a dog", "--lora_rank", "2"])
    assert cli.main(args) == 3
    files = set(os.listdir(out))
    assert {"lora_weight.pt", "lora_weight.text_encoder.pt", "lora_weight.safetensors", "lora_weight_e0_s2.pt",
            "lora_weight_e0_s2.text_encoder.pt", "logs"} <= files
    # safetensors: reference key/metadata layout (lora.py:463-483), f16 tensors, ranks recorded
    from safetensors import safe_open

    with safe_open(os.path.join(out, "lora_weight.safetensors"), framework="pt") as f:
        keys, meta = set(f.keys()), f.metadata()
        assert "unet:0:up" in keys and "unet:0:down" in keys and "text_encoder:0:up" in keys
        assert set(json.loads(meta["unet"])) == {"CrossAttention", "Attention", "GEGLU"}
        assert json.loads(meta["text_encoder"]) == ["CLIPAttention"] and meta["unet:0:rank"] == "2"
        assert f.get_tensor("unet:0:up").dtype == torch.float16 and f.get_tensor("unet:0:down").shape[0] == 2
    n_unet = len([k for k in keys if k.startswith("unet:") and k.endswith(":up")])
    lst = torch.load(os.path.join(out, "lora_weight.pt"))
    assert len(lst) == 2 * n_unet and lst[0].dtype == torch.float16  # [up0, down0, up1, ...] (lora.py:429-436)
    # the trained factors moved away from the init (up == 0)
    assert float(lst[0].float().abs().max()) > 0
    # resume: the adapters start from the saved list (lora.py:271-272, 301-303)
    out2 = str(tmp_path / "out2")
    args2 = cli.parse_args(BASE + ["--output_dir", out2, "--max_train_steps", "1", "--lora_rank", "2", "--resume_unet",
                                   os.path.join(out, "lora_weight.pt"), "--output_format", "pt", "--learning_rate", "0"])
    cli.main(args2)
    lst2 = torch.load(os.path.join(out2, "lora_weight.pt"))
    for a, b in zip(lst, lst2):
        assert torch.allclose(a.float(), b.float(), atol=2e-3)  # lr 0 + f16 round trip: unchanged


def test_cli_two_gloo_ranks(tmp_path):
    out = str(tmp_path / "dist")
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "2"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(REPO, "training_scripts", "train_lora_dreambooth.py")] + \
        BASE + ["--output_dir", out, "--max_train_steps", "2", "--output_format", "safe", "--scale_lr"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Total train batch size (w. parallel, distributed & accumulation) = 4" in r.stdout
    assert os.path.exists(os.path.join(out, "lora_weight.safetensors"))
    L.load_safeloras(os.path.join(out, "lora_weight.safetensors"))
