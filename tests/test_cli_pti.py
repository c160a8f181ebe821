"""``lora_amd.cli_lora_pti`` (drop-in for the reference's lora_pti console script) end to end on CPU at toy size."""
import inspect
import os

import torch

import lora_amd as L
from lora_amd import cli_lora_pti as pti
from tests import helpers as H


def test_keyword_surface_matches_reference():
    names = list(inspect.signature(pti.train).parameters)
    ref = """instance_data_dir pretrained_model_name_or_path output_dir train_text_encoder pretrained_vae_name_or_path
    revision perform_inversion use_template train_inpainting placeholder_tokens placeholder_token_at_data
    initializer_tokens seed resolution color_jitter train_batch_size sample_batch_size max_train_steps_tuning
    max_train_steps_ti save_steps gradient_accumulation_steps gradient_checkpointing lora_rank lora_unet_target_modules
    lora_clip_target_modules lora_dropout_p lora_scale use_extended_lora clip_ti_decay learning_rate_unet
    learning_rate_text learning_rate_ti continue_inversion continue_inversion_lr use_face_segmentation_condition
    cached_latents use_mask_captioned_data mask_temperature scale_lr lr_scheduler lr_warmup_steps lr_scheduler_lora
    lr_warmup_steps_lora weight_decay_ti weight_decay_lora use_8bit_adam device extra_args log_wandb wandb_log_prompt_cnt
    wandb_project_name wandb_entity proxy_token enable_xformers_memory_efficient_attention out_name""".split()
    assert len(ref) == 55 and names[:55] == ref  # same names, same order (ref cli_lora_pti.py:696-752)
    sig = inspect.signature(pti.train).parameters
    assert sig["max_train_steps_ti"].default == 1000 and sig["lr_scheduler"].default == "linear"
    assert sig["weight_decay_lora"].default == 0.001 and sig["device"].default == "cuda:0"
    if H.reference_available():  # defaults identical to the live reference signature, read textually (it cannot be imported)
        import ast
        import re

        src = open("/root/reference/lora_diffusion/cli_lora_pti.py").read()
        fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "train")
        ref_names = [a.arg for a in fn.args.args]
        assert ref_names == ref
        defaults = dict(zip(ref_names[len(ref_names) - len(fn.args.defaults):], fn.args.defaults))
        for k, node in defaults.items():
            try:
                want = ast.literal_eval(node)
            except ValueError:
                continue
            assert sig[k].default == want, k


def test_parse_cli_shim():
    kw = pti._parse_cli(["--instance_data_dir=synthetic:3", "--lora_rank", "8", "--use_extended_lora", "--notrain_text_encoder",
                         "--learning_rate_unet=1e-4", "--placeholder_tokens", "<s1>|<s2>"])
    assert kw == {"instance_data_dir": "synthetic:3", "lora_rank": 8, "use_extended_lora": True,
                  "train_text_encoder": False, "learning_rate_unet": 1e-4, "placeholder_tokens": "<s1>|<s2>"}


def test_pti_two_phases_write_reference_files(tmp_path):
    out = str(tmp_path / "pti")
    pti.train(instance_data_dir="synthetic:3", pretrained_model_name_or_path="standin", output_dir=out, standin="tiny",
              placeholder_tokens="<s1>|<s2>", use_template="object", resolution=64, train_batch_size=1,
              max_train_steps_ti=4, max_train_steps_tuning=4, save_steps=2, gradient_accumulation_steps=2,
              lora_rank=2, use_extended_lora=True, learning_rate_ti=5e-3, device="cpu", continue_inversion=True,
              out_name="final")
    files = set(os.listdir(out))
    assert {"step_inv_2.safetensors", "step_inv_4.safetensors", "step_2.safetensors", "step_4.safetensors",
            "final.safetensors"} <= files
    # TI-only file: embeddings only, flagged <embed> (lora.py:478-480)
    from safetensors import safe_open

    with safe_open(os.path.join(out, "step_inv_4.safetensors"), framework="pt") as f:
        assert set(f.keys()) == {"<s1>", "<s2>"} and f.metadata()["<s1>"] == L.EMBED_FLAG
        e_inv = f.get_tensor("<s1>")
    loras, embeds = L.load_safeloras_both(os.path.join(out, "final.safetensors"))
    assert set(loras) == {"unet", "text_encoder"} and set(embeds) == {"<s1>", "<s2>"}
    ups = loras["unet"][0][0::2]
    assert any(u.dim() == 4 for u in ups) and any(u.dim() == 2 for u in ups)  # conv AND linear adapters (extended)
    assert max(float(u.abs().max()) for u in ups) > 0
    assert e_inv.shape == embeds["<s1>"].shape
    assert not torch.allclose(e_inv.float(), embeds["<s1>"].float())  # continue_inversion kept moving the row


def test_placeholder_rows_equal_full_table_adamw():
    """Updating only the placeholder rows == the reference's AdamW over the whole table followed by restoring every
    other row (ref :433-479), including the norm decay."""
    torch.manual_seed(0)
    V, Hd, ids, lr, wd = 50, 16, [47, 49], 5e-3, 0.01
    emb_a = torch.nn.Embedding(V, Hd)
    emb_b = torch.nn.Embedding(V, Hd)
    emb_b.load_state_dict(emb_a.state_dict())

    class TE(torch.nn.Module):
        def __init__(self, e):
            super().__init__()
            self.e = e

        def get_input_embeddings(self):
            return self.e

    rows = pti.PlaceholderRows(TE(emb_a), ids, lr, wd)
    opt = torch.optim.AdamW(emb_b.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    orig = emb_b.weight.data.clone()
    keep = torch.ones(V, dtype=torch.bool)
    keep[ids] = False
    for step in range(4):
        tok = torch.tensor([[47, 3, 49, 5], [1, 49, 47, 2]])
        tgt = torch.randn(2, 4, Hd, generator=torch.Generator().manual_seed(step))
        for e in (emb_a, emb_b):
            ((e(tok) - tgt) ** 2).mean().backward()
        rows.step(lr, 1, True)
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            w = emb_b.weight
            pre = w[~keep].norm(dim=-1, keepdim=True)
            lam = min(1.0, 100 * lr)
            w[~keep] = torch.nn.functional.normalize(w[~keep], dim=-1) * (pre + lam * (0.4 - pre))
            w[keep] = orig[keep]
    assert torch.allclose(emb_a.weight, emb_b.weight, atol=1e-6)


def test_pti_two_gloo_ranks(tmp_path):
    """The PTI CLI under ``torch.distributed.run`` with 2 CPU ranks (a capability the reference CLI does not have):
    image shards per rank, all-reduced placeholder-row and LoRA gradients, rank-0 saves."""
    import subprocess
    import sys

    out = str(tmp_path / "dist")
    env = {**os.environ, "MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "2"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", "-m", "lora_amd.cli_lora_pti", "--instance_data_dir=synthetic:4",
           "--pretrained_model_name_or_path=standin", f"--output_dir={out}", "--standin=tiny", "--placeholder_tokens=<s1>",
           "--use_template=object", "--resolution=64", "--max_train_steps_ti=2", "--max_train_steps_tuning=2",
           "--save_steps=2", "--gradient_accumulation_steps=1", "--lora_rank=2", "--device=cpu", "--out_name=final"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=H.REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    loras, embeds = L.load_safeloras_both(os.path.join(out, "final.safetensors"))
    assert set(loras) == {"unet", "text_encoder"} and set(embeds) == {"<s1>"}
