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


# ----------------------------------------------------------------------------- 2 ranks == 1 process, through the CLI itself
def _record_steps(cli_mod, rank, steps, tail, feed=None):
    """Wrap the CLI's step: forward_backward gets explicit noise / timesteps (recorded, or replayed from ``feed`` together
    with the samples and the parameters the 2-rank run had at that step); FlatLoraState.step records the reduced gradient."""
    orig_fb, orig_step = cli_mod.T.forward_backward, cli_mod.T.FlatLoraState.step
    states = []

    def fb(unet, sched, lat, ids, cfg, **kw):
        k = len(steps)
        if feed is None:
            g = torch.Generator().manual_seed(1000 * k + rank)
            noise, ts = torch.randn(lat.shape, generator=g), torch.randint(0, 1000, (lat.shape[0],), generator=g)
        else:
            lat, ids, noise, ts = feed["steps"][k]
            states[0].flat_p.copy_(feed["tail"][k][1])   # one step deep: this step starts where the 2-rank run's did
            states[0].exp_avg.copy_(feed["tail"][k][2]), states[0].exp_avg_sq.copy_(feed["tail"][k][3])
        steps.append((lat.clone(), ids.clone(), noise, ts))
        return orig_fb(unet, sched, lat, ids, cfg, noise=noise, timesteps=ts, **kw)

    def step(self, scale=1.0, **k):
        before = (self.flat_g.clone() * scale, self.flat_p.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone())
        out = orig_step(self, scale, **k)
        tail.append(before + (self.flat_p.clone(),))
        return out

    orig_init = cli_mod.T.FlatLoraState.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        states.append(self)

    cli_mod.T.forward_backward, cli_mod.T.FlatLoraState.step, cli_mod.T.FlatLoraState.__init__ = fb, step, init
    return lambda: (setattr(cli_mod.T, "forward_backward", orig_fb), setattr(cli_mod.T.FlatLoraState, "step", orig_step),
                    setattr(cli_mod.T.FlatLoraState, "__init__", orig_init))


def _at_128(base, batch=None):
    """BASE at 128 px (16 x 16 latents: at 64 px the tiny UNet's deepest GroupNorms see 1 x 1 maps and amplify f32
    summation-order differences into tens of percent of a gradient — measured: bit-equal with equal thread counts, 30 % apart
    between 1 and 8 threads) and, optionally, another per-process batch size."""
    out = list(base)
    out[out.index("--resolution") + 1] = "128"
    if batch is not None:
        out[out.index("--train_batch_size") + 1] = str(batch)
    return out


def _dp_worker(rank, world, port, out, rec_dir):
    """One rank of `train_lora_dreambooth.py --scale_lr` under gloo.  Recorded per rank: the dataset indices its
    DistributedSampler handed out, every step's samples / noise / timesteps, the reduced gradient and the parameters before
    and after each optimiser step, who saved."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from lora_amd.standin import io as SIO

    idx_log, saves, steps, tail, seen = [], [], [], [], {}
    orig_get = SIO.DreamBoothDataset.__getitem__

    def get(self, i):
        ex = orig_get(self, i)
        idx_log.append(int(i))
        seen[int(i)] = float(ex["instance_images"].double().sum())   # fingerprint of image i as THIS rank sees it
        return ex

    SIO.DreamBoothDataset.__getitem__ = get
    _record_steps(cli, rank, steps, tail)
    for name in ("save_lora_weight", "save_safeloras"):
        orig = getattr(cli, name)
        setattr(cli, name, (lambda o, nm: lambda *a, **k: (saves.append(nm), o(*a, **k))[1])(orig, name))
    args = cli.parse_args(_at_128(BASE) + ["--output_dir", out, "--max_train_steps", "4", "--output_format", "safe",
                                           "--scale_lr", "--save_steps", "2", "--lora_rank", "2"])
    assert cli.main(args) == 4
    torch.save(dict(idx=idx_log, saves=saves, steps=steps, tail=tail, seen=seen), os.path.join(rec_dir, f"rank{rank}.pt"))


def test_cli_two_gloo_ranks_equal_the_single_process_large_batch_run(tmp_path):
    """ref train_lora_dreambooth.py:632-638 (--scale_lr: lr x accumulation x batch x processes), :744-757 (accelerate shards
    the loader: here a DistributedSampler), :877-884 (gradients averaged over ranks, then clip, then AdamW), :895 / :969
    (rank-0-only saves).  The CLI under 2 gloo ranks of batch 2 against ONE process of batch 4 fed the two ranks' samples,
    noise and timesteps side by side: per step, from the same parameters and moments (one step deep — from ``up = 0`` AdamW
    turns summation-order noise in a near-zero gradient into +-lr, so free-running trajectories part within two steps), the
    averaged gradient, the effective learning rate and the update agree; the replicas stay identical; rank 0 alone saves."""
    import torch.multiprocessing as mp
    from safetensors.torch import load_file

    from tests.test_dp_gloo import _free_port

    out2, rec = str(tmp_path / "two"), str(tmp_path / "rec")
    os.makedirs(rec)
    mp.spawn(_dp_worker, args=(2, _free_port(), out2, rec), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(rec, f"rank{r}.pt"), weights_only=False) for r in (0, 1))
    # the sampler partitions every epoch: 4 images, 2 ranks x batch 2 -> one step per epoch, disjoint halves, union = all
    assert len(r0["idx"]) == len(r1["idx"]) == 8
    for e in range(4):
        a, b = set(r0["idx"][2 * e:2 * e + 2]), set(r1["idx"][2 * e:2 * e + 2])
        assert not (a & b) and a | b == {0, 1, 2, 3}, (e, a, b)
    # ... of ONE dataset: image i is the same image on both ranks, and the four images differ
    assert set(r0["seen"]) == set(r1["seen"]) == {0, 1, 2, 3} and r0["seen"] == r1["seen"] and len(set(r0["seen"].values())) == 4
    # rank 0 alone writes: two periodic saves + the final safetensors; rank 1 nothing
    assert r0["saves"] == ["save_lora_weight", "save_lora_weight", "save_safeloras"] and r1["saves"] == []
    assert {"lora_weight_e1_s2.pt", "lora_weight_e3_s4.pt", "lora_weight.safetensors"} <= set(os.listdir(out2))
    # replicas: same reduced gradient and same parameters on both ranks at every step (bit for bit: one all-reduce result)
    for t0, t1 in zip(r0["tail"], r1["tail"]):
        assert torch.equal(t0[0], t1[0]) and torch.equal(t0[4], t1[4])
    # the saved file = rank 0's final factors (f16 on disk)
    two = load_file(os.path.join(out2, "lora_weight.safetensors"))
    assert len(two) > 10 and all(float(v.float().abs().max()) > 0 for v in two.values())

    # ONE process, batch 4, fed the concatenation of what the two ranks saw, each step from the 2-rank run's state
    feed = dict(steps=[tuple(torch.cat([a, b]) for a, b in zip(s0, s1)) for s0, s1 in zip(r0["steps"], r1["steps"])],
                tail=r0["tail"])
    assert len(feed["steps"]) == 4 and feed["steps"][0][0].shape[0] == 4
    steps, tail = [], []
    undo = _record_steps(cli, 0, steps, tail, feed)
    try:
        args = cli.parse_args(_at_128(BASE, batch=4) + ["--output_dir", str(tmp_path / "one"), "--max_train_steps", "4",
                                                        "--output_format", "safe", "--scale_lr", "--lora_rank", "2"])
        assert args.train_batch_size == 4
        assert cli.main(args) == 4 and len(steps) == 4
    finally:
        undo()
    for k, (one_t, two_t) in enumerate(zip(tail, r0["tail"])):
        g1, g2 = one_t[0], two_t[0]
        gmax = float(g2.abs().max())
        assert gmax > 0 and float((g1 - g2).abs().max()) <= 2e-3 * gmax, (k, float((g1 - g2).abs().max()), gmax)
        assert float(g1 @ g2 / (g1.norm() * g2.norm())) >= 0.99999, k
        # same effective lr (lr x batch x processes either way) and same clip: the update of every element whose gradient's
        # sign and size are not in doubt
        solid = g2.abs() > 1e-2 * gmax
        u1, u2 = (one_t[4] - one_t[1])[solid], (two_t[4] - two_t[1])[solid]
        assert int(solid.sum()) > 50 and float((u1 - u2).abs().max()) <= 0.02 * float(u2.abs().max()), k
