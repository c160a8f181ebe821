"""lora_amd.cli_lora_add (lpl / ljl / upl) and lora_amd.lora_manager (lora_join, LoRAManager.tune)."""
import copy
import os
import types

import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import cli_lora_add as A
from lora_amd.lora_manager import LoRAManager, lora_join
from lora_amd.standin import tiny_unet


def _lora_file(path, r, seed, with_token=None):
    torch.manual_seed(seed)
    unet = tiny_unet()
    L.inject_trainable_lora(unet, r=r)
    for up, down in L.extract_lora_ups_down(unet):
        up.weight.data.normal_(0, 0.05)
    embeds = {with_token: torch.randn(32)} if with_token else {}
    L.save_safeloras_with_embeds({"unet": (unet, L.UNET_DEFAULT_TARGET_REPLACE)}, embeds, path)
    return unet


def test_lpl_and_ljl_on_safetensors(tmp_path):
    p1, p2, out = (str(tmp_path / n) for n in ("a.safetensors", "b.safetensors", "o.safetensors"))
    _lora_file(p1, 2, 1, "<a>"), _lora_file(p2, 2, 2, "<b>")
    A.add(p1, p2, out, alpha_1=0.3, alpha_2=0.7, mode="lpl")
    from safetensors import safe_open

    s1, s2, so = (safe_open(p, framework="pt") for p in (p1, p2, out))
    k = "unet:3:up"
    want = 0.3 * s1.get_tensor(k) + 0.7 * s2.get_tensor(k)
    assert torch.allclose(so.get_tensor(k).float(), want.float(), atol=1e-3)
    assert so.metadata()["<a>"] == L.EMBED_FLAG and so.metadata()["<b>"] == L.EMBED_FLAG
    # join: ranks add up, down stacked on dim 0 / up on dim 1, tokens renamed per file (lora_manager.py:13-71)
    A.add(p1, p2, out, mode="ljl")
    sj = safe_open(out, framework="pt")
    assert sj.get_tensor("unet:0:down").shape[0] == 4 and sj.get_tensor("unet:0:up").shape[1] == 4
    assert sj.metadata()["unet:0:rank"] == "4" and sj.metadata()["<s0-0>"] == L.EMBED_FLAG and "<s1-0>" in sj.keys()
    assert torch.equal(sj.get_tensor("unet:0:down")[:2], s1.get_tensor("unet:0:down"))
    tensors, meta, ranks, ntok = lora_join([s1, s2])
    assert ranks == [2, 2] and ntok == [1, 1]
    with pytest.raises(ValueError):
        A.add(p1, p2, out, mode="nope")


def test_lpl_on_pt_lists(tmp_path):
    torch.manual_seed(0)
    l1, l2 = [torch.randn(8, 2), torch.randn(2, 6)], [torch.randn(8, 2), torch.randn(2, 6)]
    p1, p2, out = (str(tmp_path / n) for n in ("a.pt", "b.pt", "o.pt"))
    torch.save(l1, p1), torch.save(l2, p2)
    A.add(p1, p2, out, alpha_1=0.25, alpha_2=0.75, mode="lpl", with_text_lora=True)  # no text files: skipped
    o = torch.load(out)
    assert torch.allclose(o[0], 0.25 * l1[0] + 0.75 * l2[0]) and torch.allclose(o[1], 0.25 * l1[1] + 0.75 * l2[1])


def _manager_case(device, tmp_path):
    p1, p2 = str(tmp_path / "a.safetensors"), str(tmp_path / "b.safetensors")
    u1, u2 = _lora_file(p1, 2, 1), _lora_file(p2, 3, 2)
    torch.manual_seed(0)
    base = tiny_unet()
    pipe = types.SimpleNamespace(unet=copy.deepcopy(base).to(device), text_encoder=torch.nn.Identity(), tokenizer=None)
    mgr = LoRAManager([p1, p2], pipe)
    assert mgr.ranklist == [2, 3]
    pipe.unet.eval()
    mgr.tune([0.5, 2.0])
    x, t = torch.randn(2, 4, 16, 16, device=device), torch.tensor([10, 500], device=device)
    ehs = torch.randn(2, 7, 32, device=device)
    with torch.no_grad():
        y = pipe.unet(x, t, ehs).sample
    # the same thing spelled out: base model with the two LoRAs merged at 0.5 and 2.0
    ref = copy.deepcopy(base)
    for src, alpha in ((p1, 0.5), (p2, 2.0)):
        loras = L.load_safeloras(src)["unet"]
        L.monkeypatch_or_replace_lora(ref, list(loras[0]), target_replace_module=L.UNET_DEFAULT_TARGET_REPLACE,
                                      r=loras[1])
        L.collapse_lora(ref, alpha)
        L.monkeypatch_remove_lora(ref)
    ref.eval()
    with torch.no_grad():
        yr = ref(x.cpu(), t.cpu(), ehs.cpu()).sample
    return y.cpu(), yr


def test_manager_tune_equals_weighted_merge_cpu(tmp_path):
    y, yr = _manager_case("cpu", tmp_path)
    assert torch.allclose(y, yr, atol=2e-3, rtol=2e-3), (y - yr).abs().max()


@pytest.mark.gpu
def test_manager_tune_equals_weighted_merge_on_device(tmp_path):
    """The diagonal selector runs inside the HIP adapter kernels (``sel`` operand)."""
    y, yr = _manager_case("cuda:0", tmp_path)
    assert torch.allclose(y, yr, atol=5e-3, rtol=5e-3), (y - yr).abs().max()


def test_upl_merges_lora_into_model(tmp_path):
    lp = str(tmp_path / "l.safetensors")
    torch.manual_seed(4)
    unet = tiny_unet()
    L.inject_trainable_lora(unet, r=2)
    for up, _ in L.extract_lora_ups_down(unet):
        up.weight.data.normal_(0, 0.05)
    L.save_safeloras({"unet": (unet, L.UNET_DEFAULT_TARGET_REPLACE)}, lp)
    out = str(tmp_path / "merged")
    A.add("standin:7", lp, out, alpha_1=0.8, mode="upl")
    sd = torch.load(os.path.join(out, "unet.pt"))
    assert not any("lora" in k for k in sd)  # adapters stripped after the merge
    torch.manual_seed(7)
    base = tiny_unet().state_dict()
    moved = [k for k in sd if k in base and not torch.equal(sd[k], base[k])]
    assert moved and all(k.endswith("weight") for k in moved)
