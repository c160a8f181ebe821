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


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_manager_composition_of_one_site_on_device_vs_the_oracle(tmp_path, dt):
    """Multi-LoRA composition (lora_manager.py:13-71 ``lora_join`` + :122-131 ``tune`` -> lora.py:63-70 ``set_selector_from_diag``)
    at ONE Linear site, against oracle/lora_numpy directly: two member LoRAs of ranks 2 and 3 joined to rank 5 (down rows / up
    columns concatenated, as the join does), weights 0.5 and 2.0 repeated over the members' rank slices, the diagonal selector
    applied INSIDE the device kernels — vs oracle.lora_linear_forward(selector = diag) and vs the weighted sum of the two members'
    own branches (the statement the manager makes)."""
    from oracle import lora_numpy as O
    from tests.test_gpu_kernels import close, n, rnd

    DEV = "cuda:0"
    M, K, N, s = 300, 320, 640, 0.7
    name = "f32" if dt == torch.float32 else "bf16"
    x = rnd((M, K), name, seed=1)
    w, b = rnd((N, K), name, 0.05, seed=2), rnd((N,), name, seed=3)
    downs = [rnd((2, K), "f32", 0.3, seed=4), rnd((3, K), "f32", 0.3, seed=5)]
    ups = [rnd((N, 2), "f32", 0.2, seed=6), rnd((N, 3), "f32", 0.2, seed=7)]
    scales, ranklist = [0.5, 2.0], [2, 3]
    m = L.LoraInjectedLinear(K, N, True, r=5, dropout_p=0.0, scale=s).to(DEV)
    m.linear.weight.data.copy_(w.float()), m.linear.bias.data.copy_(b.float())
    if dt != torch.float32:
        m.linear.to(dt)
    m.lora_down.weight.data.copy_(torch.cat(downs, 0))
    m.lora_up.weight.data.copy_(torch.cat(ups, 1))
    diag = torch.repeat_interleave(torch.tensor(scales), torch.tensor(ranklist))   # LoRAManager.tune
    m.set_selector_from_diag(diag)
    m.eval()
    with torch.no_grad():
        y = m(x)
    X, W, Bv = n(x), n(w), n(b)
    A_, U_ = np.concatenate([n(d) for d in downs], 0), np.concatenate([n(u) for u in ups], 1)
    yo, _ = O.lora_linear_forward(X, W, Bv, A_, U_, s, selector=np.diag(n(diag)))
    # ... which is the weighted sum of the members' branches on top of the frozen layer
    y_sum = X @ W.T + Bv + s * sum(a * (X @ n(d).T) @ n(u).T for a, d, u in zip(scales, downs, ups))
    np.testing.assert_allclose(yo, y_sum, rtol=2e-5, atol=2e-5 * np.abs(y_sum).max())
    absy = np.abs(X) @ np.abs(W).T + np.abs(Bv) + s * (np.abs(X) @ np.abs(A_).T * np.abs(n(diag))) @ np.abs(U_).T
    close(n(y), yo, absy, name, k=2.0 if dt != torch.float32 else 2e-5, msg="y")
