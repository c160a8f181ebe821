"""lora_amd's drop-in API on CPU tensors, against fixtures the real reference produced
(tests/golden, scripts/make_golden.py) and — when /root/reference is present — the live reference."""
import copy
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.nn as nn

import lora_amd as L
from tests import helpers as H

G = H.GOLDEN


def _npz(name):
    return dict(np.load(os.path.join(G, name)))


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_adapter_contract():
    m = L.LoraInjectedLinear(16, 24, True, r=4)
    assert set(dict(m.named_children())) == {"linear", "lora_down", "dropout", "lora_up", "selector"}
    assert sorted(m.state_dict()) == ["linear.bias", "linear.weight", "lora_down.weight", "lora_up.weight"]
    assert m.r == 4 and m.scale == 1.0 and m.dropout.p == 0.1 and isinstance(m.selector, nn.Identity)
    assert m.lora_up.weight.abs().sum() == 0 and m.lora_down.weight.shape == (4, 16)
    with pytest.raises(ValueError, match="LoRA rank 17 must be less or equal than 16"):
        L.LoraInjectedLinear(16, 24, r=17)
    c = L.LoraInjectedConv2d(8, 12, 3, 2, 1, r=4)
    assert c.lora_down.weight.shape == (4, 8, 3, 3) and c.lora_down.stride == (2, 2) and c.lora_down.padding == (1, 1)
    assert c.lora_up.weight.shape == (12, 4, 1, 1) and c.conv.bias is not None
    with pytest.raises(ValueError):
        L.LoraInjectedConv2d(8, 12, 3, r=9)
    torch.manual_seed(0)
    big = L.LoraInjectedLinear(512, 512, r=8)
    assert abs(big.lora_down.weight.std().item() - 1 / 8) < 0.01  # N(0, 1/r), ref:50


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_linear_cpu_matches_reference_vectors(tag):
    d = _npz("linear_cases.npz")
    M, K = d[f"{tag}_x"].shape
    N, r = d[f"{tag}_up"].shape
    m = L.LoraInjectedLinear(K, N, f"{tag}_b" in d, r=r, dropout_p=0.0, scale=float(d[f"{tag}_scale"]))
    m.linear.weight.data = torch.from_numpy(d[f"{tag}_W"])
    if f"{tag}_b" in d:
        m.linear.bias.data = torch.from_numpy(d[f"{tag}_b"])
    m.lora_down.weight.data = torch.from_numpy(d[f"{tag}_down"])
    m.lora_up.weight.data = torch.from_numpy(d[f"{tag}_up"])
    if f"{tag}_sel" in d:
        m.set_selector_from_diag(torch.from_numpy(np.diag(d[f"{tag}_sel"]).copy()))
    x = torch.from_numpy(d[f"{tag}_x"]).requires_grad_(True)
    y = m(x)
    (y * torch.from_numpy(d[f"{tag}_gy"])).sum().backward()
    np.testing.assert_allclose(H.t2n(y), d[f"{tag}_y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(H.t2n(x.grad), d[f"{tag}_dx"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H.t2n(m.lora_down.weight.grad), d[f"{tag}_ddown"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H.t2n(m.lora_up.weight.grad), d[f"{tag}_dup"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_conv_cpu_matches_reference_vectors(tag):
    d = _npz("conv_cases.npz")
    k, s, p, r = (int(v) for v in d[f"{tag}_geom"])
    Co, Ci = d[f"{tag}_W"].shape[:2]
    m = L.LoraInjectedConv2d(Ci, Co, k, s, p, r=r, dropout_p=0.0, scale=float(d[f"{tag}_scale"]))
    for mod, key in ((m.conv, "W"), (m.lora_down, "down"), (m.lora_up, "up")):
        mod.weight.data = torch.from_numpy(d[f"{tag}_{key}"])
    m.conv.bias.data = torch.from_numpy(d[f"{tag}_b"])
    x = torch.from_numpy(d[f"{tag}_x"]).requires_grad_(True)
    y = m(x)
    (y * torch.from_numpy(d[f"{tag}_gy"])).sum().backward()
    np.testing.assert_allclose(H.t2n(y), d[f"{tag}_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(H.t2n(m.lora_up.weight.grad), d[f"{tag}_dup"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(H.t2n(m.lora_down.weight.grad), d[f"{tag}_ddown"], rtol=1e-4, atol=1e-4)


def test_traversal_order_matches_reference():
    data = json.load(open(os.path.join(G, "traversal_cases.json")))
    kind2cls = {"linear": nn.Linear, "conv": nn.Conv2d, "lora_linear": L.LoraInjectedLinear,
                "lora_conv": L.LoraInjectedConv2d}
    for case in data["cases"]:
        root = H.build_tree(data["trees"][case["tree"]], L.LoraInjectedLinear, L.LoraInjectedConv2d)
        paths = H.module_paths(root)
        got = list(L._find_modules_v2(root, set(case["ancestors"]) if case["ancestors"] is not None else None,
                                      search_class=[kind2cls[k] for k in case["kinds"]]))
        assert [paths[id(m)] for _, _, m in got] == case["paths"], case
        assert [n for _, n, _ in got] == case["names"]


def _toy_models(info):
    torch.manual_seed(7)
    unet, clip = H.build_tree(H.toy_unet_spec()), H.build_tree(H.toy_clip_spec())
    params, names = L.inject_trainable_lora(unet, r=2, scale=0.7)
    tparams, tnames = L.inject_trainable_lora(clip, target_replace_module={"CLIPAttention"}, r=3)
    assert names == info["inject_names_unet"] and tnames == info["inject_names_text"]
    assert len(params) == info["n_param_groups_unet"] == 2 * len(names)
    st = _npz("mini_ref_state.npz")
    for tag, model, tgt in (("unet", unet, L.DEFAULT_TARGET_REPLACE), ("text_encoder", clip, {"CLIPAttention"})):
        for i, (up, down) in enumerate(L.extract_lora_ups_down(model, tgt)):
            up.weight.data = torch.from_numpy(st[f"{tag}_{i}_up"])
            down.weight.data = torch.from_numpy(st[f"{tag}_{i}_down"])
    return unet, clip, st


def test_injection_contract_and_names():
    info = json.load(open(os.path.join(G, "mini_ref_info.json")))
    unet = H.build_tree(H.toy_unet_spec())
    before = {n: p for n, p in unet.named_parameters()}
    groups, names = L.inject_trainable_lora(unet, r=2)
    assert names == info["inject_names_unet"]
    first = unet.down_blocks._modules["0"].attentions._modules["0"].transformer_blocks._modules["0"].attn1.to_q
    assert isinstance(first, L.LoraInjectedLinear) and first.dropout.p == 0.0  # inject default dropout 0.0
    assert first.linear.weight is before["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"]  # aliased
    g0, g1 = list(groups[0]), list(groups[1])
    assert g0[0] is first.lora_up.weight and g1[0] is first.lora_down.weight  # up first, then down
    assert first.lora_up.weight.requires_grad and first.lora_down.weight.requires_grad
    # untouched: FeedForward.net[2], proj_in (quirk 4)
    assert type(unet.down_blocks._modules["0"].attentions._modules["0"].proj_in) is nn.Linear
    unet2 = H.build_tree(H.toy_unet_spec())
    _, names_ext = L.inject_trainable_lora_extended(unet2, r=2)
    assert names_ext == info["inject_names_unet_extended"]
    kinds = [type(m).__name__ for _, _, m in L._find_modules_v2(
        unet2, L.UNET_EXTENDED_TARGET_REPLACE, search_class=[L.LoraInjectedLinear, L.LoraInjectedConv2d])]
    assert kinds == info["extended_kinds"]
    conv1 = unet2.down_blocks._modules["0"].resnets._modules["0"].conv1
    assert conv1.dropout.p == 0.1  # extended inject keeps the ctor default (quirk 1)
    with pytest.raises(ValueError, match="No lora injected."):
        L.extract_lora_ups_down(H.build_tree(H.toy_unet_spec()))


def test_safetensors_bytes_match_reference(tmp_path):
    from safetensors import safe_open

    info = json.load(open(os.path.join(G, "mini_ref_info.json")))
    unet, clip, st = _toy_models(info)
    out = str(tmp_path / "mine.safetensors")
    quiet(L.save_safeloras_with_embeds, {"unet": (unet, L.DEFAULT_TARGET_REPLACE), "text_encoder": (clip, {"CLIPAttention"})},
          {"<s1>": torch.from_numpy(st["embed_s1"]), "<s2>": torch.from_numpy(st["embed_s2"])}, out)
    a, b = safe_open(out, framework="pt"), safe_open(os.path.join(G, "mini_ref.safetensors"), framework="pt")
    assert list(a.keys()) == list(b.keys())
    ma, mb = a.metadata(), b.metadata()
    assert set(ma) == set(mb)
    for k in ma:
        if k in ("unet", "text_encoder"):
            assert set(json.loads(ma[k])) == set(json.loads(mb[k]))
        else:
            assert ma[k] == mb[k]
    for k in a.keys():
        ta, tb = a.get_tensor(k), b.get_tensor(k)
        assert ta.dtype == tb.dtype and ta.shape == tb.shape and torch.equal(ta, tb), k
    # .pt format: fp16 list [up0, down0, ...], scale NOT folded (quirk 2)
    quiet(L.save_lora_weight, unet, str(tmp_path / "mine.pt"))
    mine, ref = torch.load(str(tmp_path / "mine.pt")), torch.load(os.path.join(G, "mini_ref.pt"))
    assert len(mine) == len(ref) and all(torch.equal(x, y) and x.dtype == torch.float16 for x, y in zip(mine, ref))


def test_parse_and_patch_roundtrip():
    info = json.load(open(os.path.join(G, "mini_ref_info.json")))
    path = os.path.join(G, "mini_ref.safetensors")
    parsed = L.load_safeloras(path)
    for name, (weights, ranks, targets) in parsed.items():
        assert ranks == info["parsed"][name]["ranks"] and sorted(targets) == info["parsed"][name]["targets"]
        assert [list(w.shape) for w in weights] == info["parsed"][name]["shapes"]
        assert all(isinstance(w, nn.Parameter) for w in weights)
    assert sorted(L.load_safeloras_embeds(path)) == info["embeds"]
    both = L.load_safeloras_both(path)
    assert set(both[0]) == {"unet", "text_encoder"} and set(both[1]) == {"<s1>", "<s2>"}

    src_unet, src_clip, _ = _toy_models(info)
    torch.manual_seed(7)  # same frozen weights as _toy_models
    unet, clip = H.build_tree(H.toy_unet_spec()), H.build_tree(H.toy_clip_spec())

    class Pipe:
        pass

    pipe = Pipe()
    pipe.unet, pipe.text_encoder = unet, clip
    from safetensors import safe_open
    out = io.StringIO()
    with redirect_stdout(out):
        L.monkeypatch_or_replace_safeloras(pipe, safe_open(path, framework="pt", device="cpu"))
        pipe2 = Pipe()
        pipe2.unet = H.build_tree(H.toy_unet_spec())
        L.monkeypatch_or_replace_safeloras(pipe2, safe_open(path, framework="pt", device="cpu"))
    assert "No model provided for text_encoder, contained in Lora" in out.getvalue()
    a = src_unet.mid_block.attentions._modules["0"].transformer_blocks._modules["0"].attn2.to_v
    b = unet.mid_block.attentions._modules["0"].transformer_blocks._modules["0"].attn2.to_v
    assert isinstance(b, L.LoraInjectedLinear) and b.scale == 1.0 and b.dropout.p == 0.1 and b.r == 2
    x = torch.randn(5, 8)
    a.eval(), b.eval()
    np.testing.assert_allclose(H.t2n(b(x)), H.t2n(a(x)), rtol=2e-3, atol=2e-3)  # fp16 storage of up*scale, down
    # replace again with explicit rank list (r popped per site)
    flat = [p.data.clone() for p in parsed["text_encoder"][0]]
    L.monkeypatch_or_replace_lora(clip, flat, {"CLIPAttention"}, r=[3] * 8)
    assert flat == [] and clip.encoder.layers._modules["1"].self_attn.out_proj.r == 3


def test_remove_add_scale_diag():
    m = H.build_tree(H.attn_spec())
    w = m.to_q.weight
    L.inject_trainable_lora(m, r=2)
    d = _npz("add_lora_case.npz")
    for i, (up, down) in enumerate(L.extract_lora_ups_down(m)):
        up.weight.data = torch.from_numpy(d[f"cur{2 * i}"])
        down.weight.data = torch.from_numpy(d[f"cur{2 * i + 1}"])
    L.monkeypatch_add_lora(m, [torch.from_numpy(d[f"new{i}"]) for i in range(8)], alpha=0.3, beta=0.9)
    for i, (up, down) in enumerate(L.extract_lora_ups_down(m)):
        np.testing.assert_allclose(H.t2n(up.weight), d[f"after{2 * i}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(H.t2n(down.weight), d[f"after{2 * i + 1}"], rtol=1e-6, atol=1e-7)
    L.tune_lora_scale(m, 0.25)
    assert all(a.scale == 0.25 for a in m.modules() if isinstance(a, L.LoraInjectedLinear))
    L.set_lora_diag(m, torch.tensor([2.0, 0.0]))
    assert isinstance(m.to_k.selector, nn.Linear) and torch.equal(m.to_k.selector.weight, torch.diag(torch.tensor([2.0, 0.0])))
    x = torch.randn(3, 8)
    want = m.to_k.linear(x) + (m.to_k.lora_down(x) * torch.tensor([2.0, 0.0])) @ m.to_k.lora_up.weight.t() * 0.25
    m.eval()
    np.testing.assert_allclose(H.t2n(m.to_k(x)), H.t2n(want), rtol=1e-5, atol=1e-6)
    moved = L.inspect_lora(m)
    assert set(moved) == {"to_q", "to_k", "to_v", "to_out.0"} and all(len(v) == 1 for v in moved.values())
    L.monkeypatch_remove_lora(m)
    assert type(m.to_q) is nn.Linear and m.to_q.weight is w and type(m.to_out._modules["0"]) is nn.Linear


def test_collapse_cpu_matches_reference_vectors():
    d = _npz("collapse_cases.npz")
    for c in json.load(open(os.path.join(G, "collapse_cases.json"))):
        tag = c["tag"]
        wdt, abdt = H.TORCH_DT[c["w_dtype"]], H.TORCH_DT[c["ab_dtype"]]
        if c["kind"] == "linear":
            m, root = L.LoraInjectedLinear(40, 24, False, r=4, scale=3.0), H.named_class("CrossAttention")()
            frozen = m.linear
        else:
            m, root = L.LoraInjectedConv2d(8, 12, 3, 1, 1, r=4, scale=3.0), H.named_class("ResnetBlock2D")()
            frozen = m.conv
        frozen.weight.data = torch.from_numpy(d[f"{tag}_W"]).to(wdt)
        m.lora_up.weight.data = torch.from_numpy(d[f"{tag}_up"]).to(abdt)
        m.lora_down.weight.data = torch.from_numpy(d[f"{tag}_down"]).to(abdt)
        root.add_module("site", m)
        old = frozen.weight
        quiet(L.collapse_lora, root, c["alpha"])
        assert frozen.weight is not old and isinstance(frozen.weight, nn.Parameter) and frozen.weight.dtype == wdt
        assert torch.equal(frozen.weight.float(), torch.from_numpy(d[f"{tag}_out"])), tag


def test_path_helpers_and_constants():
    assert L._text_lora_path("a/b.c.pt") == "a/b.c.text_encoder.pt" and L._ti_lora_path("x.pt") == "x.ti.pt"
    with pytest.raises(AssertionError):
        L._ti_lora_path("x.safetensors")
    assert L.UNET_DEFAULT_TARGET_REPLACE == {"CrossAttention", "Attention", "GEGLU"}
    assert L.UNET_EXTENDED_TARGET_REPLACE == L.UNET_DEFAULT_TARGET_REPLACE | {"ResnetBlock2D"}
    assert L.TEXT_ENCODER_DEFAULT_TARGET_REPLACE == {"CLIPAttention"} and L.EMBED_FLAG == "<embed>"
    assert L.DEFAULT_TARGET_REPLACE is L.UNET_DEFAULT_TARGET_REPLACE and L.safetensors_available
    assert L.monkeypatch_lora is L.monkeypatch_or_replace_lora


@pytest.mark.skipif(not H.reference_available(), reason="live reference not mounted")
def test_differential_against_live_reference(tmp_path):
    """Random trees: injection order, saved bytes and forward values identical to the real reference."""
    ref = H.load_reference()
    rng = np.random.default_rng(0)

    def rand_spec(depth):
        kids = []
        for i in range(int(rng.integers(1, 4))):
            roll = rng.random()
            if depth >= 3 or roll < 0.35:
                kids.append([f"l{i}", {"cls": "Linear", "kind": "linear"}])
            elif roll < 0.45:
                kids.append([f"c{i}", {"cls": "Conv2d", "kind": "conv", "k": 1}])
            else:
                cls = ["CrossAttention", "GEGLU", "Attention", "ResnetBlock2D", "Other", "ModuleList"][int(rng.integers(0, 6))]
                kids.append([f"m{i}", {"cls": cls, "children": rand_spec(depth + 1)["children"]}])
        return {"cls": "Root", "children": kids}

    for trial in range(12):
        spec = rand_spec(0)
        torch.manual_seed(trial)
        a = H.build_tree(spec)
        b = copy.deepcopy(a)
        ext = trial % 2 == 1
        if ext:
            torch.manual_seed(99); _, na = ref.inject_trainable_lora_extended(a, r=2)
            torch.manual_seed(99); _, nb = L.inject_trainable_lora_extended(b, r=2)
            tgt = ref.UNET_EXTENDED_TARGET_REPLACE
        else:
            torch.manual_seed(99); _, na = ref.inject_trainable_lora(a, r=2, scale=0.5)
            torch.manual_seed(99); _, nb = L.inject_trainable_lora(b, r=2, scale=0.5)
            tgt = ref.DEFAULT_TARGET_REPLACE
        assert na == nb
        if not na:
            continue
        for (ua, da), (ub, db) in zip(ref.extract_lora_ups_down(a, tgt), L.extract_lora_ups_down(b, tgt)):
            assert torch.equal(da.weight, db.weight)  # same RNG stream consumption
            ua.weight.data.normal_(0, 0.1)
            ub.weight.data.copy_(ua.weight.data)
        pa, pb = str(tmp_path / f"a{trial}.safetensors"), str(tmp_path / f"b{trial}.safetensors")
        quiet(ref.save_safeloras, {"unet": (a, tgt)}, pa)
        quiet(L.save_safeloras, {"unet": (b, tgt)}, pb)
        fa, fb = ref.load_safeloras(pa), L.load_safeloras(pb)
        assert fa["unet"][1] == fb["unet"][1]
        assert all(torch.equal(x, y) for x, y in zip(fa["unet"][0], fb["unet"][0]))
        quiet(ref.collapse_lora, a, 0.8)
        quiet(L.collapse_lora, b, 0.8)
        for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
            assert n1 == n2 and torch.equal(p1, p2), n1
