"""Pin the oracle (oracle/lora_numpy.py, oracle/torch_ref.py) against vectors produced by the REAL
reference (scripts/make_golden.py ran /root/reference/lora_diffusion/lora.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import lora_numpy as O
from oracle import torch_ref as TR
from tests import helpers as H

G = H.GOLDEN


def _npz(name):
    return dict(np.load(os.path.join(G, name)))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_linear_forward_backward_matches_reference(tag):
    d = _npz("linear_cases.npz")
    sel = d.get(f"{tag}_sel")
    y, _ = O.lora_linear_forward(d[f"{tag}_x"], d[f"{tag}_W"], d.get(f"{tag}_b"), d[f"{tag}_down"], d[f"{tag}_up"],
                                 float(d[f"{tag}_scale"]), sel)
    np.testing.assert_allclose(y, d[f"{tag}_y"], rtol=1e-5, atol=1e-5)
    dx, dd, du, _, _ = O.lora_linear_backward(d[f"{tag}_gy"], d[f"{tag}_x"], d[f"{tag}_W"], d[f"{tag}_down"],
                                              d[f"{tag}_up"], float(d[f"{tag}_scale"]), sel)
    np.testing.assert_allclose(dx, d[f"{tag}_dx"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dd, d[f"{tag}_ddown"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(du, d[f"{tag}_dup"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_torch_ref_linear_matches_reference(tag):
    d = _npz("linear_cases.npz")
    t = lambda k: torch.from_numpy(d[k]) if k in d else None
    x = t(f"{tag}_x").requires_grad_(True)
    down, up = t(f"{tag}_down").requires_grad_(True), t(f"{tag}_up").requires_grad_(True)
    y = TR.linear_adapter_forward(x, t(f"{tag}_W"), t(f"{tag}_b"), down, up, float(d[f"{tag}_scale"]), t(f"{tag}_sel"))
    (y * t(f"{tag}_gy")).sum().backward()
    np.testing.assert_allclose(H.t2n(y), d[f"{tag}_y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(H.t2n(x.grad), d[f"{tag}_dx"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H.t2n(down.grad), d[f"{tag}_ddown"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(H.t2n(up.grad), d[f"{tag}_dup"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_conv_forward_matches_reference(tag):
    d = _npz("conv_cases.npz")
    k, s, p, r = (int(v) for v in d[f"{tag}_geom"])
    y, _ = O.lora_conv2d_forward(d[f"{tag}_x"], d[f"{tag}_W"], d[f"{tag}_b"], d[f"{tag}_down"], d[f"{tag}_up"],
                                 float(d[f"{tag}_scale"]), (s, s), (p, p))
    np.testing.assert_allclose(y, d[f"{tag}_y"], rtol=1e-4, atol=1e-4)
    # torch restatement incl. gradients
    t = lambda key: torch.from_numpy(d[key])
    x = t(f"{tag}_x").requires_grad_(True)
    down, up = t(f"{tag}_down").requires_grad_(True), t(f"{tag}_up").requires_grad_(True)
    y2 = TR.conv_adapter_forward(x, t(f"{tag}_W"), t(f"{tag}_b"), down, up, float(d[f"{tag}_scale"]), (s, s), (p, p),
                                 (1, 1), 1)
    (y2 * t(f"{tag}_gy")).sum().backward()
    np.testing.assert_allclose(H.t2n(y2), d[f"{tag}_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(H.t2n(x.grad), d[f"{tag}_dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(H.t2n(down.grad), d[f"{tag}_ddown"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(H.t2n(up.grad), d[f"{tag}_dup"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("fn,tag", [("conv_cases.npz", t) for t in "abc"] +
                         [("conv_native_cases.npz", t) for t in ("n1", "n2", "n3", "n4", "n5")])
def test_conv_backward_oracle_matches_reference(fn, tag):
    """numpy restatement of the conv adapter's autograd vs gradients produced by the reference itself."""
    d = _npz(fn)
    k, s, p, r = (int(v) for v in d[f"{tag}_geom"])
    sc = float(d[f"{tag}_scale"])
    y, _ = O.lora_conv2d_forward(d[f"{tag}_x"], d[f"{tag}_W"], d[f"{tag}_b"], d[f"{tag}_down"], d[f"{tag}_up"], sc,
                                 (s, s), (p, p))
    np.testing.assert_allclose(y, d[f"{tag}_y"], rtol=1e-4, atol=1e-4)
    dx, ddown, dup = O.lora_conv2d_backward(d[f"{tag}_gy"], d[f"{tag}_x"], d[f"{tag}_W"], d[f"{tag}_down"],
                                            d[f"{tag}_up"], sc, (s, s), (p, p))
    np.testing.assert_allclose(dx, d[f"{tag}_dx"], rtol=1e-4, atol=1e-4 * np.abs(d[f"{tag}_dx"]).max())
    np.testing.assert_allclose(ddown, d[f"{tag}_ddown"], rtol=1e-4, atol=1e-4 * np.abs(d[f"{tag}_ddown"]).max())
    np.testing.assert_allclose(dup, d[f"{tag}_dup"], rtol=1e-4, atol=1e-4 * np.abs(d[f"{tag}_dup"]).max())


def test_conv_selector_oracle_is_consistent_with_torch():
    """The reference cannot run a conv selector (it installs a 2-D Conv2d weight, lora.py:151); the oracle's
    selector maths is pinned against torch autograd of the intended 1x1 rank-mixing conv instead."""
    g = torch.Generator().manual_seed(3)
    B, Ci, Co, Hh, Ww, r = 2, 6, 5, 4, 8, 3
    x = torch.randn(B, Ci, Hh, Ww, generator=g, requires_grad=True)
    W, down = torch.randn(Co, Ci, 3, 3, generator=g) * 0.2, (torch.randn(r, Ci, 3, 3, generator=g) * 0.3).requires_grad_(True)
    up, sel = (torch.randn(Co, r, 1, 1, generator=g) * 0.4).requires_grad_(True), torch.randn(r, r, generator=g)
    gy = torch.randn(B, Co, Hh, Ww, generator=g)
    F = torch.nn.functional
    y = F.conv2d(x, W, None, 1, 1) + 0.8 * F.conv2d(F.conv2d(F.conv2d(x, down, None, 1, 1), sel.reshape(r, r, 1, 1)), up)
    (y * gy).sum().backward()
    yo, _ = O.lora_conv2d_forward(H.t2n(x), H.t2n(W), None, H.t2n(down), H.t2n(up), 0.8, (1, 1), (1, 1), selector=H.t2n(sel))
    dx, ddown, dup = O.lora_conv2d_backward(H.t2n(gy), H.t2n(x), H.t2n(W), H.t2n(down), H.t2n(up), 0.8, (1, 1), (1, 1),
                                            selector=H.t2n(sel))
    np.testing.assert_allclose(yo, H.t2n(y), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dx, H.t2n(x.grad), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ddown, H.t2n(down.grad), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dup, H.t2n(up.grad), rtol=1e-4, atol=1e-5)


def test_collapse_matches_reference_per_dtype():
    d = _npz("collapse_cases.npz")
    cases = json.load(open(os.path.join(G, "collapse_cases.json")))
    for c in cases:
        tag = c["tag"]
        out = O.collapse(d[f"{tag}_W"], d[f"{tag}_up"], d[f"{tag}_down"], c["alpha"], c["w_dtype"], c["ab_dtype"])
        ref = d[f"{tag}_out"]
        if c["w_dtype"] == "f32":
            np.testing.assert_allclose(out, ref, rtol=0, atol=2e-7, err_msg=tag)
        else:
            # same rounding sequence: identical up to last-place ties in the r-term f32 dot product
            ulp = np.abs(ref) * (2.0 ** -7 if c["w_dtype"] == "bf16" else 2.0 ** -10)
            assert np.all(np.abs(out - ref) <= ulp + 1e-30), tag
            assert (out != ref).mean() < 0.01, f"{tag}: {(out != ref).mean():.4f} of elements differ"


def test_traversal_order_matches_reference():
    data = json.load(open(os.path.join(G, "traversal_cases.json")))
    for case in data["cases"]:
        root = O.Node.from_spec(H.oracle_spec(data["trees"][case["tree"]]))
        got = O.find_modules(root, case["ancestors"], case["kinds"])
        assert got == case["paths"], (case["tree"], case["ancestors"], case["kinds"])


def test_file_layout_and_parse_order_match_reference():
    from safetensors import safe_open

    f = safe_open(os.path.join(G, "mini_ref.safetensors"), framework="np")
    st = _npz("mini_ref_state.npz")
    info = json.load(open(os.path.join(G, "mini_ref_info.json")))
    models = {}
    for name, scale, targets in (("unet", info["scale_unet"], ["CrossAttention", "Attention", "GEGLU"]),
                                 ("text_encoder", 1.0, ["CLIPAttention"])):
        pairs, i = [], 0
        while f"{name}_{i}_up" in st:
            up, down = O.realize_as_lora(st[f"{name}_{i}_up"], st[f"{name}_{i}_down"], scale)
            pairs.append((up.astype(np.float16), down.astype(np.float16)))
            i += 1
        models[name] = (pairs, targets)
    weights, meta = O.safeloras_layout(models, {"<s1>": st["embed_s1"], "<s2>": st["embed_s2"]})
    assert set(weights) == set(f.keys())
    fmeta = f.metadata()
    assert set(meta) == set(fmeta)
    for k, v in meta.items():
        if k in ("unet", "text_encoder"):
            assert set(json.loads(v)) == set(json.loads(fmeta[k]))
        else:
            assert v == fmeta[k]
    for k, w in weights.items():
        got = f.get_tensor(k)
        assert got.dtype == w.dtype and got.shape == w.shape
        assert got.tobytes() == np.ascontiguousarray(w).tobytes(), k
    parsed = O.parse_order(list(f.keys()), fmeta)
    for name, (order, ranks, targets) in parsed.items():
        assert ranks == info["parsed"][name]["ranks"]
        assert sorted(targets) == info["parsed"][name]["targets"]
        assert [list(f.get_tensor(k).shape) for k in order] == info["parsed"][name]["shapes"]


def test_example_loras_manifest_parse_order():
    man = json.load(open(os.path.join(G, "example_loras_manifest.json")))
    for fn, ent in man.items():
        parsed = O.parse_order(ent["keys_in_file_order"], ent["metadata"])
        for name, (order, ranks, targets) in parsed.items():
            assert len(order) == ent["parsed"][name]["n"]
            assert ranks == ent["parsed"][name]["ranks"]
            assert sorted(targets) == ent["parsed"][name]["targets"]
            assert order[0].endswith(":0:up") and order[1].endswith(":0:down")


def test_add_blend_matches_reference():
    d = _npz("add_lora_case.npz")
    i = 0
    while f"cur{i}" in d:
        np.testing.assert_allclose(O.add_lora_blend(d[f"cur{i}"], d[f"new{i}"], 0.3, 0.9), d[f"after{i}"], rtol=1e-6,
                                   atol=1e-7)
        i += 1
    assert i == 8


def test_clip_and_adamw_match_torch():
    d = _npz("optimizer_case.npz")
    p = d["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    lr = np.concatenate([np.full(700, 1e-2), np.full(300, 5e-3)]).astype(np.float32)
    for step in (1, 2, 3):
        (g,), total = O.clip_grad_norm([d[f"g{step}"]], 1.0)
        assert abs(total - float(d[f"norm{step}"])) <= 1e-5 * max(1.0, total)
        pa, ma, va = O.adamw_step(p[:700], g[:700], m[:700], v[:700], step, lr=1e-2)
        pb, mb, vb = O.adamw_step(p[700:], g[700:], m[700:], v[700:], step, lr=5e-3)
        p, m, v = np.concatenate([pa, pb]), np.concatenate([ma, mb]), np.concatenate([va, vb])
        np.testing.assert_allclose(p, d[f"p{step}"], rtol=2e-6, atol=2e-7)


def test_ddpm_schedule_and_loss():
    a = O.ddpm_alphas_cumprod()
    assert a.shape == (1000,) and abs(a[0] - (1 - 0.00085)) < 1e-6 and 0.004 < a[-1] < 0.005
    rng = np.random.default_rng(0)
    x, n = rng.standard_normal((2, 4, 8, 8)).astype(np.float32), rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    z = O.add_noise(x, n, [0, 999], a)
    np.testing.assert_allclose(z[0], np.sqrt(a[0]) * x[0] + np.sqrt(1 - a[0]) * n[0], rtol=1e-6)
    assert abs(O.dreambooth_loss(x, n) - float(((x - n) ** 2).mean())) < 1e-6
    lp = O.dreambooth_loss(x, n, True, 0.5)
    assert abs(lp - (float(((x[:1] - n[:1]) ** 2).mean()) + 0.5 * float(((x[1:] - n[1:]) ** 2).mean()))) < 1e-5


def test_host_pass_oracles_match_torch_vectors():
    """GroupNorm(+SiLU), LayerNorm and the GEGLU gate (the ATen calls of the UNet blocks between the adapter sites):
    numpy restatement vs the torch CPU vectors of scripts/make_golden.py::hostops_cases."""
    d = np.load(os.path.join(H.GOLDEN, "hostops_cases.npz"))
    for i in range(3):
        groups, act = (int(v) for v in d[f"gn{i}_meta"])
        y, cache = O.group_norm_act(d[f"gn{i}_x"], groups, d[f"gn{i}_w"], d[f"gn{i}_b"], 1e-5, bool(act))
        np.testing.assert_allclose(y, d[f"gn{i}_y"], rtol=2e-5, atol=2e-5)
        dx = O.group_norm_act_backward(d[f"gn{i}_go"], cache)
        np.testing.assert_allclose(dx, d[f"gn{i}_dx"], rtol=2e-4, atol=2e-5)
    for i in range(2):
        y, cache = O.layer_norm(d[f"ln{i}_x"], d[f"ln{i}_w"], d[f"ln{i}_b"])
        np.testing.assert_allclose(y, d[f"ln{i}_y"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(O.layer_norm_backward(d[f"ln{i}_go"], cache), d[f"ln{i}_dx"], rtol=2e-4, atol=2e-5)
    for i in range(2):
        np.testing.assert_allclose(O.geglu(d[f"gg{i}_y"]), d[f"gg{i}_out"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(O.geglu_backward(d[f"gg{i}_y"], d[f"gg{i}_go"]), d[f"gg{i}_dy"], rtol=2e-5, atol=2e-6)
