"""Parity of the HIP path (through the C-ABI) with the oracle and the reference-generated fixtures.

Tolerances (stated per the north star: fp16/bf16 tolerance for floats, bit-level for layout):
  * f32 data, f32 accumulation: |err| <= 2e-5 * (sum of |terms|)   [summation-order only]
  * bf16 / f16 outputs: one rounding of the exact result -> rel 2^-8 / 2^-11 of the value, plus the above
  * merge in reference rounding mode: <= 1 ulp of the weight dtype, and < 1 % of elements differ at all
"""
import json
import os

import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C, ops
from oracle import lora_numpy as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = H.GOLDEN
DEV = "cuda:0"
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
EPS = {"f32": 0.0, "bf16": 2.0 ** -8, "f16": 2.0 ** -11}


def _npz(name):
    return dict(np.load(os.path.join(G, name)))


def n(t):
    return t.detach().float().cpu().numpy()


def rnd(shape, dt, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DT[dt]).to(DEV)


def close(got, want, absref, dt_out="f32", k=2e-5, msg=""):
    """|got-want| <= k*absref + eps_out*|want| (absref = sum of |terms| of the reduction)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    tol = k * np.asarray(absref, dtype=np.float64) + EPS[dt_out] * np.abs(want) + 1e-30
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{msg}: {bad.sum()} of {bad.size} outside tolerance; worst {np.abs(got - want).max():.3e}"


SHAPES = [(512, 320, 4), (300, 768, 4), (64, 1280, 8), (130, 2560, 4), (33, 640, 16), (17, 320, 1), (40, 4096, 64),
          (9, 77, 3), (1, 8, 1), (257, 1288, 5)]


@pytest.mark.parametrize("M,K,r", SHAPES)
@pytest.mark.parametrize("dt", ["bf16", "f32", "f16"])
@pytest.mark.parametrize("layout", [_C.FACTOR_RK, _C.FACTOR_KR])
def test_rowdot_matches_oracle(M, K, r, dt, layout):
    x = rnd((M, K), dt, seed=1)
    f = rnd((r, K) if layout == _C.FACTOR_RK else (K, r), "f32", 0.3, seed=2)
    t = _C.rowdot(x, f, layout, 0.7)
    fr = n(f) if layout == _C.FACTOR_RK else n(f).T
    want = 0.7 * (n(x).astype(np.float64) @ fr.T.astype(np.float64))
    absref = 0.7 * (np.abs(n(x)) @ np.abs(fr.T))
    close(n(t), want, absref, msg=f"rowdot {M}x{K} r{r} {dt}")


def test_rowdot_selector_and_strided_rows():
    M, K, r = 100, 640, 4
    big = rnd((M, K + 64), "bf16", seed=3)
    x = big[:, :K]  # ldx = K+64
    f = rnd((r, K), "f32", 0.3, seed=4)
    sel = rnd((r, r), "f32", seed=5)
    base = n(x).astype(np.float64) @ n(f).T.astype(np.float64)
    absref = (np.abs(n(x)) @ np.abs(n(f)).T) @ np.abs(n(sel)).T
    close(n(_C.rowdot(x, f, _C.FACTOR_RK, 1.0, sel, False)), base @ n(sel).T, absref, msg="sel fwd")
    close(n(_C.rowdot(x, f, _C.FACTOR_RK, 1.0, sel, True)), base @ n(sel), (np.abs(n(x)) @ np.abs(n(f)).T) @ np.abs(n(sel)),
          msg="sel bwd")
    # unaligned base pointer -> scalar-lane kernel
    xo = big.view(-1)[1:1 + M * K].view(M, K)
    close(n(_C.rowdot(xo, f, _C.FACTOR_RK)), n(xo).astype(np.float64) @ n(f).T, np.abs(n(xo)) @ np.abs(n(f)).T, msg="unaligned")


@pytest.mark.parametrize("M,N,r", [(512, 320, 4), (300, 2560, 4), (64, 10240, 4), (33, 640, 16), (129, 1280, 8),
                                   (40, 512, 64), (9, 77, 3), (1, 8, 1), (1000, 1288, 2)])
@pytest.mark.parametrize("dt", ["bf16", "f32", "f16"])
@pytest.mark.parametrize("layout", [_C.FACTOR_RK, _C.FACTOR_KR])
def test_rank_update_matches_oracle(M, N, r, dt, layout):
    y0 = rnd((M, N), dt, seed=6)
    t = rnd((M, r), "f32", seed=7)
    f = rnd((r, N) if layout == _C.FACTOR_RK else (N, r), "f32", 0.3, seed=8)
    fr = n(f) if layout == _C.FACTOR_RK else n(f).T
    want = n(y0).astype(np.float64) + 0.6 * (n(t).astype(np.float64) @ fr)
    absref = np.abs(n(y0)) + 0.6 * (np.abs(n(t)) @ np.abs(fr))
    y = y0.clone()
    _C.rank_update_(y, t, f, layout, 0.6)
    close(n(y), want, absref, dt, msg=f"rank_update {M}x{N} r{r} {dt}")


@pytest.mark.parametrize("M,K,r", [(512, 320, 4), (300, 768, 4), (1000, 2560, 4), (70, 4104, 8), (33, 640, 16), (64, 320, 40),
                                   (9, 77, 3), (1, 8, 1), (4096, 1280, 4)])
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("layout", [_C.FACTOR_RK, _C.FACTOR_KR])
def test_colreduce_matches_oracle(M, K, r, dt, layout):
    x = rnd((M, K), dt, seed=9)
    t = rnd((M, r), "f32", seed=10)
    want = 0.5 * (n(t).T.astype(np.float64) @ n(x).astype(np.float64))
    absref = 0.5 * (np.abs(n(t)).T @ np.abs(n(x)))
    d = _C.colreduce(x, t, layout, 0.5)
    got = n(d) if layout == _C.FACTOR_RK else n(d).T
    close(got, want, absref, msg=f"colreduce {M}x{K} r{r}")
    # accumulate into an existing buffer (beta = 1): the trainer's flat-grad path
    prev = rnd(tuple(d.shape), "f32", seed=11)
    acc = prev.clone()
    _C.colreduce(x, t, layout, 0.5, out=acc, beta=1.0)
    got2 = n(acc) if layout == _C.FACTOR_RK else n(acc).T
    pv = n(prev) if layout == _C.FACTOR_RK else n(prev).T
    close(got2, want + pv, absref + np.abs(pv), msg="colreduce beta=1")


def test_dropout_mask_is_shared_by_forward_and_backward_kernels():
    M, N, p = 200, 640, 0.25
    ones_t = torch.ones(M, 1, device=DEV)
    ones_f = torch.ones(1, N, device=DEV)
    y = torch.zeros(M, N, device=DEV)
    _C.rank_update_(y, ones_t, ones_f, _C.FACTOR_RK, 1.0, p, 1234, 77)
    mask = n(y)
    keep = 1.0 / (1.0 - p)
    assert set(np.unique(mask)).issubset({0.0, np.float32(keep)})
    rate = (mask > 0).mean()
    assert abs(rate - (1 - p)) < 4 * np.sqrt(p * (1 - p) / mask.size) + 1e-3, rate  # 4 sigma + 16-bit threshold step
    y2 = torch.zeros(M, N, device=DEV)
    _C.rank_update_(y2, ones_t, ones_f, _C.FACTOR_RK, 1.0, p, 1234, 77)
    assert torch.equal(y, y2)  # deterministic in (seed, offset)
    y3 = torch.zeros(M, N, device=DEV)
    _C.rank_update_(y3, ones_t, ones_f, _C.FACTOR_RK, 1.0, p, 1234, 78)
    assert not torch.equal(y, y3)
    ones_x = torch.ones(M, N, device=DEV)
    rs = _C.rowdot(ones_x, ones_f, _C.FACTOR_RK, 1.0, None, False, p, 1234, 77)
    np.testing.assert_allclose(n(rs)[:, 0], mask.sum(1), rtol=1e-5)
    cs = _C.colreduce(ones_x, ones_t, _C.FACTOR_RK, 1.0, dropout_p=p, seed=1234, offset=77)
    np.testing.assert_allclose(n(cs)[0], mask.sum(0), rtol=1e-5)
    # bf16 activations use the same element indexing
    yb = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
    _C.rank_update_(yb, ones_t, ones_f, _C.FACTOR_RK, 1.0, p, 1234, 77)
    assert np.array_equal(n(yb) > 0, mask > 0)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_linear_module_matches_reference_vectors(tag):
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
    m.to(DEV)
    x = torch.from_numpy(d[f"{tag}_x"]).to(DEV).requires_grad_(True)
    y = m(x)
    (y * torch.from_numpy(d[f"{tag}_gy"]).to(DEV)).sum().backward()
    for name, got, want in (("y", y, d[f"{tag}_y"]), ("dx", x.grad, d[f"{tag}_dx"]),
                            ("ddown", m.lora_down.weight.grad, d[f"{tag}_ddown"]),
                            ("dup", m.lora_up.weight.grad, d[f"{tag}_dup"])):
        np.testing.assert_allclose(n(got), want, rtol=2e-4, atol=2e-4 * np.abs(want).max(), err_msg=f"{tag} {name}")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_linear_module_low_precision_and_autocast(dt):
    M, K, N, r, s = 384, 320, 640, 4, 0.8
    torch.manual_seed(0)
    m = L.LoraInjectedLinear(K, N, True, r=r, dropout_p=0.0, scale=s)
    m.lora_up.weight.data.normal_(0, 0.05)
    x32 = torch.randn(2, M // 2, K)
    gy32 = torch.randn(2, M // 2, N)
    # (1) resident low-precision frozen weights, f32 factors
    mb = L.LoraInjectedLinear(K, N, True, r=r, dropout_p=0.0, scale=s)
    mb.load_state_dict(m.state_dict())
    mb.linear.to(DT[dt])
    mb.to(DEV)
    xb = x32.to(DT[dt]).to(DEV).requires_grad_(True)
    yb = mb(xb)
    assert yb.dtype == DT[dt] and yb.shape == (2, M // 2, N)
    (yb.float() * gy32.to(DEV)).sum().backward()
    W, b = n(mb.linear.weight), n(mb.linear.bias)
    X, A, B = n(xb).reshape(M, K), n(mb.lora_down.weight), n(mb.lora_up.weight)
    Gy = n(gy32.to(DT[dt])).reshape(M, N)  # autograd hands the kernel a dt-rounded G
    yo, _ = O.lora_linear_forward(X, W, b, A, B, s)
    dxo, ddo, duo, _, _ = O.lora_linear_backward(Gy, X, W, A, B, s)
    e = 4 * EPS[dt]
    np.testing.assert_allclose(n(yb).reshape(M, N), yo, rtol=e, atol=e * np.abs(yo).max())
    np.testing.assert_allclose(n(xb.grad).reshape(M, K), dxo, rtol=e, atol=e * np.abs(dxo).max())
    np.testing.assert_allclose(n(mb.lora_down.weight.grad), ddo, rtol=1e-3, atol=1e-3 * np.abs(ddo).max())
    np.testing.assert_allclose(n(mb.lora_up.weight.grad), duo, rtol=1e-3, atol=1e-3 * np.abs(duo).max())
    assert mb.lora_up.weight.grad.dtype == torch.float32
    # (2) reference-style autocast: f32 weights + activations, compute dtype from the autocast context
    ma = L.LoraInjectedLinear(K, N, True, r=r, dropout_p=0.0, scale=s)
    ma.load_state_dict(m.state_dict())
    ma.to(DEV)
    xa = x32.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=DT[dt]):
        ya = ma(xa)
    assert ya.dtype == DT[dt]
    (ya.float() * gy32.to(DEV)).sum().backward()
    yo2, _ = O.lora_linear_forward(O.round_to(n(xa).reshape(M, K), dt), O.round_to(n(ma.linear.weight), dt),
                                   O.round_to(n(ma.linear.bias), dt), A, B, s)
    np.testing.assert_allclose(n(ya).reshape(M, N), yo2, rtol=e, atol=e * np.abs(yo2).max())
    assert xa.grad.dtype == torch.float32 and ma.lora_down.weight.grad is not None


def test_linear_module_dropout_train_eval_and_grad_consistency():
    M, K, N, r, p = 256, 320, 320, 4, 0.5
    torch.manual_seed(1)
    m = L.LoraInjectedLinear(K, N, False, r=r, dropout_p=p, scale=1.0).to(DEV)
    m.lora_up.weight.data.normal_(0, 0.1)
    x = torch.randn(M, K, device=DEV)
    W, A, B = n(m.linear.weight), n(m.lora_down.weight), n(m.lora_up.weight)
    m.eval()
    yo, _ = O.lora_linear_forward(n(x), W, None, A, B, 1.0)
    np.testing.assert_allclose(n(m(x)), yo, rtol=2e-4, atol=2e-4 * np.abs(yo).max())  # eval: dropout off
    m.train()
    xg = x.clone().requires_grad_(True)
    torch.manual_seed(77)
    y = m(xg)
    # regenerate the mask the kernels used from its (seed, offset): no mask tensor was ever stored.  The offset is a
    # device value drawn from torch's generator (ops.next_dropout_stream), so the same seed draws it again
    torch.manual_seed(77)
    seed, off = ops.next_dropout_stream(DEV)
    mk = torch.zeros(M, N, device=DEV)
    _C.rank_update_(mk, torch.ones(M, 1, device=DEV), torch.ones(1, N, device=DEV), _C.FACTOR_RK, 1.0, p, seed, off)
    mask = n(mk)
    assert 0.45 < (mask > 0).mean() < 0.55 and set(np.unique(mask)) == {0.0, 2.0}
    yo, _ = O.lora_linear_forward(n(x), W, None, A, B, 1.0, None, mask)
    np.testing.assert_allclose(n(y), yo, rtol=2e-4, atol=2e-4 * np.abs(yo).max())
    gy = torch.randn(M, N, device=DEV)
    (y * gy).sum().backward()
    dxo, ddo, duo, _, _ = O.lora_linear_backward(n(gy), n(x), W, A, B, 1.0, None, mask)
    np.testing.assert_allclose(n(xg.grad), dxo, rtol=2e-4, atol=2e-4 * np.abs(dxo).max())
    np.testing.assert_allclose(n(m.lora_up.weight.grad), duo, rtol=2e-4, atol=2e-4 * np.abs(duo).max())
    np.testing.assert_allclose(n(m.lora_down.weight.grad), ddo, rtol=2e-4, atol=2e-4 * np.abs(ddo).max())


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_conv_module_matches_reference_vectors(tag):
    d = _npz("conv_cases.npz")
    k, s, p, r = (int(v) for v in d[f"{tag}_geom"])
    Co, Ci = d[f"{tag}_W"].shape[:2]
    m = L.LoraInjectedConv2d(Ci, Co, k, s, p, r=r, dropout_p=0.0, scale=float(d[f"{tag}_scale"]))
    for mod, key in ((m.conv, "W"), (m.lora_down, "down"), (m.lora_up, "up")):
        mod.weight.data = torch.from_numpy(d[f"{tag}_{key}"])
    m.conv.bias.data = torch.from_numpy(d[f"{tag}_b"])
    m.to(DEV)
    x = torch.from_numpy(d[f"{tag}_x"]).to(DEV).requires_grad_(True)
    y = m(x)
    (y * torch.from_numpy(d[f"{tag}_gy"]).to(DEV)).sum().backward()
    for name, got, want in (("y", y, d[f"{tag}_y"]), ("dx", x.grad, d[f"{tag}_dx"]),
                            ("ddown", m.lora_down.weight.grad, d[f"{tag}_ddown"]),
                            ("dup", m.lora_up.weight.grad, d[f"{tag}_dup"])):
        np.testing.assert_allclose(n(got), want, rtol=5e-4, atol=5e-4 * np.abs(want).max(), err_msg=f"{tag} {name}")


def _ulp(x, dt):
    return np.abs(x) * {"bf16": 2.0 ** -7, "f16": 2.0 ** -10, "f32": 2.0 ** -22}[dt] + 1e-30


def test_collapse_matches_reference_vectors_on_device():
    d = _npz("collapse_cases.npz")
    for c in json.load(open(os.path.join(G, "collapse_cases.json"))):
        tag = c["tag"]
        wdt, abdt = DT[c["w_dtype"]], DT[c["ab_dtype"]]
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
        root.to(DEV)
        old = frozen.weight
        L.collapse_lora(root, c["alpha"])
        assert frozen.weight is not old and frozen.weight.dtype == wdt and frozen.weight.shape == old.shape
        got, ref = n(frozen.weight), d[f"{tag}_out"]
        if c["w_dtype"] == "f32":  # f32: only the r-term dot product's summation order can differ (|q| ~ 1)
            np.testing.assert_allclose(got, ref, rtol=0, atol=3e-7, err_msg=tag)
            continue
        # 1 ulp of the largest intermediate (W, alpha*up@down or their sum): a last-place tie in the r-term
        # f32 dot product may flip one rounding of the reference's sequence
        mag = np.maximum(np.abs(ref), np.maximum(np.abs(d[f"{tag}_W"]), np.abs(ref - d[f"{tag}_W"])))
        assert np.all(np.abs(got - ref) <= _ulp(mag, c["w_dtype"])), tag
        # 16-bit factors: the CPU reference's half matmul rounds its partial sums differently -> more last-place flips
        frac_ok = 0.01 if c["ab_dtype"] == "f32" else 0.05
        assert (got != ref).mean() < frac_ok, f"{tag}: {(got != ref).mean():.4f} differ"


def test_batched_merge_many_sites_one_launch():
    """Ragged mix of site shapes/ranks incl. column-tiled (K > LDS slab), odd K (scalar lanes), in-place."""
    shapes = [(320, 320, 4), (640, 768, 4), (2560, 320, 4), (1280, 2880, 4), (320, 1280, 16), (77, 33, 3), (8, 8, 1),
              (640, 5760, 8), (130, 648, 64)]
    for wdt, abdt in (("bf16", "f32"), ("f32", "f32"), ("f16", "f16"), ("bf16", "bf16")):
        sites, want, w0s = [], [], []
        for i, (N, K, r) in enumerate(shapes):
            w = rnd((N, K), wdt, 0.05, seed=20 + i)
            w0s.append(n(w))
            up = rnd((N, r), abdt, 0.1, seed=40 + i)
            down = rnd((r, K), abdt, 0.5, seed=60 + i)
            out = w if i % 2 else torch.empty_like(w)  # odd sites merge in place
            want.append(O.collapse(n(w), n(up), n(down), 0.9, wdt, abdt))
            sites.append((w, out, up, down))
        plan = _C.MergePlan(sites)
        plan.launch(0.9, _C.ROUND_REFERENCE)
        torch.cuda.synchronize()
        for i, (s, ref) in enumerate(zip(sites, want)):
            got = n(s[1])
            if wdt == "f32":
                np.testing.assert_allclose(got, ref, rtol=0, atol=3e-7)
                continue
            mag = np.maximum(np.abs(ref), np.maximum(np.abs(w0s[i]), np.abs(ref - w0s[i])))
            # 16-bit factors: the matmul result itself is rounded to 16 bits, so an f32 summation-order difference can
            # flip one place of p = up@down; that flip survives the alpha-multiply and may add to a last-place flip of
            # the final W + q rounding -> at most two places of the largest operand (never more; checked below)
            places = 1 if abdt == "f32" else 2
            assert np.all(np.abs(got - ref) <= places * _ulp(mag, wdt)), (wdt, abdt, shapes[i])
            assert (got != ref).mean() < (0.01 if abdt == "f32" else 0.05), (wdt, abdt, shapes[i], (got != ref).mean())
    # single-rounding mode is at least as close to the exact f64 result
    w, up, down = rnd((640, 320), "bf16", 0.05, seed=1), rnd((640, 4), "f32", 0.1, seed=2), rnd((4, 320), "f32", 0.5, seed=3)
    o_ref, o_once = torch.empty_like(w), torch.empty_like(w)
    _C.MergePlan([(w, o_ref, up, down)]).launch(0.9, _C.ROUND_REFERENCE)
    _C.MergePlan([(w, o_once, up, down)]).launch(0.9, _C.ROUND_ONCE)
    exact = n(w).astype(np.float64) + 0.9 * (n(up).astype(np.float64) @ n(down).astype(np.float64))
    assert np.abs(n(o_once) - exact).mean() <= np.abs(n(o_ref) - exact).mean() + 1e-12
    assert np.all(np.abs(n(o_once) - exact) <= _ulp(exact, "bf16") * 0.51 + 1e-9)


def test_merge_full_unet_size_properties():
    """All 144 SD1.5 sites (bf16, rank 4): alpha=0 is the identity; result equals torch's expression on device."""
    from lora_amd.standin import sd15_lora_site_shapes

    sites = []
    for i, (N, K) in enumerate(sd15_lora_site_shapes()):
        w = rnd((N, K), "bf16", 0.03, seed=i)
        sites.append((w, torch.empty_like(w), rnd((N, 4), "f32", 0.05, seed=1000 + i), rnd((4, K), "f32", 0.25, seed=2000 + i)))
    assert len(sites) == 144
    plan = _C.MergePlan(sites)
    assert plan.bytes_algorithmic == sum(2 * N * K * 2 + (N + K) * 4 * 4 for N, K in sd15_lora_site_shapes())
    plan.launch(0.0)
    assert all(torch.equal(s[0], s[1]) for s in sites)
    plan.launch(0.75)
    for w, out, up, down in sites[::7]:
        ref = w + 0.75 * (up @ down).type(w.dtype)
        diff = (out.float() - ref.float()).abs()
        assert (diff <= ref.float().abs() * 2.0 ** -7 + 1e-30).all() and (diff > 0).float().mean() < 0.01


def test_sumsq_and_clip_adamw_match_torch_vectors():
    d = _npz("optimizer_case.npz")
    p = torch.from_numpy(d["p0"]).to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    groups = _C.make_adamw_groups([(0, 700, 1e-2, 1e-2), (700, 1000, 5e-3, 1e-2)], DEV)
    ss = torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        g = torch.from_numpy(d[f"g{step}"]).to(DEV)
        _C.sumsq(g, ss)
        assert abs(ss.sqrt().item() - float(d[f"norm{step}"])) <= 1e-5 * max(1.0, float(d[f"norm{step}"]))
        _C.clip_adamw(p, g, m, v, groups, 2, ss, 1.0, 1.0, 0.9, 0.999, 1e-8, step, True)
        np.testing.assert_allclose(n(p), d[f"p{step}"], rtol=3e-6, atol=3e-7)
        assert g.abs().sum().item() == 0.0  # zero_grad fused
    # grad_scale = 1/world_size after a SUM all-reduce, no clipping
    p2 = torch.from_numpy(d["p0"]).to(DEV)
    m2, v2 = torch.zeros_like(p2), torch.zeros_like(p2)
    g = (torch.from_numpy(d["g1"]) * 4).to(DEV)
    _C.clip_adamw(p2, g, m2, v2, groups, 2, None, 0.25, 0.0, 0.9, 0.999, 1e-8, 1, False)
    pa, _, _ = O.adamw_step(d["p0"][:700], d["g1"][:700], np.zeros(700), np.zeros(700), 1, lr=1e-2)
    np.testing.assert_allclose(n(p2)[:700], pa, rtol=3e-6, atol=3e-7)
    assert g.abs().sum().item() > 0


def test_full_size_sd15_shapes_against_device_torch():
    """BASELINE config-1 sizes (B=4, 512^2): the GEGLU site M=16384,K=320,N=2560 and attn 1280."""
    for (M, K, N) in ((16384, 320, 2560), (1024, 1280, 1280), (308, 768, 320)):
        x, g = rnd((M, K), "bf16", seed=1), rnd((M, N), "bf16", seed=2)
        A, B = rnd((4, K), "f32", 0.25, seed=3), rnd((N, 4), "f32", 0.05, seed=4)
        t = _C.rowdot(x, A, _C.FACTOR_RK)
        t_ref = x.float() @ A.t()
        assert torch.allclose(t, t_ref, rtol=1e-4, atol=1e-4 * t_ref.abs().max().item())
        y = rnd((M, N), "bf16", seed=5)
        y_ref = (y.float() + 0.8 * (t_ref @ B.t())).to(torch.bfloat16)
        _C.rank_update_(y, t, B, _C.FACTOR_KR, 0.8)
        assert ((y.float() - y_ref.float()).abs() <= y_ref.float().abs() * 2.0 ** -7 + 1e-6).all()
        d_up = _C.colreduce(g, t, _C.FACTOR_KR, 0.8)
        d_ref = 0.8 * (g.float().t() @ t_ref)
        assert torch.allclose(d_up, d_ref, rtol=2e-4, atol=2e-4 * d_ref.abs().max().item())
        gt = _C.rowdot(g, B, _C.FACTOR_KR, 0.8)
        d_down = _C.colreduce(x, gt, _C.FACTOR_RK)
        dd_ref = (0.8 * (g.float() @ B)).t() @ x.float()
        assert torch.allclose(d_down, dd_ref, rtol=2e-4, atol=2e-4 * dd_ref.abs().max().item())


def test_bad_arguments_raise():
    x = rnd((8, 16), "bf16")
    with pytest.raises(ValueError):
        _C.rowdot(x.cpu(), rnd((4, 16), "f32"), _C.FACTOR_RK)
    with pytest.raises(TypeError):
        _C.rowdot(x.to(torch.float64), rnd((4, 16), "f32"), _C.FACTOR_RK)
    with pytest.raises(ValueError):
        _C.rowdot(rnd((8, 80), "bf16"), rnd((65, 80), "f32"), _C.FACTOR_RK)  # rank > 64
    with pytest.raises(ValueError):
        L.LoraInjectedLinear(4, 4, r=5)


# ------------------------------------------------------------------------------------------ fused K1/K2
FUSED_SHAPES = [(512, 320, 320, 4), (300, 768, 320, 4), (64, 1280, 1280, 8), (130, 320, 2560, 4), (100, 1280, 10240, 4),
                (33, 640, 640, 16), (17, 320, 320, 1), (64, 2560, 320, 4), (308, 768, 768, 4), (1, 32, 64, 2),
                (4096, 640, 640, 4)]


@pytest.mark.parametrize("M,K,N,r", FUSED_SHAPES)
@pytest.mark.parametrize("dt", ["bf16", "f32", "f16"])
def test_fused_linear_kernels_match_oracle(M, K, N, r, dt):
    assert _C.linear_plan(M, K, N, r).fused == 1
    x, W = rnd((M, K), dt, seed=1), rnd((N, K), dt, 0.05, seed=2)
    A, B = rnd((r, K), "f32", 0.3, seed=3), rnd((N, r), "f32", 0.2, seed=4)
    g = rnd((M, N), dt, seed=5)
    s = 0.7
    Xn, Wn, An, Bn, Gn = n(x), n(W), n(A), n(B), n(g)
    # forward (y0 stands for the frozen GEMM's output)
    y0 = rnd((M, N), dt, seed=6)
    y = y0.clone()
    t = _C.linear_fwd_(x, y, A, B, s, None, 0.0, 0, 0)
    T64 = Xn.astype(np.float64) @ An.T
    close(n(t), T64, np.abs(Xn) @ np.abs(An.T), msg="T")
    close(n(y), n(y0) + s * (T64 @ Bn.T), np.abs(n(y0)) + s * (np.abs(Xn) @ np.abs(An.T)) @ np.abs(Bn.T), dt, msg="y")
    # backward
    plan = _C.linear_plan(M, K, N, r)
    gt_part = torch.empty(plan.gt_part_floats, device=DEV)
    up_part = torch.empty(plan.up_part_floats, device=DEV)
    down_part = torch.empty(plan.down_part_floats, device=DEV)
    _C.linear_bwd_g(g, t, B, gt_part, up_part, s, 0.0, 0, 0)
    dx = (g.float() @ W.float()).to(DT[dt])  # stands for the frozen dX GEMM
    dx0 = n(dx)
    _C.linear_bwd_x(x, dx, gt_part, plan.nct_g, A, None, down_part)
    d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
    rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
            (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    _C.reduce_batched(*_C.make_reduce_table(rows, DEV))
    Tn = n(t).astype(np.float64)
    Gt = s * (Gn.astype(np.float64) @ Bn)
    close(n(gt_part).reshape(plan.nct_g, M, r).sum(0), Gt, s * (np.abs(Gn) @ np.abs(Bn)), msg="Gt")
    close(n(d_up), s * (Gn.T.astype(np.float64) @ Tn), s * (np.abs(Gn.T) @ np.abs(Tn)), msg="d_up")
    close(n(d_down), Gt.T @ Xn, np.abs(Gt.T) @ np.abs(Xn), msg="d_down")
    close(n(dx), dx0 + Gt @ An, np.abs(dx0) + np.abs(Gt) @ np.abs(An), dt, msg="dx")


def test_fused_linear_selector_dropout_and_bf16_factors():
    M, K, N, r, s, p = 192, 640, 320, 4, 1.3, 0.3
    x, y0, g = rnd((M, K), "bf16", seed=1), rnd((M, N), "bf16", seed=2), rnd((M, N), "bf16", seed=3)
    A, B = rnd((r, K), "bf16", 0.3, seed=4), rnd((N, r), "bf16", 0.2, seed=5)  # monkeypatched adapters: W's dtype
    sel = rnd((r, r), "f32", seed=6)
    mk = torch.zeros(M, N, device=DEV)
    _C.rank_update_(mk, torch.ones(M, 1, device=DEV), torch.ones(1, N, device=DEV), _C.FACTOR_RK, 1.0, p, 99, 5)
    mask = n(mk)
    y = y0.clone()
    t = _C.linear_fwd_(x, y, A, B, s, sel, p, 99, 5)
    yo, to = O.lora_linear_forward(n(x), np.zeros((N, K), np.float32), None, n(A), n(B), s, n(sel), mask)
    np.testing.assert_allclose(n(t), to, rtol=1e-4, atol=1e-4 * np.abs(to).max())
    np.testing.assert_allclose(n(y), n(y0) + yo, rtol=2 ** -7, atol=2 ** -7 * np.abs(yo).max())
    plan = _C.linear_plan(M, K, N, r)
    gt_part, up_part, down_part = (torch.empty(k, device=DEV) for k in
                                   (plan.gt_part_floats, plan.up_part_floats, plan.down_part_floats))
    _C.linear_bwd_g(g, t, B, gt_part, up_part, s, p, 99, 5)
    dx = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
    _C.linear_bwd_x(x, dx, gt_part, plan.nct_g, A, sel, down_part)
    d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
    rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
            (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    _C.reduce_batched(*_C.make_reduce_table(rows, DEV))
    dxo, ddo, duo, _, _ = O.lora_linear_backward(n(g), n(x), np.zeros((N, K), np.float32), n(A), n(B), s, n(sel), mask)
    np.testing.assert_allclose(n(d_up), duo, rtol=1e-3, atol=1e-3 * np.abs(duo).max())
    np.testing.assert_allclose(n(d_down), ddo, rtol=1e-3, atol=1e-3 * np.abs(ddo).max())
    np.testing.assert_allclose(n(dx), dxo, rtol=2 ** -7, atol=2 ** -7 * np.abs(dxo).max())
    # dX optional (first layer): same parameter gradients
    _C.linear_bwd_x(x, None, gt_part, plan.nct_g, A, sel, down_part)
    d_down2 = torch.empty(r, K, device=DEV)
    _C.reduce_batched(*_C.make_reduce_table([rows[1][:1] + (d_down2,) + rows[1][2:]], DEV))
    assert torch.equal(d_down, d_down2)


def test_plan_rejects_unfriendly_shapes():
    assert _C.linear_plan(16, 24, 40, 4).fused == 0    # too few 16-byte chunks per row
    assert _C.linear_plan(16, 320, 321, 4).fused == 0  # N % 8
    assert _C.linear_plan(16, 320, 320, 32).fused == 0  # rank > 16 -> primitives
    assert _C.linear_plan(16384, 320, 320, 4).fused == 1


def test_training_steps_on_device_match_cpu_path():
    """tiny UNet, 3 optimiser steps: device (fused kernels, deferred batched reduce, fused clip+AdamW) vs the CPU path."""
    import copy

    from lora_amd import trainer as T
    from lora_amd.standin import DDPMScheduler, tiny_unet

    torch.manual_seed(0)
    cpu = tiny_unet()
    cpu.requires_grad_(False)
    torch.manual_seed(5)
    L.inject_trainable_lora(cpu, r=4)
    for up, _ in L.extract_lora_ups_down(cpu):
        up.weight.data.normal_(0, 0.05)
    dev = copy.deepcopy(cpu).to(DEV)
    cpu.train(), dev.train()
    st_c = T.FlatLoraState([{"params": T.lora_params(cpu), "lr": 1e-3}], max_grad_norm=1.0)
    st_d = T.FlatLoraState([{"params": T.lora_params(dev), "lr": 1e-3}], max_grad_norm=1.0, device=torch.device(DEV))
    assert st_d.attach_direct_grads(dev) == len(L.extract_lora_ups_down(dev))
    sched = DDPMScheduler()
    for it in range(3):
        g = torch.Generator().manual_seed(it)
        lat, ehs = torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 7, 32, generator=g)
        noise, t = torch.randn(4, 4, 16, 16, generator=g), torch.randint(0, 1000, (4,), generator=g)
        lc = T.forward_backward(cpu, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)
        ld = T.forward_backward(dev, sched, lat.to(DEV), ehs.to(DEV), T.StepConfig(), noise=noise.to(DEV), timesteps=t.to(DEV))
        assert abs(lc.item() - ld.item()) < 1e-4 * max(1.0, abs(lc.item()))
        if it == 0:
            st_d.reduce_pending()
            np.testing.assert_allclose(n(st_d.flat_g), st_c.flat_g.numpy(), rtol=2e-3, atol=2e-5)
        st_c.step(st_c.all_reduce())
        st_d.step(st_d.all_reduce())
    # AdamW normalises the update (a near-zero gradient's sign decides +-lr): allow 5 % of the 3*lr bound
    np.testing.assert_allclose(n(st_d.flat_p), st_c.flat_p.numpy(), rtol=1e-3, atol=1.5e-4)
    # a second backward before the step flushes the first one's partials instead of overwriting them
    g = torch.Generator().manual_seed(9)
    lat, ehs = torch.randn(2, 4, 16, 16, generator=g).to(DEV), torch.randn(2, 7, 32, generator=g).to(DEV)
    noise, t = torch.randn(2, 4, 16, 16, generator=g).to(DEV), torch.randint(0, 1000, (2,), generator=g).to(DEV)
    T.forward_backward(dev, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)
    st_d.reduce_pending()
    once = st_d.flat_g.clone()
    st_d.zero_grad()
    T.forward_backward(dev, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)
    T.forward_backward(dev, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)
    st_d.reduce_pending()
    assert torch.allclose(st_d.flat_g, 2 * once, rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------- K4 native conv adapter (csrc/conv.hip)
@pytest.mark.parametrize("tag", ["n1", "n2", "n3", "n4", "n5"])
def test_conv_native_module_matches_reference_vectors(tag):
    """LoraInjectedConv2d on native geometries vs vectors produced by the reference itself (f32)."""
    d = _npz("conv_native_cases.npz")
    k, s, p, r = (int(v) for v in d[f"{tag}_geom"])
    Co, Ci = d[f"{tag}_W"].shape[:2]
    B, _, Hh, Ww = d[f"{tag}_x"].shape
    assert _C.conv_plan(B, Ci, Co, Hh, Ww, k, r).native == 1
    m = L.LoraInjectedConv2d(Ci, Co, k, s, p, r=r, dropout_p=0.0, scale=float(d[f"{tag}_scale"]))
    for mod, key in ((m.conv, "W"), (m.lora_down, "down"), (m.lora_up, "up")):
        mod.weight.data = torch.from_numpy(d[f"{tag}_{key}"])
    m.conv.bias.data = torch.from_numpy(d[f"{tag}_b"])
    m.to(DEV)
    x = torch.from_numpy(d[f"{tag}_x"]).to(DEV).requires_grad_(True)
    y = m(x)
    (y * torch.from_numpy(d[f"{tag}_gy"]).to(DEV)).sum().backward()
    for name, got, want in (("y", y, d[f"{tag}_y"]), ("dx", x.grad, d[f"{tag}_dx"]),
                            ("ddown", m.lora_down.weight.grad, d[f"{tag}_ddown"]),
                            ("dup", m.lora_up.weight.grad, d[f"{tag}_dup"])):
        assert got.shape == want.shape, (tag, name)
        # the frozen conv runs in MIOpen (its own summation order); the low-rank terms are f32 fma chains
        np.testing.assert_allclose(n(got), want, rtol=5e-4, atol=5e-4 * np.abs(want).max(), err_msg=f"{tag} {name}")


CONV_KERNEL_CASES = [(2, 32, 48, 16, 16, 3, 4, "f32", False, 0.0), (1, 40, 24, 8, 32, 3, 16, "bf16", False, 0.0),
                     (3, 16, 64, 8, 8, 1, 4, "bf16", True, 0.0), (2, 24, 24, 24, 24, 3, 8, "f16", True, 0.0),
                     (4, 64, 32, 8, 8, 3, 3, "f32", False, 0.25), (2, 320, 320, 32, 32, 3, 4, "bf16", False, 0.0),
                     (1, 640, 320, 16, 16, 1, 16, "bf16", False, 0.0), (1, 8, 8, 96, 96, 3, 5, "f32", False, 0.0)]


@pytest.mark.parametrize("B,Ci,Co,Hh,Ww,ks,r,dt,use_sel,p", CONV_KERNEL_CASES)
def test_conv_kernels_match_oracle(B, Ci, Co, Hh, Ww, ks, r, dt, use_sel, p):
    """The four conv entry points called through the C-ABI vs the numpy oracle (frozen conv = 0 so that
    only the low-rank terms are compared; the dropout mask is recovered from the kernel itself)."""
    plan = _C.conv_plan(B, Ci, Co, Hh, Ww, ks, r)
    assert plan.native == 1
    pad, HW, scale = (ks - 1) // 2, Hh * Ww, 0.7
    x, g = rnd((B, Ci, Hh, Ww), dt, seed=1), rnd((B, Co, Hh, Ww), dt, seed=2)
    down, up = rnd((r, Ci, ks, ks), "f32", 0.2, seed=3), rnd((Co, r, 1, 1), "f32", 0.3, seed=4)
    sel = rnd((r, r), "f32", 0.5, seed=5) if use_sel else None
    seed, off = 1234, 7
    bufs = ops.conv_buffers(plan, B, r, HW, DEV)
    t_part, gt_part, gt, up_part, down_part = bufs
    t = torch.empty((B, r, Hh, Ww), dtype=torch.float32, device=DEV)
    _C.conv_down_fwd(x, down, sel, t_part, t, ks)
    y0 = rnd((B, Co, Hh, Ww), dt, seed=6)
    mask = None
    if p > 0:  # recover the mask: update zeros with T = ones-projection
        probe = torch.zeros((B, Co, Hh, Ww), dtype=DT[dt], device=DEV)
        ones_t = torch.ones((B, 1, Hh, Ww), dtype=torch.float32, device=DEV)
        _C.conv_up_fwd_(probe, ones_t, torch.ones((Co, 1, 1, 1), device=DEV), 1.0, p, seed, off)
        mask = n(probe)
        keep = (mask != 0).mean()
        assert abs(keep - (1 - p)) < 0.02, keep
        np.testing.assert_allclose(mask[mask != 0], 1.0 / (1 - p), rtol=1e-2)
        mask = (mask != 0).astype(np.float32) / np.float32(1 - p)
    y = y0.clone()
    _C.conv_up_fwd_(y, t, up, scale, p, seed, off)
    seln = n(sel) if use_sel else None
    yo, t_o = O.lora_conv2d_forward(n(x), np.zeros((Co, Ci, ks, ks), np.float32), None, n(down), n(up), scale, (1, 1),
                                    (pad, pad), (1, 1), mask=mask, selector=seln)
    absx = np.abs(n(x)).max() * np.abs(n(down)).sum(axis=(1, 2, 3)).max()
    close(n(t), t_o, absx, "f32", msg="T")
    absy = np.abs(t_o).max() * np.abs(n(up)).sum(axis=1).max() * scale * (1.0 / (1 - p))
    close(n(y), yo + n(y0), absy + np.abs(n(y0)), dt, msg="Y")
    # backward
    dx0 = rnd((B, Ci, Hh, Ww), dt, seed=8)
    dx = dx0.clone()
    _C.conv_bwd_g(g, t, up, sel, gt_part, gt, up_part, scale, p, seed, off)
    _C.conv_bwd_x(x, dx, gt, down, down_part, ks)
    d_up = torch.empty((Co, r, 1, 1), device=DEV)
    d_down = torch.empty((r, Ci, ks, ks), device=DEV)
    table, nn_, total = _C.make_reduce_table(ops.conv_reduce_rows(bufs, plan, Ci, Co, ks, r, d_up, d_down, 0.0), DEV)
    _C.reduce_batched(table, nn_, total)
    dxo, ddo, duo = O.lora_conv2d_backward(n(g), n(x), np.zeros((Co, Ci, ks, ks), np.float32), n(down), n(up), scale,
                                           (1, 1), (pad, pad), (1, 1), selector=seln, mask=mask)
    kk = 1e-4  # f32 sums over up to B*H*W (dUp/dDown) or C*9 (T, Gt) terms in a different order than numpy
    np.testing.assert_allclose(n(d_up), duo, rtol=kk * 10, atol=kk * np.abs(duo).max() + 1e-6, err_msg="dUp")
    np.testing.assert_allclose(n(d_down), ddo, rtol=kk * 10, atol=kk * np.abs(ddo).max() + 1e-6, err_msg="dDown")
    close(n(dx), dxo + n(dx0), np.abs(dxo).max() + np.abs(n(dx0)), dt, k=1e-4, msg="dX")  # ONE rounding, any rank
    # dDown-only variant (input does not need a gradient) writes the same partials
    down_part.zero_()
    _C.conv_bwd_x(x, None, gt, down, down_part, ks)
    d_down2 = torch.empty_like(d_down)
    rows = ops.conv_reduce_rows(bufs, plan, Ci, Co, ks, r, d_up, d_down2, 0.0)[1:]
    table, nn_, total = _C.make_reduce_table(rows, DEV)
    _C.reduce_batched(table, nn_, total)
    np.testing.assert_allclose(n(d_down2), n(d_down), rtol=1e-6, atol=1e-7)


def test_conv_native_bf16_module_against_device_torch():
    """SD1.5-sized ResNet conv site (bf16, rank 16, dropout off): module vs the same op sequence in torch on device."""
    torch.manual_seed(0)
    B, Ci, Co, Hh = 2, 640, 320, 32
    m = L.LoraInjectedConv2d(Ci, Co, 3, 1, 1, r=16, dropout_p=0.0, scale=1.0)
    m.lora_up.weight.data.normal_(0, 0.05)
    m.to(DEV).to(torch.bfloat16)
    m.lora_up.weight.data = m.lora_up.weight.data.float()
    m.lora_down.weight.data = m.lora_down.weight.data.float()
    x = torch.randn(B, Ci, Hh, Hh, device=DEV).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(B, Co, Hh, Hh, device=DEV).to(torch.bfloat16)
    y = m(x)
    (y.float() * gy.float()).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    dn, upw = m.lora_down.weight.detach().clone().requires_grad_(True), m.lora_up.weight.detach().clone().requires_grad_(True)
    F = torch.nn.functional
    yr = F.conv2d(xr, m.conv.weight.float(), m.conv.bias.float(), 1, 1) + F.conv2d(F.conv2d(xr, dn, None, 1, 1), upw)
    (yr * gy.float()).sum().backward()
    assert (y.float() - yr).abs().max() <= 2.0 ** -7 * yr.abs().max() + 1e-3
    assert (x.grad.float() - xr.grad).abs().max() <= 2.0 ** -6 * xr.grad.abs().max()
    for got, want in ((m.lora_down.weight.grad, dn.grad), (m.lora_up.weight.grad, upw.grad)):
        assert (got - want).abs().max() <= 2e-3 * want.abs().max(), (got - want).abs().max() / want.abs().max()


def test_conv_plan_geometry():
    assert _C.conv_plan(4, 320, 320, 64, 64, 3, 4).native == 1
    assert _C.conv_plan(1, 1280, 1280, 12, 12, 3, 16).native == 0   # 12-pixel rows are not 16-byte chunks
    assert _C.conv_plan(1, 1280, 1280, 12, 12, 1, 16).native == 1   # 1x1 only needs H*W % 8 == 0
    assert _C.conv_plan(1, 64, 64, 8, 8, 5, 4).native == 0
    assert _C.conv_plan(1, 64, 64, 8, 8, 3, 17).native == 0
    pl = _C.conv_plan(1, 320, 320, 96, 96, 3, 16)
    assert pl.cpw_in == 60 and pl.ngroups_in == (96 * 96 // 8 + 59) // 60 and pl.rank_pad == 16


@pytest.mark.parametrize("extended", [False, True])
def test_training_steps_extended_injection_device_vs_cpu(extended):
    """tiny UNet with Linear (+Conv2d) adapters, 2 optimiser steps, 32x32 latents: native conv kernels where the maps
    are 16-byte friendly, the library-conv branch elsewhere; gradients land in the flat buffer either way."""
    import copy

    from lora_amd import trainer as T
    from lora_amd.standin import DDPMScheduler, tiny_unet

    torch.manual_seed(0)
    cpu = tiny_unet()
    cpu.requires_grad_(False)
    torch.manual_seed(5)
    if extended:
        L.inject_trainable_lora_extended(cpu, r=4)
    else:
        L.inject_trainable_lora(cpu, r=4)
    for mod in cpu.modules():
        if isinstance(mod, (L.LoraInjectedLinear, L.LoraInjectedConv2d)):
            mod.dropout.p = 0.0
            mod.lora_up.weight.data.normal_(0, 0.05)
    dev = copy.deepcopy(cpu).to(DEV)
    cpu.train(), dev.train()
    st_c = T.FlatLoraState([{"params": T.lora_params(cpu), "lr": 1e-3}], max_grad_norm=1.0)
    st_d = T.FlatLoraState([{"params": T.lora_params(dev), "lr": 1e-3}], max_grad_norm=1.0, device=torch.device(DEV))
    n_sites = st_d.attach_direct_grads(dev)
    assert n_sites == len(L.extract_lora_ups_down(dev, L.UNET_EXTENDED_TARGET_REPLACE))
    sched = DDPMScheduler()
    for it in range(2):
        g = torch.Generator().manual_seed(it)
        lat, ehs = torch.randn(2, 4, 32, 32, generator=g), torch.randn(2, 7, 32, generator=g)
        noise, t = torch.randn(2, 4, 32, 32, generator=g), torch.randint(0, 1000, (2,), generator=g)
        lc = T.forward_backward(cpu, sched, lat, ehs, T.StepConfig(), noise=noise, timesteps=t)
        ld = T.forward_backward(dev, sched, lat.to(DEV), ehs.to(DEV), T.StepConfig(), noise=noise.to(DEV), timesteps=t.to(DEV))
        assert abs(lc.item() - ld.item()) < 2e-4 * max(1.0, abs(lc.item()))
        if it == 0:
            st_d.reduce_pending()
            np.testing.assert_allclose(n(st_d.flat_g), st_c.flat_g.numpy(), rtol=5e-3, atol=5e-5)
        st_c.step(st_c.all_reduce())
        st_d.step(st_d.all_reduce())
    # AdamW normalises the update: where a gradient is ~0 its sign decides +-lr, so allow 15 % of the 2*lr bound
    np.testing.assert_allclose(n(st_d.flat_p), st_c.flat_p.numpy(), rtol=2e-3, atol=3e-4)


# ----------------------------------------------------------------------------- K1 fully fused MFMA GEMM + LoRA (gemm_fused.hip)
GEMM_SHAPES = [(16384, 320, 320, 4), (4096, 640, 640, 4), (1024, 1280, 1280, 8), (308, 768, 320, 4), (256, 1280, 10240, 4),
               (1000, 320, 2560, 16), (77, 64, 24, 3), (130, 768, 768, 1)]


@pytest.mark.parametrize("M,K,N,r", GEMM_SHAPES)
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("tile", [0, 21, 22, 23, 24, 31, 32, 33, 34])
def test_fused_gemm_matches_oracle(M, K, N, r, dt, tile):
    """Y = X W^T + b + s (X down^T) up^T and T through lora_amd_linear_gemm_fwd vs the numpy oracle (asymmetric
    operands: a transposed tile or fragment would not pass).  Tolerance: f32-accumulated 16-bit products, one output
    rounding, plus the reference-faithful rounding of T to the activation dtype before the up-projection."""
    x, w = rnd((M, K), dt, 1.0, seed=1), rnd((N, K), dt, 0.05, seed=2)
    b = rnd((N,), dt, 0.5, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    y, t = _C.linear_gemm_fwd(x, w, b, down, up, 0.7, tile)
    X, W, Bv, A, U = n(x), n(w), n(b), n(down), n(up)  # `down` enters at f32 precision (hi + lo split)
    t_ref = X @ A.T
    close(n(t), t_ref, np.abs(X) @ np.abs(A).T, "f32", k=3e-5, msg="T")
    U16 = O.round_to(0.7 * U, dt)
    y_ref = X @ W.T + Bv + O.round_to(t_ref, dt) @ U16.T
    absref = np.abs(X) @ np.abs(W).T + np.abs(Bv) + np.abs(t_ref) @ np.abs(U16).T
    close(n(y), y_ref, absref, dt, k=2e-3 if dt == "bf16" else 3e-4, msg="Y")


@pytest.mark.parametrize("M,K,N,r", [(4096, 640, 640, 4), (1000, 320, 2560, 16), (308, 768, 320, 4), (130, 768, 768, 3)])
@pytest.mark.parametrize("tile", [0, 22, 24, 33])
def test_fused_gemm_input_gradient(M, K, N, r, tile):
    """dX = G W + s (G up) down and Gt = s G up through the same MFMA kernel on W^T with the factors read in place
    (factor_layout 3), then the dUp-only / dDown-only passes: against the oracle's backward."""
    dt, s = "bf16", 0.6
    x, g, w = rnd((M, K), dt, 1.0, seed=1), rnd((M, N), dt, 1.0, seed=2), rnd((N, K), dt, 0.05, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    dx, gt = _C.linear_gemm_dx(g, _C.weight_t(w), down, up, s, tile)
    G, X, W, A, U = n(g), n(x), n(w), n(down), n(up)
    gt_ref = s * (G @ U)
    close(n(gt), gt_ref, s * (np.abs(G) @ np.abs(U)), "f32", k=3e-5, msg="Gt")
    dx_ref = G @ W + O.round_to(G @ U, dt) @ O.round_to(s * A, dt)
    close(n(dx), dx_ref, np.abs(G) @ np.abs(W) + np.abs(gt_ref) @ np.abs(A), dt, k=2e-3, msg="dX")
    # parameter gradients from the two light passes
    plan = _C.linear_plan(M, K, N, r)
    if plan.fused:
        t = torch.from_numpy(X @ A.T).to(DEV)
        up_part = torch.empty(plan.up_part_floats, device=DEV)
        down_part = torch.empty(plan.down_part_floats, device=DEV)
        _C.linear_bwd_g(g, t, up, None, up_part, s, 0.0, 0, 0)
        _C.linear_bwd_x(x, None, gt, 1, down, None, down_part)
        d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
        rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
                (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
        table, nn_, total = _C.make_reduce_table(rows, DEV)
        _C.reduce_batched(table, nn_, total)
        _, ddo, duo, _, _ = O.lora_linear_backward(G, X, W, A, U, s)
        np.testing.assert_allclose(n(d_up), duo, rtol=1e-3, atol=1e-3 * np.abs(duo).max())
        np.testing.assert_allclose(n(d_down), ddo, rtol=1e-3, atol=1e-3 * np.abs(ddo).max())
        # the ONE-launch form of the same two passes (what the autograd function uses) writes identical partials
        up_part2, down_part2 = torch.zeros_like(up_part), torch.zeros_like(down_part)
        _C.linear_bwd_factors(g, t, up_part2, x, gt, down_part2, r, s)
        prev = _C.rank16_mfma(0)  # bit identity is between the two VALU forms (rank tile 16 otherwise runs csrc/rank16_mfma.hip)
        try:
            up_part_v = torch.empty_like(up_part)
            _C.linear_bwd_g(g, t, up, None, up_part_v, s, 0.0, 0, 0)
        finally:
            _C.rank16_mfma(prev)
        assert torch.equal(up_part2, up_part_v) and torch.equal(down_part2, down_part)


@pytest.mark.parametrize("M,K,N,r,d,D", [(4096, 320, 320, 4, 40, 64), (1000, 640, 640, 8, 80, 128), (300, 320, 320, 4, 40, 48),
                                         (260, 1280, 1280, 16, 160, 192), (77, 320, 640, 3, 40, 64)])
@pytest.mark.parametrize("tile", [22, 24, 21, 33])
def test_fused_gemm_head_padded_layouts(M, K, N, r, d, D, tile):
    """Head-padded operands of the fused kernel: the padded output is exactly pack(dense output) with a zero pad, a
    padded input gives exactly the dense result whatever sits in its pad columns (NaN here), in both directions."""
    dt, s = "bf16", 0.7
    x, w, b = rnd((M, K), dt, 1.0, seed=1), rnd((N, K), dt, 0.05, seed=2), rnd((N,), dt, 0.5, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    lay_n, lay_k = (N // d, d, D), (K // d, d, D)

    def nan_pad(t, lay):
        p = ops.pack_heads(t, lay)
        p.view(t.shape[0], lay[0], lay[2])[..., lay[1]:] = float("nan")
        return p

    y, t = _C.linear_gemm_fwd(x, w, b, down, up, s, tile)
    yp, tp = _C.linear_gemm_fwd(x, w, b, down, up, s, tile, y_heads=lay_n)
    assert torch.equal(yp, ops.pack_heads(y, lay_n)) and torch.equal(tp, t)
    y2, t2 = _C.linear_gemm_fwd(nan_pad(x, lay_k), w, b, down, up, s, tile, x_heads=lay_k)
    assert torch.equal(y2, y) and torch.equal(t2, t)
    y3, _ = _C.linear_gemm_fwd(nan_pad(x, lay_k), w, b, down, up, s, tile, x_heads=lay_k, y_heads=lay_n)
    assert torch.equal(y3, ops.pack_heads(y, lay_n))
    # input-gradient direction: G padded in, dX padded out
    g = rnd((M, N), dt, 1.0, seed=6)
    wt = _C.weight_t(w)
    dx, gt = _C.linear_gemm_dx(g, wt, down, up, s, tile)
    dxp, gtp = _C.linear_gemm_dx(nan_pad(g, lay_n), wt, down, up, s, tile, g_heads=lay_n, dx_heads=lay_k)
    assert torch.equal(dxp, ops.pack_heads(dx, lay_k)) and torch.equal(gtp, gt)
    # both factor-gradient partials from padded G and X
    plan = _C.linear_plan(M, K, N, r)
    if plan.fused:
        up_part, down_part = torch.zeros(plan.up_part_floats, device=DEV), torch.zeros(plan.down_part_floats, device=DEV)
        up_part2, down_part2 = torch.zeros_like(up_part), torch.zeros_like(down_part)
        _C.linear_bwd_factors(g, t, up_part, x, gt, down_part, r, s)
        _C.linear_bwd_factors(nan_pad(g, lay_n), t, up_part2, nan_pad(x, lay_k), gt, down_part2, r, s, None,
                              g_heads=lay_n, x_heads=lay_k)
        assert torch.equal(up_part2, up_part) and torch.equal(down_part2, down_part)


def test_attention_block_in_padded_head_layout_equals_regular_path(monkeypatch):
    """CrossAttention with adapters: q/k/v written and the attention output read in the padded head layout by the fused
    kernels (forward and backward) vs the regular pad-and-slice path."""
    import torch.nn as nn

    from lora_amd.standin import attention
    from lora_amd.standin.unet import CrossAttention

    torch.manual_seed(0)
    att = CrossAttention(320, None, heads=8, dim_head=40).to(DEV).to(torch.bfloat16).requires_grad_(False)
    L.inject_trainable_lora(att, target_replace_module={"CrossAttention"}, r=4)
    for m in att.modules():
        if type(m).__name__ == "LoraInjectedLinear":
            nn.init.normal_(m.lora_up.weight, std=0.05)
            m.lora_up.weight.data = m.lora_up.weight.data.float()
            m.lora_down.weight.data = m.lora_down.weight.data.float()
    x = (torch.randn(2, 1024, 320, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    params = [p for p in att.parameters() if p.requires_grad]

    def run():
        y = att(x)
        return y, torch.autograd.grad(y.float().square().mean(), [x] + params)

    monkeypatch.setenv("LORA_AMD_GEMM", "22")      # fused tiles both ways (what the tuner picks for these sites)
    monkeypatch.setenv("LORA_AMD_GEMM_BWD", "22")
    y0, g0 = run()
    monkeypatch.setattr(attention, "FORCE_PAD", 64)
    y1, g1 = run()
    torch.testing.assert_close(y1.float(), y0.float(), rtol=2e-2, atol=2e-2 * float(y0.float().abs().max()))
    for a, b in zip(g1, g0):
        torch.testing.assert_close(a.float(), b.float(), rtol=5e-2, atol=3e-2 * float(b.float().abs().max()) + 1e-8)


def test_fused_gemm_random_shapes_against_device_torch():
    """40 seeded random (M, K, N, r, dtype, tile) draws incl. ragged M / N tails: fused kernel vs torch on the device."""
    rng = np.random.default_rng(0)
    for it in range(40):
        M = int(rng.integers(1, 700))
        K = 64 * int(rng.integers(1, 9))
        N = 8 * int(rng.integers(2, 90))
        r = int(rng.integers(1, 17))
        dt = ["bf16", "f16"][it % 2]
        tile = [0, 21, 22, 23, 24, 31, 32, 33, 34][it % 9]
        x, w, b = rnd((M, K), dt, 1.0, seed=it), rnd((N, K), dt, 0.1, seed=100 + it), rnd((N,), dt, 0.3, seed=200 + it)
        down, up = rnd((r, K), "f32", 0.2, seed=300 + it), rnd((N, r), "f32", 0.3, seed=400 + it)
        y, t = _C.linear_gemm_fwd(x, w, b if it % 3 else None, down, up, 0.5, tile)
        xf, wf = x.float(), w.float()
        t_ref = xf @ down.t()
        y_ref = xf @ wf.t() + (b.float() if it % 3 else 0.0) + 0.5 * (t_ref @ up.t())
        tag = (M, K, N, r, dt, tile)
        assert (t - t_ref).abs().max() <= 1e-4 * (xf.abs() @ down.abs().t()).max() + 1e-6, tag
        tol = (2.0 ** -7 if dt == "bf16" else 2.0 ** -10) * (y_ref.abs().max() + (t_ref.abs() @ up.abs().t()).max())
        assert (y.float() - y_ref).abs().max() <= tol, (tag, float((y.float() - y_ref).abs().max()), float(tol))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_ti_rows_step_matches_full_table_adamw(dt):
    """lora_amd_ti_rows_step (placeholder rows only, one launch) == the reference's AdamW over the whole embedding
    table + norm decay toward 0.4 + restoring every other row (cli_lora_pti.py:433-479), 4 steps."""
    torch.manual_seed(0)
    V, Hd, ids, lr, wd = 300, 768, [297, 299], 5e-3, 0.01
    table0 = torch.randn(V, Hd) * 0.02
    table = table0.clone().to(DT[dt]).to(DEV)
    ref = torch.nn.Parameter(table0.clone().to(DT[dt]).float())
    opt = torch.optim.AdamW([ref], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    orig = ref.data.clone()
    keep = torch.ones(V, dtype=torch.bool)
    keep[ids] = False
    idt = torch.tensor(ids, device=DEV)
    rows = table[idt].float().clone()
    m, v = torch.zeros_like(rows), torch.zeros_like(rows)
    for step in range(1, 5):
        g = torch.randn(V, Hd, generator=torch.Generator().manual_seed(step)) * 0.1
        g = g.to(DT[dt])
        ref.grad = g.float().clone()
        opt.step()
        with torch.no_grad():
            pre = ref[~keep].norm(dim=-1, keepdim=True)
            lam = min(1.0, 100 * lr)
            ref[~keep] = torch.nn.functional.normalize(ref[~keep], dim=-1) * (pre + lam * (0.4 - pre))
            ref[keep] = orig[keep]
        _C.ti_rows_step(table, g.to(DEV), idt, rows, m, v, lr, step, weight_decay=wd, decay_lambda=lam)
    np.testing.assert_allclose(n(rows), ref.data[~keep].numpy(), rtol=2e-5, atol=2e-6)
    tol = 0 if dt == "f32" else 2.0 ** -8
    np.testing.assert_allclose(n(table)[ids], ref.data[~keep].numpy(), rtol=tol + 2e-5, atol=2e-6)
    assert torch.equal(table[keep.to(DEV)].cpu(), table0.to(DT[dt])[keep])  # every other row untouched
