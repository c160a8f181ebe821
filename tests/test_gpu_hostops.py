"""Frozen host-model fusions (csrc/hostops.hip) against the ATen sequences they replace, evaluated in f32.

Tolerances: f32 tensors rel 1e-4 / abs 2e-5 (fast exp / rcp, A&S erf: <= 1.5e-7 absolute); bf16 / f16 outputs one
rounding of the f32 result (rel 2^-7 / 2^-10) plus the same absolute floor scaled to the data.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from lora_amd import _C
from lora_amd.standin import fused

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: (1e-4, 2e-5), torch.bfloat16: (2.0 ** -7, 2e-2), torch.float16: (2.0 ** -10, 2e-3)}


def _close(got, want, dt, scale=1.0, msg=""):
    rtol, atol = TOL[dt]
    torch.testing.assert_close(got.float().cpu(), want.float().cpu(), rtol=rtol, atol=atol * scale, msg=lambda m: f"{msg}: {m}")


GN_SHAPES = [(4, 320, 64, 64, 32), (2, 640, 32, 32, 32), (2, 1280, 8, 8, 32), (1, 960, 16, 16, 32), (3, 64, 4, 6, 8),
             (2, 32, 8, 8, 32), (1, 1920, 48, 48, 32), (2, 96, 2, 4, 3)]


@pytest.mark.parametrize("B,C,H,W,G", GN_SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("act", [True, False])
def test_groupnorm_act_forward_backward(B, C, H, W, G, dt, act):
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    # per-channel offsets several sigma wide: the statistics must not lose them to cancellation
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + torch.randn(1, C, 1, 1, generator=g) * 3.0).to(dt).to(DEV)
    norm = nn.GroupNorm(G, C, eps=1e-5).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.3)
    norm = norm.to(dt).requires_grad_(False)
    gout = torch.randn(B, C, H, W, generator=g).to(dt).to(DEV)
    assert fused._gn_native(x, norm), "geometry expected on the HIP path"

    xg = x.clone().requires_grad_(True)
    y = fused.group_norm_act(xg, norm, act)
    y.backward(gout)

    xr = x.float().requires_grad_(True)
    yr = F.group_norm(xr, G, norm.weight.float(), norm.bias.float(), norm.eps)
    if act:
        yr = F.silu(yr)
    yr.backward(gout.float())
    _close(y, yr, dt, msg="forward")
    _close(xg.grad, xr.grad, dt, scale=float(xr.grad.abs().max()) + 1e-6, msg="input gradient")


NHWC_SHAPES = [(4, 320, 64, 64, 32), (2, 640, 32, 32, 32), (2, 1280, 8, 8, 32), (1, 960, 16, 16, 32), (3, 64, 4, 6, 8),
               (2, 32, 8, 8, 32), (1, 1920, 24, 24, 32), (2, 96, 3, 5, 3), (2, 2560, 8, 8, 32), (1, 88, 5, 7, 11)]


@pytest.mark.parametrize("B,C,H,W,G", NHWC_SHAPES)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("act", [True, False])
def test_groupnorm_act_channels_last_forward_backward(B, C, H, W, G, dt, act):
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + torch.randn(1, C, 1, 1, generator=g) * 3.0).to(dt).to(DEV)
    x = x.contiguous(memory_format=torch.channels_last)
    norm = nn.GroupNorm(G, C, eps=1e-5).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.3)
    norm = norm.to(dt).requires_grad_(False)
    gout = torch.randn(B, C, H, W, generator=g).to(dt).to(DEV)  # NCHW-contiguous on purpose: converted inside
    assert fused._gn_native_nhwc(x, norm) and not fused._gn_native(x, norm)

    xg = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    y = fused.group_norm_act(xg, norm, act)
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gout)
    assert xg.grad.shape == x.shape

    xr = x.float().contiguous().requires_grad_(True)
    yr = F.group_norm(xr, G, norm.weight.float(), norm.bias.float(), norm.eps)
    if act:
        yr = F.silu(yr)
    yr.backward(gout.float())
    _close(y, yr, dt, msg="forward")
    _close(xg.grad, xr.grad, dt, scale=float(xr.grad.abs().max()) + 1e-6, msg="input gradient")


@pytest.mark.parametrize("B,C,H,W,G", [(4, 320, 32, 32, 32), (2, 1280, 8, 8, 32), (3, 64, 4, 6, 8)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_groupnorm_addend_is_x_plus_broadcast(B, C, H, W, G, dt, layout):
    """group_norm_act(x, addend=t) == group_norm_act(x + t[:, :, None, None]) incl. gradients w.r.t. x and t (the
    channels_last kernels fold t into the statistics / affine; NCHW adds it first)."""
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5).to(dt).to(DEV)
    t = (torch.randn(B, C, generator=g) * 2.0).to(dt).to(DEV)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    norm = nn.GroupNorm(G, C).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.3)
    norm = norm.to(dt).requires_grad_(False)
    gout = torch.randn(B, C, H, W, generator=g).to(dt).to(DEV)
    xg, tg = x.clone(memory_format=torch.preserve_format).requires_grad_(True), t.clone().requires_grad_(True)
    y = fused.group_norm_act(xg, norm, True, addend=tg)
    y.backward(gout)
    xr, tr = x.float().contiguous().requires_grad_(True), t.float().requires_grad_(True)
    yr = F.silu(F.group_norm(xr + tr[:, :, None, None], G, norm.weight.float(), norm.bias.float(), norm.eps))
    yr.backward(gout.float())
    _close(y, yr, dt, msg="forward")
    _close(xg.grad, xr.grad, dt, scale=float(xr.grad.abs().max()) + 1e-6, msg="dx")
    _close(tg.grad, tr.grad, dt, scale=float(tr.grad.abs().max()) + 1e-6, msg="d addend")


def test_groupnorm_statistics_and_fallbacks():
    x = torch.randn(2, 64, 8, 8, device=DEV) * 2 + 5
    norm = nn.GroupNorm(8, 64).to(DEV).requires_grad_(False)
    y, stats = _C.groupnorm_fwd(x, norm.weight, norm.bias, 8, norm.eps, False)
    xg = x.view(2, 8, -1)
    torch.testing.assert_close(stats[:, 0], xg.mean(-1).reshape(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[:, 1], (xg.var(-1, unbiased=False) + norm.eps).rsqrt().reshape(-1), rtol=1e-5, atol=1e-6)
    # no gradient requested for x -> backward skipped, still usable under no_grad
    with torch.no_grad():
        _close(fused.group_norm_act(x, norm, True), F.silu(norm(x)), torch.float32)
    # HW % 8 != 0, trainable affine, channels_last and CPU tensors take the ATen sequence
    assert not fused._gn_native(torch.randn(1, 32, 6, 6, device=DEV), nn.GroupNorm(8, 32).to(DEV).requires_grad_(False))
    assert not fused._gn_native(x, nn.GroupNorm(8, 64).to(DEV))
    assert not fused._gn_native(x.contiguous(memory_format=torch.channels_last), norm)
    assert not fused._gn_native(x.cpu(), norm)
    t = nn.GroupNorm(8, 64).to(DEV)
    fused.group_norm_act(x.clone().requires_grad_(True), t, True).sum().backward()
    assert t.weight.grad is not None and t.bias.grad is not None
    with pytest.raises(ValueError):
        _C.groupnorm_fwd(torch.randn(1, 32, 6, 6, device=DEV), norm.weight[:32], norm.bias[:32], 8, 1e-5, True)


@pytest.mark.parametrize("shape", [(4, 4096, 2560), (2, 1024, 5120), (3, 77, 64), (1, 1, 16), (5, 333, 48)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float16])
def test_geglu_forward_backward(shape, dt):
    g = torch.Generator().manual_seed(sum(shape))
    y = (torch.randn(*shape, generator=g) * 2.0).to(dt).to(DEV)
    gout = torch.randn(*shape[:-1], shape[-1] // 2, generator=g).to(dt).to(DEV)
    yg = y.clone().requires_grad_(True)
    out = fused.geglu(yg)
    assert out.shape == gout.shape
    out.backward(gout)

    yr = y.float().requires_grad_(True)
    h, gate = yr.chunk(2, dim=-1)
    outr = h * F.gelu(gate)
    outr.backward(gout.float())
    _close(out, outr, dt, msg="forward")
    _close(yg.grad, yr.grad, dt, scale=float(yr.grad.abs().max()) + 1e-6, msg="gradient")


@pytest.mark.parametrize("shape", [(4, 4096, 320), (4, 1024, 640), (2, 256, 1280), (3, 77, 768), (1, 5, 8), (7, 33, 2560),
                                   (2, 9, 1024)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32, torch.float16])
def test_layernorm_forward_backward(shape, dt):
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 1.5 + torch.randn(shape[-1], generator=g) * 2.0).to(dt).to(DEV)
    norm = nn.LayerNorm(shape[-1]).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(shape[-1], generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(shape[-1], generator=g) * 0.3)
    norm = norm.to(dt).requires_grad_(False)
    gout = torch.randn(*shape, generator=g).to(dt).to(DEV)
    xg = x.clone().requires_grad_(True)
    y = fused.layer_norm(xg, norm)
    assert type(y.grad_fn).__name__ == "_LayerNormBackward", "expected the HIP path"
    y.backward(gout)
    xr = x.float().requires_grad_(True)
    yr = F.layer_norm(xr, (shape[-1],), norm.weight.float(), norm.bias.float(), norm.eps)
    yr.backward(gout.float())
    _close(y, yr, dt, msg="forward")
    _close(xg.grad, xr.grad, dt, scale=float(xr.grad.abs().max()) + 1e-6, msg="input gradient")


@pytest.mark.parametrize("shape", [(4, 4096, 320), (2, 256, 1280), (3, 77, 768), (1, 5, 8)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_add_layernorm_equals_add_then_layernorm(shape, dt):
    g = torch.Generator().manual_seed(sum(shape) + 1)
    a = (torch.randn(*shape, generator=g) * 1.5).to(dt).to(DEV)
    b = (torch.randn(*shape, generator=g) * 1.5 + torch.randn(shape[-1], generator=g)).to(dt).to(DEV)
    norm = nn.LayerNorm(shape[-1]).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(shape[-1], generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(shape[-1], generator=g) * 0.3)
    norm = norm.to(dt).requires_grad_(False)
    gs, gy = torch.randn(*shape, generator=g).to(dt).to(DEV), torch.randn(*shape, generator=g).to(dt).to(DEV)
    ag, bg = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    s, y = fused.add_layer_norm(ag, bg, norm)
    assert type(y.grad_fn).__name__ == "_AddLayerNormBackward"
    torch.autograd.backward([s, y], [gs, gy])
    assert torch.equal(s, a + b), "the sum must be exactly what the separate add produces"
    ar, br = a.float().requires_grad_(True), b.float().requires_grad_(True)
    sr = (ar + br).to(dt).float() if dt != torch.float32 else ar + br  # the norm sees the rounded sum
    sr_ = (ar + br)
    yr = F.layer_norm(sr_ + (sr - sr_).detach(), (shape[-1],), norm.weight.float(), norm.bias.float(), norm.eps)
    torch.autograd.backward([sr_, yr], [gs.float(), gy.float()])
    _close(y, yr, dt, msg="norm output")
    _close(ag.grad, ar.grad, dt, scale=float(ar.grad.abs().max()) + 1e-6, msg="gradient of a")
    assert torch.equal(ag.grad, bg.grad)


def test_layernorm_fallbacks():
    x = torch.randn(4, 10, 100, device=DEV)  # K % 8 != 0
    n = nn.LayerNorm(100).to(DEV).requires_grad_(False)
    torch.testing.assert_close(fused.layer_norm(x, n), n(x))
    assert not _C.layernorm_supported(100) and not _C.layernorm_supported(4096) and _C.layernorm_supported(2560)
    t = nn.LayerNorm(64).to(DEV)  # trainable affine -> ATen (its gradients must exist)
    fused.layer_norm(torch.randn(3, 64, device=DEV, requires_grad=True), t).sum().backward()
    assert t.weight.grad is not None


def test_transformer2d_linear_projection_equals_conv(monkeypatch):
    from lora_amd.standin.unet import Transformer2DModel

    torch.manual_seed(0)
    m = Transformer2DModel(64, 2, 32, groups=8).to(DEV).requires_grad_(False)
    x = torch.randn(2, 64, 8, 8, device=DEV, requires_grad=True)
    ctx = torch.randn(2, 7, 32, device=DEV)
    y1 = m(x, ctx)
    (g1,) = torch.autograd.grad(y1.square().sum(), x)
    y0 = m(x.contiguous(memory_format=torch.channels_last), ctx)  # channels_last keeps the convolutions
    (g0,) = torch.autograd.grad(y0.square().sum(), x)
    torch.testing.assert_close(y1, y0.contiguous(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(g1, g0.contiguous(), rtol=1e-3, atol=1e-4)


def test_geglu_tails_and_fallback():
    # gelu tails: the erf approximation must not leak (gate -> -inf gives 0, +inf gives h)
    gate = torch.tensor([-30.0, -8.0, -1e-3, 0.0, 1e-3, 8.0, 30.0, 1.0], device=DEV)
    y = torch.cat([torch.full((8,), 2.0, device=DEV), gate]).view(1, 16)
    torch.testing.assert_close(fused.geglu(y), 2.0 * F.gelu(gate).view(1, 8), rtol=1e-5, atol=3e-7)
    odd = torch.randn(3, 20, device=DEV)  # inner = 10: not 16-byte friendly -> ATen sequence
    h, gt = odd.chunk(2, dim=-1)
    torch.testing.assert_close(fused.geglu(odd), h * F.gelu(gt))
    with pytest.raises(RuntimeError):
        _C.geglu_fwd(odd)


def test_host_passes_match_golden_vectors_and_oracle():
    """HIP passes on the committed torch-CPU vectors (tests/golden/hostops_cases.npz) and against the numpy oracle."""
    import os

    import numpy as np

    from oracle import lora_numpy as O
    from tests import helpers as H

    d = np.load(os.path.join(H.GOLDEN, "hostops_cases.npz"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    for i in range(3):
        groups, act = (int(v) for v in d[f"gn{i}_meta"])
        x = dev(d[f"gn{i}_x"]).requires_grad_(True)
        norm = nn.GroupNorm(groups, x.shape[1]).to(DEV)
        with torch.no_grad():
            norm.weight.copy_(dev(d[f"gn{i}_w"]))
            norm.bias.copy_(dev(d[f"gn{i}_b"]))
        norm.requires_grad_(False)
        assert fused._gn_native(x, norm)
        y = fused.group_norm_act(x, norm, bool(act))
        y.backward(dev(d[f"gn{i}_go"]))
        _close(y, torch.from_numpy(d[f"gn{i}_y"]), torch.float32, msg=f"gn{i} forward vs golden")
        _close(x.grad, torch.from_numpy(d[f"gn{i}_dx"]), torch.float32, scale=float(np.abs(d[f"gn{i}_dx"]).max()),
               msg=f"gn{i} dx vs golden")
        yo, cache = O.group_norm_act(d[f"gn{i}_x"], groups, d[f"gn{i}_w"], d[f"gn{i}_b"], 1e-5, bool(act))
        _close(y, torch.from_numpy(yo), torch.float32, msg=f"gn{i} forward vs oracle")
    for i in range(2):
        x = dev(d[f"ln{i}_x"]).requires_grad_(True)
        ln = nn.LayerNorm(x.shape[-1]).to(DEV)
        with torch.no_grad():
            ln.weight.copy_(dev(d[f"ln{i}_w"]))
            ln.bias.copy_(dev(d[f"ln{i}_b"]))
        ln.requires_grad_(False)
        y = fused.layer_norm(x, ln)
        assert type(y.grad_fn).__name__ == "_LayerNormBackward"
        y.backward(dev(d[f"ln{i}_go"]))
        _close(y, torch.from_numpy(d[f"ln{i}_y"]), torch.float32, msg=f"ln{i} forward vs golden")
        _close(x.grad, torch.from_numpy(d[f"ln{i}_dx"]), torch.float32, scale=float(np.abs(d[f"ln{i}_dx"]).max()),
               msg=f"ln{i} dx vs golden")
    for i in range(2):
        yin = dev(d[f"gg{i}_y"]).requires_grad_(True)
        out = fused.geglu(yin)
        assert type(out.grad_fn).__name__ == "_GegluBackward"
        out.backward(dev(d[f"gg{i}_go"]))
        _close(out, torch.from_numpy(d[f"gg{i}_out"]), torch.float32, msg=f"geglu{i} forward vs golden")
        _close(yin.grad, torch.from_numpy(d[f"gg{i}_dy"]), torch.float32, scale=float(np.abs(d[f"gg{i}_dy"]).max()),
               msg=f"geglu{i} gradient vs golden")
        _close(out, torch.from_numpy(O.geglu(d[f"gg{i}_y"])), torch.float32, msg=f"geglu{i} vs oracle")


def test_standin_unet_same_loss_and_gradients_with_and_without_fusions(monkeypatch):
    """Tiny UNet in f32: the fused passes must reproduce the ATen sequence's loss and adapter gradients."""
    import lora_amd as L
    from lora_amd.standin import tiny_unet

    torch.manual_seed(0)
    unet = tiny_unet().to(DEV)
    unet.requires_grad_(False)
    L.inject_trainable_lora(unet, r=4)
    for _, _, up, down in _iter_adapters(unet):
        nn.init.normal_(up.weight, std=0.05)
    lat = torch.randn(2, 4, 16, 16, device=DEV)
    ctx = torch.randn(2, 7, 32, device=DEV)
    t = torch.tensor([10, 500], device=DEV)

    def run(enabled):
        monkeypatch.setattr(fused, "_ENABLED", enabled)
        for p in unet.parameters():
            p.grad = None
        loss = unet(lat, t, ctx).sample.float().pow(2).mean()
        loss.backward()
        grads = torch.cat([p.grad.flatten() for p in unet.parameters() if p.requires_grad])
        return loss.detach(), grads

    l1, g1 = run(True)
    l0, g0 = run(False)
    torch.testing.assert_close(l1, l0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(g1, g0, rtol=1e-3, atol=1e-5 * float(g0.abs().max()))


def _iter_adapters(model):
    from lora_amd.lora import LoraInjectedLinear

    for name, m in model.named_modules():
        if isinstance(m, LoraInjectedLinear):
            yield name, m, m.lora_up, m.lora_down
