"""K4, channels-last form (csrc/conv_nhwc.hip) through the C-ABI vs the numpy oracle.

What is compared.  The three MFMA kernels multiply 16-bit operands exactly and accumulate in f32, so against the oracle
evaluated on the SAME operands (factor / Gt rounded to the activation dtype where the kernel rounds them: the forward
factor at rank > 8, the input-gradient factor, Gt in both gradients) the bound is the f32-sum bound of
tests/test_gpu_kernels.py (2e-5 x sum|terms|) plus one output rounding for dX.  The rounding of the operands itself is
what the reference's autocast does to lora_down.weight and to the gradient of lora_up's input (bf16 convolutions);
the module-level tests bound the total against the reference's f32 op sequence at the tolerance of the other bf16
module tests.  At rank <= 8 the forward keeps the f32 factor (hi + lo fragment rows) and is compared unrounded.
"""
import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C, ops
from oracle import lora_numpy as O
from oracle import torch_ref as TR
from tests.test_gpu_kernels import DT, close, n, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def rows_to_nchw(t, B, H, W):
    """[B*H*W, r] -> numpy [B, r, H, W]."""
    return n(t).reshape(B, H, W, -1).transpose(0, 3, 1, 2)


def rounded(t, dt):
    return t.to(DT[dt]).float()


CASES = [
    # B, C, H, W, r, dt
    (2, 64, 16, 16, 4, "bf16"), (1, 128, 8, 32, 16, "bf16"), (2, 64, 24, 24, 8, "f16"), (3, 64, 5, 7, 12, "bf16"),
    (1, 192, 12, 12, 16, "f16"), (4, 320, 32, 32, 4, "bf16"), (1, 64, 1, 1, 4, "bf16"), (1, 64, 3, 70, 16, "bf16"),
]


@pytest.mark.parametrize("B,C,Hh,Ww,r,dt", CASES)
def test_conv3_nhwc_kernels_match_oracle(B, C, Hh, Ww, r, dt):
    plan = _C.conv3_nhwc_plan(B, C, Hh, Ww, r)
    assert plan.native == 1
    x = cl(rnd((B, C, Hh, Ww), dt, seed=1))
    down = rnd((r, C, 3, 3), "f32", 0.2, seed=3)
    pf, pd = _C.conv3_nhwc_pack(down, DT[dt], plan)
    eye = np.eye(r, dtype=np.float32).reshape(r, r, 1, 1)
    geom = ((1, 1), (1, 1), (1, 1))
    # ---- T = conv3x3(X; down): f32-accurate factor at r <= 8, activation-dtype factor above
    t = _C.conv3_nhwc_down_fwd(x, pf, r)
    assert t.shape == (B * Hh * Ww, r)
    down_f = down if r <= 8 else rounded(down, dt)
    _, t_o = O.lora_conv2d_forward(n(x), None, None, n(down_f), eye, 1.0, *geom)
    absx = np.abs(n(x)).max() * np.abs(n(down)).sum(axis=(1, 2, 3)).max()
    close(rows_to_nchw(t, B, Hh, Ww), t_o, absx, "f32", k=3e-5, msg="T")
    # ---- gradients for a given Gt [B*H*W, r]
    gt = rnd((B * Hh * Ww, r), "f32", 1.0, seed=5)
    gt_nchw = rounded(gt, dt).reshape(B, Hh, Ww, r).permute(0, 3, 1, 2).contiguous()
    down_r = rounded(down, dt)
    dx0 = cl(rnd((B, C, Hh, Ww), dt, seed=8))
    dx = dx0.clone(memory_format=torch.preserve_format)
    _C.conv3_nhwc_bwd_dx_(dx, gt, pd)
    dxo, _, _ = O.lora_conv2d_backward(n(gt_nchw), n(x), None, n(down_r), eye, 1.0, *geom)
    absg = np.abs(n(gt)).max() * np.abs(n(down)).sum(axis=(0, 2, 3)).max()
    close(n(dx), dxo + n(dx0), absg + np.abs(n(dx0)), dt, k=3e-5, msg="dX")
    part = torch.full((int(plan.down_part_floats),), float("nan"), device=DEV)
    _C.conv3_nhwc_bwd_down(x, gt, part)
    d_down = torch.empty((r, C, 3, 3), device=DEV)
    table, nn_, total = _C.make_reduce_table([(part, d_down, plan.nsplit, plan.rank_pad, C * 9, r, _C.FACTOR_RK, 1.0, 0.0)],
                                             DEV)
    _C.reduce_batched(table, nn_, total)
    _, ddo, _ = O.lora_conv2d_backward(n(gt_nchw), n(x), None, n(down), eye, 1.0, *geom)
    kk = 1e-4
    np.testing.assert_allclose(n(d_down), ddo, rtol=kk * 10, atol=kk * np.abs(ddo).max() + 1e-6, err_msg="dDown")


def test_sum_parts():
    part = rnd((5, 1000 * 8), "f32", seed=2)
    out = _C.sum_parts(part, 5, 8000)
    np.testing.assert_allclose(n(out), n(part).sum(0), rtol=1e-6, atol=1e-6)


def _module_case(Ci, Co, ks, r, Hh, Ww, B, dt, use_sel=False, p=0.0, sink=False):
    torch.manual_seed(0)
    m = L.LoraInjectedConv2d(Ci, Co, ks, 1, (ks - 1) // 2, r=r, dropout_p=p, scale=0.8)
    m.conv.weight.data.mul_(0.5)
    m.lora_up.weight.data.normal_(0, 0.05)
    if use_sel:
        m.set_selector_from_diag(torch.linspace(0.5, 1.5, r))
    x_c = torch.randn(B, Ci, Hh, Ww).to(DT[dt]).float()
    gy_c = torch.randn(B, Co, Hh, Ww).to(DT[dt]).float()
    w16 = m.conv.weight.detach().to(DT[dt]).float()
    b16 = m.conv.bias.detach().to(DT[dt]).float()
    xr = x_c.clone().requires_grad_(True)
    dn = m.lora_down.weight.detach().clone().requires_grad_(True)
    upw = m.lora_up.weight.detach().clone().requires_grad_(True)
    sel = torch.diag(torch.linspace(0.5, 1.5, r)) if use_sel else None
    yr = TR.conv_adapter_forward(xr, w16, b16, dn, upw, 0.8, 1, (ks - 1) // 2, 1, 1, selector=sel) if use_sel else \
        TR.conv_adapter_forward(xr, w16, b16, dn, upw, 0.8, 1, (ks - 1) // 2, 1, 1)
    (yr * gy_c).sum().backward()
    m.to(DEV)
    m.conv.to(DT[dt])
    x = cl(x_c.to(DEV).to(DT[dt])).requires_grad_(True)
    return m, x, gy_c, (yr, xr.grad, dn.grad, upw.grad)


@pytest.mark.parametrize("Ci,Co,ks,r,Hh,Ww,B,dt", [
    (320, 320, 3, 16, 32, 32, 2, "bf16"),    # ResnetBlock2D conv, extended LoRA rank 16
    (1280, 1280, 3, 16, 12, 12, 1, "bf16"),  # the 12x12 maps of 768^2 images: native in this form
    (640, 320, 3, 4, 16, 16, 2, "f16"),
    (320, 640, 1, 8, 24, 24, 1, "bf16"),     # conv_shortcut: the Linear adapter on the pixel rows
])
def test_channels_last_module_matches_reference_ops(Ci, Co, ks, r, Hh, Ww, B, dt):
    """LoraInjectedConv2d on channels_last activations vs the reference's op sequence in f32 on the CPU
    (oracle/torch_ref.conv_adapter_forward, lora.py:130-135 and its autograd)."""
    m, x, gy_c, (yr, dxr, ddr, dur) = _module_case(Ci, Co, ks, r, Hh, Ww, B, dt)
    y = m(x)
    assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == (B, Co, Hh, Ww)
    (y.float() * gy_c.to(DEV)).sum().backward()
    e = 2.0 ** -7 if dt == "bf16" else 2.0 ** -9
    assert (n(y) - n(yr)).__abs__().max() <= e * float(yr.abs().max()) + 1e-3
    assert np.abs(n(x.grad) - n(dxr)).max() <= 2 * e * float(dxr.abs().max())
    for name, got, want in (("ddown", m.lora_down.weight.grad, ddr), ("dup", m.lora_up.weight.grad, dur)):
        err = np.abs(n(got) - n(want)).max() / float(want.abs().max())
        assert got.shape == want.shape and err <= 4e-3, (name, err)


def test_channels_last_module_equals_nchw_module_with_selector_and_sink():
    """The same site through both layouts (selector set, gradients leaving through a trainer's GradSink): the NHWC form
    must agree with the NCHW kernels of conv.hip to bf16 rounding — they share the oracle."""
    from lora_amd import trainer as T

    res = {}
    for layout in ("nchw", "nhwc"):
        torch.manual_seed(3)
        m = L.LoraInjectedConv2d(128, 64, 3, 1, 1, r=8, dropout_p=0.0, scale=0.8)
        m.lora_up.weight.data.normal_(0, 0.05)
        m.set_selector_from_diag(torch.linspace(0.5, 1.5, 8))
        m.to(DEV)
        m.conv.to(torch.bfloat16)
        st = T.FlatLoraState([{"params": [m.lora_up.weight, m.lora_down.weight], "lr": 1e-3}], device=torch.device(DEV))
        assert st.attach_direct_grads(m) == 1
        x = torch.randn(2, 128, 16, 16, device=DEV).to(torch.bfloat16)
        gy = torch.randn(2, 64, 16, 16, device=DEV).to(torch.bfloat16)
        if layout == "nhwc":
            x, gy = cl(x), cl(gy)
        x.requires_grad_(True)
        y = m(x)
        y.backward(gy)
        st.reduce_pending()
        assert float(st.flat_g.abs().max()) > 0
        res[layout] = (n(y), n(x.grad), n(st.flat_g).copy())
    for a, b, tol in zip(res["nchw"], res["nhwc"], (2.0 ** -7, 2.0 ** -6, 4e-3)):
        assert np.abs(a - b).max() <= tol * np.abs(a).max() + 1e-6


# (this test's time goes to the numpy oracle's im2col at these sizes: four sites, not every one)
REAL_SITES = [(1, 320, 96, 96), (1, 2560, 24, 24), (1, 1280, 12, 12), (4, 1280, 8, 8)]


@pytest.mark.parametrize("B,C,Hh,Ww", REAL_SITES)
def test_conv3_nhwc_real_size_sites_rank16(B, C, Hh, Ww):
    """The true channel counts / maps of configs[3] (768^2, batch 1) and of configs[1]-sized batches, bf16, rank 16."""
    r, dt = 16, "bf16"
    plan = _C.conv3_nhwc_plan(B, C, Hh, Ww, r)
    assert plan.native == 1
    x = cl(rnd((B, C, Hh, Ww), dt, seed=1))
    down = rnd((r, C, 3, 3), "f32", 0.05, seed=3)
    pf, pd = _C.conv3_nhwc_pack(down, DT[dt], plan)
    eye = np.eye(r, dtype=np.float32).reshape(r, r, 1, 1)
    geom = ((1, 1), (1, 1), (1, 1))
    t = _C.conv3_nhwc_down_fwd(x, pf, r)
    down_r = rounded(down, dt)
    _, t_o = O.lora_conv2d_forward(n(x), None, None, n(down_r), eye, 1.0, *geom)
    absx = np.abs(n(x)).max() * np.abs(n(down)).sum(axis=(1, 2, 3)).max()
    close(rows_to_nchw(t, B, Hh, Ww), t_o, absx, "f32", k=3e-5, msg="T")
    gt = rnd((B * Hh * Ww, r), "f32", 1.0, seed=5)
    gt_nchw = rounded(gt, dt).reshape(B, Hh, Ww, r).permute(0, 3, 1, 2).contiguous()
    dx0 = cl(rnd((B, C, Hh, Ww), dt, seed=8))
    dx = dx0.clone(memory_format=torch.preserve_format)
    _C.conv3_nhwc_bwd_dx_(dx, gt, pd)
    part = torch.full((int(plan.down_part_floats),), float("nan"), device=DEV)
    _C.conv3_nhwc_bwd_down(x, gt, part)
    d_down = torch.empty((r, C, 3, 3), device=DEV)
    table, nn_, total = _C.make_reduce_table([(part, d_down, plan.nsplit, plan.rank_pad, C * 9, r, _C.FACTOR_RK, 1.0, 0.0)],
                                             DEV)
    _C.reduce_batched(table, nn_, total)
    dxo, _, _ = O.lora_conv2d_backward(n(gt_nchw), n(x), None, n(down_r), eye, 1.0, *geom)
    _, ddo, _ = O.lora_conv2d_backward(n(gt_nchw), n(x), None, n(down), eye, 1.0, *geom)
    absg = np.abs(n(gt)).max() * np.abs(n(down)).sum(axis=(0, 2, 3)).max()
    close(n(dx), dxo + n(dx0), absg + np.abs(n(dx0)), dt, k=3e-5, msg="dX")
    kk = 2e-4
    np.testing.assert_allclose(n(d_down), ddo, rtol=kk * 10, atol=kk * np.abs(ddo).max() + 1e-6, err_msg="dDown")


# ----------------------------------------------------------------------------- round 6: launch fusion of the 3x3 site
FUSED_SITES = [
    # B, Ci, Co, H, W, r, p          geometry exercised
    (1, 320, 320, 96, 96, 16, 0.1),   # configs[3]'s 96^2 maps: pt 2, one channel share per tile
    (1, 640, 1280, 24, 24, 16, 0.1),  # small map: channel shares, the last arriver folds and projects
    (1, 1280, 1280, 12, 12, 8, 0.0),  # rank 8 (hi + lo fragment rows of `down`), 12 tiles x 10 shares
    (4, 320, 640, 64, 64, 4, 0.25),   # configs[1]-sized batch: pt 4
    (2, 64, 96, 20, 20, 12, 0.1),     # ragged tile edges (20 = 16 + 4 columns), C_out = 3 column groups
]


@pytest.mark.parametrize("B,Ci,Co,Hh,Ww,r,p", FUSED_SITES)
def test_conv3_fused_forward_one_launch_vs_oracle_and_vs_the_launch_sequence(B, Ci, Co, Hh, Ww, r, p):
    """lora.py:130-135's branch as ONE launch (lora_amd_conv3_nhwc_fwd_fused on the packs of
    lora_amd_conv3_nhwc_pack_batched): T against the f64 convolution of the oracle, Y = Y0 + s mask o (T up^T) against numpy
    with the Philox mask tests/helpers restates, and against rounds 3-5's launch sequence (pack + down + rank_update) on the
    same inputs.  Run twice: the arrival counters reset themselves."""
    from tests import helpers as H

    dt, s_, seed, off = "bf16", 0.8, 0x5EED0042, 991
    plan = _C.conv3_nhwc_plan(B, Ci, Hh, Ww, r)
    assert plan.native == 1 and _C.conv3_nhwc_fused_ok(torch.empty(B, Ci, Hh, Ww, dtype=torch.bfloat16), Co, r)
    M = B * Hh * Ww
    x = cl(rnd((B, Ci, Hh, Ww), dt, seed=1))
    y0 = cl(rnd((B, Co, Hh, Ww), dt, seed=2))
    down, up = rnd((r, Ci, 3, 3), "f32", 0.1, seed=3), rnd((Co, r), "f32", 0.05, seed=4)
    # the batched pack (two sites in one launch) = the per-site pack, bit for bit
    pf = torch.empty(int(plan.pf_elems), dtype=torch.bfloat16, device=DEV)
    pd = torch.empty(int(plan.pd_elems), dtype=torch.bfloat16, device=DEV)
    pu = torch.empty(Co * 32, dtype=torch.bfloat16, device=DEV)
    other = [torch.empty_like(pf), torch.empty_like(pd), torch.empty_like(pu)]
    arr, total = _C.conv3_nhwc_pack_table([(down * 2, up * 3, *other), (down, up, pf, pd, pu)])
    _C.conv3_nhwc_pack_batched(_C.table_to_device(arr, DEV), 2, total, torch.bfloat16)
    pf1, pd1 = _C.conv3_nhwc_pack(down, torch.bfloat16, plan)
    assert torch.equal(pf, pf1) and torch.equal(pd, pd1)
    t_part = torch.full((max(int(plan.t_part_floats), 4),), float("nan"), device=DEV)
    counters = torch.zeros(int(plan.fwd_tiles), dtype=torch.int32, device=DEV)
    ys = []
    for _ in range(2):
        y = y0.clone(memory_format=torch.preserve_format)
        t = _C.conv3_nhwc_fwd_fused_(x, pf, pu, y, r, s_, t_part, counters, p, seed, off)
        ys.append((n(y), n(t)))
        assert int(counters.abs().max()) == 0
    assert np.array_equal(ys[0][0], ys[1][0]) and np.array_equal(ys[0][1], ys[1][1])
    # ---- T
    eye = np.eye(r, dtype=np.float32).reshape(r, r, 1, 1)
    geom = ((1, 1), (1, 1), (1, 1))
    down_f = down if r <= 8 else rounded(down, dt)
    _, t_o = O.lora_conv2d_forward(n(x), None, None, n(down_f), eye, 1.0, *geom)
    absx = np.abs(n(x)).max() * np.abs(n(down)).sum(axis=(1, 2, 3)).max()
    close(rows_to_nchw(t, B, Hh, Ww), t_o, absx, "f32", k=3e-5, msg="T")
    # ---- Y: rows [M, Co] of the channels-last tensor
    mask = H.philox_dropout_mask(M * Co, p, seed, off).view(M, Co).double().numpy() if p > 0 else np.ones((M, Co))
    T64 = n(t).astype(np.float64)
    want = n(y0).transpose(0, 2, 3, 1).reshape(M, Co) + s_ * mask * (T64 @ n(up).astype(np.float64).T)
    got = ys[0][0].transpose(0, 2, 3, 1).reshape(M, Co)
    bound = np.abs(n(y0)).transpose(0, 2, 3, 1).reshape(M, Co) + s_ * mask * (np.abs(T64) @ np.abs(n(up)).T)
    close(got, want, bound, dt, k=3e-5, msg="Y")
    # ---- rounds 3-5's launch sequence on the same inputs
    y_old = y0.clone(memory_format=torch.preserve_format)
    t_old = _C.conv3_nhwc_down_fwd(x, pf1, r)
    _C.rank_update_(y_old.permute(0, 2, 3, 1).view(M, Co), t_old, up, _C.FACTOR_KR, s_, p, seed, off)
    assert np.array_equal(n(t_old), ys[0][1])   # same partial sums in the same order
    assert np.abs(n(y_old) - ys[0][0]).max() <= 2.0 ** -7 * np.abs(n(y_old)).max()


@pytest.mark.parametrize("M,N,r,p", [(9216, 320, 16, 0.1), (576, 1280, 16, 0.1), (144, 1280, 12, 0.0), (2304, 640, 16, 0.25)])
def test_g_pass_with_the_fold_inside_the_launch_equals_bwd_g_plus_sum_parts(M, N, r, p):
    """lora_amd_linear_bwd_g_folded (Gt column-tile partials folded by the last-arriving workgroup of a row block) against
    lora_amd_linear_bwd_g + lora_amd_sum_parts: Gt and the dUp partials bit for bit (same tiles summed in the same order);
    twice (the counters reset themselves)."""
    g, t = rnd((M, N), "bf16", seed=1), rnd((M, r), "f32", 0.5, seed=2)
    up = rnd((N, r), "f32", 0.05, seed=3)
    lp = _C.linear_plan(M, 320, N, r)
    assert lp.fused and _C.linear_bwd_g_folded_ok(g, up, r)
    f = lambda k: torch.full((max(int(k), 4),), float("nan"), device=DEV)  # noqa: E731
    gp0, up0 = f(lp.gt_part_floats), f(lp.up_part_floats)
    _C.linear_bwd_g(g, t, up, gp0, up0, 0.7, p, 77, 5)
    gt0 = _C.sum_parts(gp0, lp.nct_g, M * r)
    counters = torch.zeros(_C.linear_bwd_g_blocks(M, N, r), dtype=torch.int32, device=DEV)
    for _ in range(2):
        gp1, up1, gt1 = f(lp.gt_part_floats), f(lp.up_part_floats), f(M * r)
        _C.linear_bwd_g_folded(g, t, up, gp1, gt1, counters, up1, 0.7, p, 77, 5)
        assert torch.equal(gt1[:M * r], gt0) and torch.equal(up1[:int(lp.up_part_floats)], up0[:int(lp.up_part_floats)])
        assert int(counters.abs().max()) == 0


@pytest.mark.parametrize("fused", [True, False])
def test_channels_last_module_fused_and_unfused_paths_match_reference_ops(fused, monkeypatch):
    """The module (LoraInjectedConv2d, dropout 0, gradients through a trainer's GradSink so that the per-step pack table is in
    play, two optimiser steps so that the packs are re-made from the UPDATED factors) on the one-launch path and on rounds
    3-5's launch sequence: same outputs and gradients against the reference's op sequence."""
    from lora_amd import trainer as T

    monkeypatch.setattr(ops, "CONV3_FUSED", fused)
    Ci, Co, r, B, Hh, Ww = 320, 640, 16, 1, 24, 24
    torch.manual_seed(0)
    m = L.LoraInjectedConv2d(Ci, Co, 3, 1, 1, r=r, dropout_p=0.0, scale=0.8)
    m.conv.weight.data.mul_(0.5)
    m.lora_up.weight.data.normal_(0, 0.05)
    m.to(DEV)
    m.conv.to(torch.bfloat16)
    st = T.FlatLoraState([{"params": [m.lora_up.weight, m.lora_down.weight], "lr": 1e-2}], device=torch.device(DEV))
    assert st.attach_direct_grads(m) == 1
    x_c, gy_c = torch.randn(B, Ci, Hh, Ww).bfloat16().float(), torch.randn(B, Co, Hh, Ww).bfloat16().float()
    for step in range(2):
        w16, b16 = m.conv.weight.detach().float().cpu(), m.conv.bias.detach().float().cpu()
        dn = m.lora_down.weight.detach().cpu().clone().requires_grad_(True)
        upw = m.lora_up.weight.detach().cpu().clone().requires_grad_(True)
        xr = x_c.clone().requires_grad_(True)
        yr = TR.conv_adapter_forward(xr, w16, b16, dn, upw, 0.8, 1, 1, 1, 1)
        (yr * gy_c).sum().backward()
        x = cl(x_c.to(DEV).bfloat16()).requires_grad_(True)
        y = m(x)
        (y.float() * gy_c.to(DEV)).sum().backward()
        st.reduce_pending()
        e = 2.0 ** -7
        assert (n(y) - n(yr)).__abs__().max() <= e * float(yr.abs().max()) + 1e-3, step
        assert np.abs(n(x.grad) - n(xr.grad)).max() <= 2 * e * float(xr.grad.abs().max()), step
        gu, gd = st.flat_g[:Co * r].view(Co, r), st.flat_g[Co * r:].view(r, Ci, 3, 3)
        for name, got, want in (("dup", gu, upw.grad.view(Co, r)), ("ddown", gd, dn.grad)):
            err = np.abs(n(got) - n(want)).max() / float(want.abs().max())
            assert err <= 4e-3, (step, name, err)
        st.step(st.all_reduce())   # the factors move: step 1 must see the re-made packs
    reg = st.__dict__.get("_conv_packs")
    assert (reg is not None and len(reg.sites) == 1) if fused else reg is None
