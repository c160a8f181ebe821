"""Round-3 device parity ("next round" items 1 and 7 + ADVICE):

* the configuration ``bench.py`` actually times — SD1.5-size UNet, bf16, channels_last, head-padded projections, grouped
  q/k/v, hostops passes, hipGraph — against ``oracle/torch_ref.dreambooth_step`` (f32, CPU, the reference's op
  sequence) on the same weights / noise / timesteps, and again with every host-model option off;
* the fused dropout kernels (weight-stationary forward, input gradient, one-launch factor gradient) DIRECTLY against
  the numpy oracle with the mask the kernels generate, not against other HIP kernels;
* SVD distillation of the largest conv site (1280 x 23040 = [1280, 2560, 3, 3]);
* ranks above 64 with dropout (Linear and NCHW conv adapters);
* a 2-rank RCCL test of the flat-gradient all-reduce + step (skipped below 2 GPUs).
Everything goes through the C-ABI (``lora_amd/_C.py``)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C
from lora_amd import trainer as T
from lora_amd.standin import DDPMScheduler, sd15_unet
from oracle import lora_numpy as O
from oracle import torch_ref as TR
from tests import helpers as H
from tests.test_gpu_kernels import close, n, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mask(M, N, p, seed, off):
    """The dropout multiplier (0 or 1/(1-p)) the kernels apply for (seed, offset): a rank-1 update of zeros."""
    mk = torch.zeros(M, N, device=DEV)
    _C.rank_update_(mk, torch.ones(M, 1, device=DEV), torch.ones(1, N, device=DEV), _C.FACTOR_RK, 1.0, p, seed, off)
    return n(mk)


# ----------------------------------------------------------------------------- fused dropout kernels vs the oracle
@pytest.mark.parametrize("M,K,N,r,p", [(4096, 320, 320, 16, 0.1), (2048, 640, 640, 4, 0.25), (308, 768, 320, 8, 0.1),
                                       (1000, 320, 2560, 16, 0.1)])
def test_ws_dropout_forward_vs_oracle_with_extracted_mask(M, K, N, r, p):
    """lora.py:53-58 with nn.Dropout(p) on the branch, through lora_amd_linear_ws, vs oracle.lora_linear_forward(mask=)."""
    dt, s, seed = "bf16", 0.9, 4321
    x, w, b = rnd((M, K), dt, seed=1), rnd((N, K), dt, 0.05, seed=2), rnd((N,), dt, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.5, seed=5)
    off = torch.tensor([977], dtype=torch.int64, device=DEV)
    mask = _mask(M, N, p, seed, off)
    drop = float((mask == 0).mean())
    assert abs(drop - p) < 0.02 and np.allclose(mask[mask != 0], 1.0 / (1.0 - p))
    y, t = _C.linear_ws_fwd(x, w, b, down, up, s, 0, p, seed, off)
    X, W, Bv, A, U = n(x), n(w), n(b), n(down), n(up)
    yo, to = O.lora_linear_forward(X, W, Bv, A, U, s, None, mask)
    close(n(t), to, np.abs(X) @ np.abs(A).T, "f32", k=3e-5, msg="T")
    # the kernel rounds T and s*up to bf16 before the rank-r MFMA (what autocast does to lora_down's output): each a
    # relative 2^-9 perturbation of the branch, on top of the single output rounding
    branch_abs = (np.abs(to) @ np.abs(s * U).T) * mask
    absref = np.abs(X) @ np.abs(W).T + np.abs(Bv) + branch_abs
    tol = 2e-3 * absref + 2.0 ** -8 * np.abs(yo) + 2.0 ** -7 * branch_abs
    err = np.abs(n(y) - yo)
    assert (err <= tol).all(), f"{(err > tol).sum()} outside tolerance, worst {err.max():.3e}"
    # dropped elements carry the frozen product only
    frozen = X @ W.T + Bv
    dropped = mask == 0
    assert np.abs(n(y) - frozen)[dropped].max() <= 2.0 ** -8 * np.abs(frozen)[dropped].max() + 2e-3 * absref[dropped].max()


@pytest.mark.parametrize("M,K,N,r,p", [(4096, 320, 320, 16, 0.1), (2048, 1280, 640, 4, 0.25), (576, 320, 1280, 16, 0.1)])
def test_ws_dropout_backward_vs_oracle_with_extracted_mask(M, K, N, r, p):
    """Autograd of the same site: Gt / dX from the weight-stationary input-gradient launch, dUp / dDown from
    lora_amd_linear_bwd_factors_drop, vs oracle.lora_linear_backward(mask=)."""
    dt, s, seed, off = "bf16", 0.8, 99, 31337
    g, x = rnd((M, N), dt, seed=1), rnd((M, K), dt, seed=6)
    w = rnd((N, K), dt, 0.05, seed=2)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    mask = _mask(M, N, p, seed, off)
    G, X, W, A, U = n(g), n(x), n(w), n(down), n(up)
    dxo, ddo, duo, _, _ = O.lora_linear_backward(G, X, W, A, U, s, None, mask)
    gt_o = s * ((G * mask) @ U)
    dx, gt = _C.linear_ws_dx(g, w, down, up, s, 0, p, seed, off)
    close(n(gt), gt_o, s * ((np.abs(G) * mask) @ np.abs(U)), "f32", k=3e-5, msg="Gt")
    absdx = np.abs(G) @ np.abs(W) + np.abs(gt_o) @ np.abs(A)
    close(n(dx), dxo, absdx, dt, k=2e-3, msg="dX")
    t = torch.from_numpy((X @ A.T).astype(np.float32)).to(DEV)  # the forward's saved T
    plan = _C.linear_plan(M, K, N, r)
    up_part, down_part = (torch.empty(max(int(k), 1), device=DEV) for k in (plan.up_part_floats, plan.down_part_floats))
    _C.linear_bwd_factors(g, t, up_part, x, gt, down_part, r, s, dropout=(p, seed, off))
    d_up, d_down = torch.empty((N, r), device=DEV), torch.empty((r, K), device=DEV)
    rows = [(up_part, d_up, plan.nparts_up, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
            (down_part, d_down, plan.nparts_down, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)]
    _C.reduce_batched(*_C.make_reduce_table(rows, DEV))
    close(n(d_up), duo, s * ((np.abs(G) * mask).T @ np.abs(X @ A.T)), "f32", k=3e-5, msg="dUp")
    close(n(d_down), ddo, np.abs(gt_o).T @ np.abs(X), "f32", k=3e-5, msg="dDown")


# ----------------------------------------------------------------------------- ranks above 64 with dropout
def test_rank128_dropout_linear_module_vs_oracle():
    """The reference accepts any r <= min(in, out) with dropout; on device the rank is chunked and the mask (which
    belongs to the SUM over ranks) is drawn once and applied to the full product."""
    torch.manual_seed(0)
    K, N, r, M, p, s = 256, 192, 128, 72, 0.1, 0.7
    m = L.LoraInjectedLinear(K, N, True, r=r, dropout_p=p, scale=s).to(DEV)
    m.lora_up.weight.data.normal_(0, 0.05)
    m.train()
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    g = torch.randn(M, N, device=DEV)
    torch.manual_seed(3)
    y = m(x)
    y.backward(g)
    # recover (seed, offset) of the forward: ops.next_dropout_stream draws the offset from torch's device generator
    torch.manual_seed(3)
    off = torch.empty(1, dtype=torch.int64, device=DEV).random_(0, 1 << 62)
    mask = _mask(M, N, p, int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, off)
    assert 0.03 < float((mask == 0).mean()) < 0.2
    X, W, Bv, A, U, Gn = (n(v) for v in (x, m.linear.weight, m.linear.bias, m.lora_down.weight, m.lora_up.weight, g))
    yo, _ = O.lora_linear_forward(X, W, Bv, A, U, s, None, mask)
    dxo, ddo, duo, _, _ = O.lora_linear_backward(Gn, X, W, A, U, s, None, mask)
    for got, want, nm in ((y, yo, "y"), (x.grad, dxo, "dx"), (m.lora_down.weight.grad, ddo, "ddown"),
                          (m.lora_up.weight.grad, duo, "dup")):
        # the frozen f32 GEMM may run on reduced-precision matrix cores
        assert np.abs(n(got) - want).max() <= 2e-3 * np.abs(want).max(), nm


def test_rank80_dropout_conv_nchw_branch_runs_and_masks_consistently():
    """LoraInjectedConv2d outside the native geometry (stride 2) with r > 64 and dropout 0.1 (extended injection keeps
    the constructor's 0.1): forward must not refuse, eval == reference op sequence, and the train-mode backward must
    use the forward's mask (finite-difference-free check: d/dup of <y, g> is linear in the same masked branch)."""
    torch.manual_seed(0)
    B, Ci, Co, Hh, r, s = 2, 96, 128, 8, 80, 0.5
    m = L.LoraInjectedConv2d(Ci, Co, 3, 2, 1, r=r, dropout_p=0.1, scale=s)
    m.lora_up.weight.data.normal_(0, 0.05)
    x_c = torch.randn(B, Ci, Hh, Hh)
    m.eval()
    want = TR.conv_adapter_forward(x_c, m.conv.weight, m.conv.bias, m.lora_down.weight, m.lora_up.weight, s, 2, 1, 1, 1)
    m.to(DEV)
    got = m(x_c.to(DEV))
    np.testing.assert_allclose(n(got), want.detach().numpy(), rtol=2e-3, atol=2e-3 * float(want.abs().max()))
    m.train()
    x = x_c.to(DEV).requires_grad_(True)
    torch.manual_seed(9)
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    frozen = torch.nn.functional.conv2d(x.detach(), m.conv.weight, m.conv.bias, 2, 1)
    branch = (y.detach() - frozen)
    dropped = float((branch.abs() < 1e-7).float().mean())
    assert 0.03 < dropped < 0.25
    # <branch, gy> = <up, dUp> because the masked branch is linear in up (Euler): the backward saw the same mask
    lhs = float((branch.double() * gy.double()).sum())
    rhs = float((m.lora_up.weight.detach().double() * m.lora_up.weight.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), 1.0)


# ----------------------------------------------------------------------------- f1: the largest conv site of configs[4]
def test_distill_largest_conv_site_on_device_vs_reference_recipe():
    """cli_svd.py:55-92 at up_blocks' 2560 -> 1280 3x3 conv: the residual [1280, 2560, 3, 3] is flattened from dim 1 to
    1280 x 23040 (the widest matrix of the 224 sites).  Same comparison as the Linear sites (tests/test_gpu_parity_r2.py):
    factors after sign alignment, the signed-quantile clamp threshold, the clamped product."""
    from lora_amd import cli_svd as S
    from tests.test_cli_svd import _planted
    from tests.test_gpu_parity_r2 import _reference_recipe

    N, Ci, r = 1280, 2560, 8
    K = Ci * 9
    tuned, base = _planted(N, K, r + 4, 2e-3 / (N ** 0.5 + K ** 0.5), 2, "cpu")
    res = (tuned - base).float()
    # the reference's recipe (cli_svd.py:57-74: exact SVD, top r, signed-quantile clamp).  LAPACK's thin SVD of this 1280 x
    # 23040 matrix takes the host 40 s; the SAME top-r triplets come from the f64 eigen-decomposition of the 1280 x 1280 Gram
    # matrix dW dW^T (exact to ~1e-13 here: sigma_8 / sigma_1 = 0.03), evaluated with library f64 kernels
    A = res.double().to(DEV)
    ev, Q = torch.linalg.eigh(A @ A.T)
    Sg = ev.flip(0)[:r].sqrt()
    Ur = Q.flip(1)[:, :r]
    U_ref = (Ur * Sg).float().cpu()
    Vh_ref = ((Ur.T @ A) / Sg[:, None]).float().cpu()
    hi_ref = float(torch.quantile(torch.cat([U_ref.flatten(), Vh_ref.flatten()]), 0.99))
    up_ref, down_ref = U_ref.clamp(-hi_ref, hi_ref), Vh_ref.clamp(-hi_ref, hi_ref)
    del A, Q
    t4, b4 = tuned.view(N, Ci, 3, 3).to(DEV), base.view(N, Ci, 3, 3).to(DEV)
    up, down = S.distill_pair(t4, b4, r, 0.99, generator=torch.Generator(device=DEV).manual_seed(0))
    assert up.shape == (N, r) and down.shape == (r, K) and up.is_cuda
    U, Sg, Vh = S.topr_svd(res.to(DEV), r, generator=torch.Generator(device=DEV).manual_seed(0))
    U, Vh = (U @ torch.diag(Sg)).cpu(), Vh.cpu()
    sgn = torch.sign((Vh * Vh_ref).sum(1))
    Ua, Vha = U * sgn[None, :], Vh * sgn[:, None]
    assert (Ua - U_ref).abs().max() <= 5e-3 * U_ref.abs().max()
    assert (Vha - Vh_ref).abs().max() <= 5e-3 * Vh_ref.abs().max()
    hi_aligned = float(torch.quantile(torch.cat([Ua.flatten(), Vha.flatten()]), 0.99))
    assert abs(hi_aligned - hi_ref) <= 5e-3 * hi_ref
    prod, prod_ref = (up @ down).cpu(), up_ref @ down_ref
    assert (prod - prod_ref).norm() <= 0.05 * prod_ref.norm()


# ----------------------------------------------------------------------------- a1/a2/a4: the merged-weight path
def _heads_pack(a, lay):
    h, d, D = lay
    out = np.zeros(a.shape[:-1] + (h * D,), dtype=a.dtype)
    out.reshape(a.shape[:-1] + (h, D))[..., :d] = a.reshape(a.shape[:-1] + (h, d))
    return out


@pytest.mark.parametrize("M,K,N,r,dt,gh,xh", [
    (16384, 320, 320, 4, "bf16", None, None), (4096, 640, 640, 8, "bf16", None, None),
    (1024, 1280, 1280, 16, "bf16", None, None), (256, 1280, 1280, 4, "f16", None, None),
    (308, 768, 320, 4, "bf16", None, None), (4096, 320, 2560, 4, "bf16", None, None),
    (1000, 1280, 10240, 4, "bf16", None, None), (2048, 320, 320, 4, "bf16", (8, 40, 64), None),
    (2048, 320, 320, 16, "bf16", None, (8, 40, 64)), (777, 64, 96, 4, "f32", None, None)])
def test_factors_self_kernel_vs_oracle(M, K, N, r, dt, gh, xh):
    """autograd of lora.py:53-58 for the factors (dB = s G^T (X A^T), dA = (s G B)^T X) through
    lora_amd_linear_bwd_factors_self — no T, no Gt supplied — vs oracle.lora_linear_backward; head-padded G / X too."""
    s = 0.7
    x, g = rnd((M, K), dt, seed=1), rnd((M, N), dt, seed=2)
    down, up = rnd((r, K), "f32", 0.2, seed=3), rnd((N, r), "f32", 0.3, seed=4)
    X, G, A, U = n(x), n(g), n(down), n(up)
    _, ddo, duo, _, _ = O.lora_linear_backward(G, X, np.zeros((N, K), np.float32), A, U, s)
    gd = torch.from_numpy(_heads_pack(G, gh)).to(DEV).to(g.dtype) if gh else g
    xd = torch.from_numpy(_heads_pack(X, xh)).to(DEV).to(x.dtype) if xh else x
    if gh:  # pad columns of a real G need not be zero: the kernel must not read them
        gd.view(M, gh[0], gh[2])[:, :, gh[1]:] = 7.0
    if xh:
        xd.view(M, xh[0], xh[2])[:, :, xh[1]:] = -3.0
    plan = _C.factors_self_plan(M, K, N, r)
    assert plan.supported
    up_part = torch.full((int(plan.up_part_floats),), float("nan"), device=DEV)
    down_part = torch.full((int(plan.down_part_floats),), float("nan"), device=DEV)
    _C.linear_bwd_factors_self(gd, xd, down, up, up_part, down_part, s, g_heads=gh, x_heads=xh)
    d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
    table, cnt, total = _C.make_reduce_table(
        [(up_part, d_up, plan.nparts, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
         (down_part, d_down, plan.nparts, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)], DEV)
    _C.reduce_batched(table, cnt, total)
    T_abs, Gt_abs = np.abs(X) @ np.abs(A).T, s * (np.abs(G) @ np.abs(U))
    close(n(d_up), duo, s * (np.abs(G).T @ T_abs), "f32", k=1e-4, msg="dUp")
    close(n(d_down), ddo, Gt_abs.T @ np.abs(X), "f32", k=1e-4, msg="dDown")


@pytest.mark.parametrize("N,K,r,heads", [(320, 320, 4, None), (1280, 320, 8, None), (320, 768, 16, None),
                                          (320, 320, 4, (40, 64)), (10240, 1280, 4, None)])
def test_merge_transposed_site_is_the_bitwise_transpose(N, K, r, heads):
    """lora.py:635-669 through ``lora_amd_merge_site.transposed``: W^T + alpha (up down)^T merged from the transposed
    frozen weight equals the transpose of the ordinary merge BIT FOR BIT (same fma chain per element), also when the
    transposed result's columns are head-padded (``out_heads`` on the transposed site = a head-padded OUTPUT of the
    adapter); and both equal oracle.collapse within one rounding."""
    w = rnd((N, K), "bf16", 0.05, seed=1)
    up, down = rnd((N, r), "f32", 0.3, seed=2), rnd((r, K), "f32", 0.3, seed=3)
    w_eff = torch.empty_like(w)
    _C.MergePlan([(w, w_eff, up, down)]).launch(0.7, _C.ROUND_ONCE)
    wt = w.t().contiguous()
    n_out = N if heads is None else (N // heads[0]) * heads[1]
    w_eff_t = torch.zeros(K, n_out, dtype=w.dtype, device=DEV)
    _C.MergePlan([(wt, w_eff_t, down, up, heads, True)]).launch(0.7, _C.ROUND_ONCE)
    got = w_eff_t
    if heads is not None:
        d, D = heads
        assert torch.all(w_eff_t.view(K, N // d, D)[:, :, d:] == 0)
        got = w_eff_t.view(K, N // d, D)[:, :, :d].reshape(K, N)
    assert torch.equal(got, w_eff.t())
    want = O.collapse(n(w), n(up), n(down), 0.7)
    assert np.abs(n(w_eff) - want).max() <= 2.0 ** -8 * np.abs(want).max()


@pytest.mark.parametrize("rows", [0, 32, 128])
def test_factors_self_ragged_one_launch_for_several_sites_vs_oracle(rows, monkeypatch):
    """lora_amd_linear_bwd_factors_self_ragged: the factor gradients of several sites of different shapes (one of them
    with head-padded G, one with head-padded X, one wider than a 256-chunk tile) in ONE launch over a device table, per
    site vs oracle.lora_linear_backward; ``rows`` = rows per block the caller asks for (0: the per-site default)."""
    monkeypatch.setattr(_C, "SELF_ROWS_DEFERRED", rows)
    r, s_ = 4, 0.9
    specs = [(4096, 320, 320, None, None), (1000, 640, 640, None, None), (2048, 320, 320, (8, 40, 64), None),
             (2048, 320, 320, None, (8, 40, 64)), (308, 768, 1280, None, None), (512, 320, 2560, None, None)]
    sites, refs = [], []
    for i, (M, K, N, gh, xh) in enumerate(specs):
        x, g = rnd((M, K), "bf16", seed=10 + i), rnd((M, N), "bf16", seed=30 + i)
        down, up = rnd((r, K), "f32", 0.2, seed=50 + i), rnd((N, r), "f32", 0.3, seed=70 + i)
        X, G, A, U = n(x), n(g), n(down), n(up)
        _, ddo, duo, _, _ = O.lora_linear_backward(G, X, np.zeros((N, K), np.float32), A, U, s_)
        gd = torch.from_numpy(_heads_pack(G, gh)).to(DEV).bfloat16() if gh else g
        xd = torch.from_numpy(_heads_pack(X, xh)).to(DEV).bfloat16() if xh else x
        plan = _C.factors_self_plan(M, K, N, r, rows)
        assert plan.supported
        up_part = torch.full((int(plan.up_part_floats),), float("nan"), device=DEV)
        down_part = torch.full((int(plan.down_part_floats),), float("nan"), device=DEV)
        sites.append((gd, xd, down, up, up_part, down_part, s_, gh, xh))
        refs.append((plan, N, K, duo, ddo, s_ * (np.abs(G).T @ (np.abs(X) @ np.abs(A).T)),
                     (s_ * np.abs(G) @ np.abs(U)).T @ np.abs(X)))
    arr, grid = _C.factors_self_ragged_table(sites, torch.bfloat16)
    dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    _C.linear_bwd_factors_self_ragged(dev, len(sites), grid, r, torch.bfloat16)
    for (gd, xd, down, up, up_part, down_part, _, _, _), (plan, N, K, duo, ddo, absu, absd) in zip(sites, refs):
        d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
        table, cnt, total = _C.make_reduce_table(
            [(up_part, d_up, plan.nparts, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
             (down_part, d_down, plan.nparts, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)], DEV)
        _C.reduce_batched(table, cnt, total)
        close(n(d_up), duo, absu, "f32", k=1e-4, msg=f"dUp {N}x{K}")
        close(n(d_down), ddo, absd, "f32", k=1e-4, msg=f"dDown {N}x{K}")


@pytest.mark.parametrize("in_heads,out_heads,r,bias", [(None, None, 4, True), (None, (8, 40, 64), 4, False),
                                                       ((8, 40, 64), None, 8, True), (None, None, 16, False)])
def test_merged_weight_adapter_forward_backward_vs_oracle(in_heads, out_heads, r, bias):
    """lora.py:53-58 + autograd with the adapter on ops.MergedWeights (K3 merge into the scratch weight, dense GEMMs on
    it, one launch for both factor gradients) vs oracle.lora_linear_forward / backward; head-padded layouts included:
    pad columns of the output and of dX must be exactly zero."""
    from lora_amd import ops

    M, K, N, s = 2048, 320, 320, 0.8
    torch.manual_seed(0)
    m = L.LoraInjectedLinear(K, N, bias, r=r, dropout_p=0.0, scale=s).to(DEV).to(torch.bfloat16)
    m.linear.requires_grad_(False)
    T.promote_lora_to_fp32(m)
    m.lora_up.weight.data.normal_(0, 0.05)
    m.__dict__["_merged"] = mw = ops.MergedWeights()
    x, gy = rnd((M, K), "bf16", seed=5), rnd((M, N), "bf16", seed=6)
    X, G = n(x), n(gy)
    W, A, U = n(m.linear.weight), n(m.lora_down.weight), n(m.lora_up.weight)
    Bv = n(m.linear.bias) if bias else None
    xd = (torch.from_numpy(_heads_pack(X, in_heads)).to(DEV).bfloat16() if in_heads else x).requires_grad_(True)
    gd = torch.from_numpy(_heads_pack(G, out_heads)).to(DEV).bfloat16() if out_heads else gy
    for rep in range(2):  # second pass: through refresh()'s batched plan instead of the entry's first merge
        mw.refresh()
        xd.grad = None
        m.lora_up.weight.grad = m.lora_down.weight.grad = None
        y = m.forward_heads(xd, in_heads, out_heads)
        y.backward(gd)
        yo, _ = O.lora_linear_forward(X, W, Bv, A, U, s)
        dxo, ddo, duo, _, _ = O.lora_linear_backward(G, X, W, A, U, s)
        yv, dxv = n(y), n(xd.grad)
        if out_heads:
            h, d, D = out_heads
            assert np.all(yv.reshape(M, h, D)[:, :, d:] == 0)
            yv = yv.reshape(M, h, D)[:, :, :d].reshape(M, N)
        if in_heads:
            h, d, D = in_heads
            assert np.all(dxv.reshape(M, h, D)[:, :, d:] == 0)
            dxv = dxv.reshape(M, h, D)[:, :, :d].reshape(M, K)
        absy = np.abs(X) @ (np.abs(W) + s * np.abs(U) @ np.abs(A)).T + (np.abs(Bv) if bias else 0.0)
        # W_eff is rounded to bf16 once (2^-9 relative on W + s U A), the GEMM output once
        assert np.all(np.abs(yv - yo) <= 2.0 ** -8 * absy + 2.0 ** -8 * np.abs(yo) + 1e-3), rep
        absdx = np.abs(G) @ (np.abs(W) + s * np.abs(U) @ np.abs(A))
        assert np.all(np.abs(dxv - dxo) <= 2.0 ** -8 * absdx + 2.0 ** -8 * np.abs(dxo) + 1e-3), rep
        close(n(m.lora_up.weight.grad), duo, s * (np.abs(G).T @ (np.abs(X) @ np.abs(A).T)), "f32", k=1e-4, msg="dUp")
        close(n(m.lora_down.weight.grad), ddo, (s * np.abs(G) @ np.abs(U)).T @ np.abs(X), "f32", k=1e-4, msg="dDown")
    assert mw.refreshes == 1 and len(mw.entries) == 1  # the first refresh() found no site yet


@pytest.mark.parametrize("out_heads", [None, (8, 40, 64)])
def test_merged_weight_group_of_sites_on_one_input_vs_oracle(out_heads):
    """to_q / to_k / to_v on one tensor through lora.lora_linear_group with the adapters on ops.MergedWeights: each output
    vs oracle.lora_linear_forward, the input gradient vs the SUM of the three oracle input gradients (accumulated inside
    the GEMMs), every factor gradient vs oracle.lora_linear_backward."""
    from lora_amd import ops

    M, K, N, r, s = 1024, 320, 320, 4, 1.0
    torch.manual_seed(1)
    mw = ops.MergedWeights()
    mods = []
    for i in range(3):
        m = L.LoraInjectedLinear(K, N, False, r=r, dropout_p=0.0, scale=s).to(DEV).to(torch.bfloat16)
        m.linear.requires_grad_(False)
        T.promote_lora_to_fp32(m)
        m.lora_up.weight.data.normal_(0, 0.05)
        m.__dict__["_merged"] = mw
        mods.append(m)
    x = rnd((M, K), "bf16", seed=7).requires_grad_(True)
    gs = [rnd((M, N), "bf16", seed=8 + i) for i in range(3)]
    outs = L.lora_linear_group(mods, x, out_heads=out_heads)
    assert outs is not None and len(outs) == 3
    gd = [torch.from_numpy(_heads_pack(n(g), out_heads)).to(DEV).bfloat16() if out_heads else g for g in gs]
    torch.autograd.backward(outs, gd)
    X = n(x)
    dx_sum, absdx = 0.0, 0.0
    for m, y, g in zip(mods, outs, gs):
        W, A, U, G = n(m.linear.weight), n(m.lora_down.weight), n(m.lora_up.weight), n(g)
        yo, _ = O.lora_linear_forward(X, W, None, A, U, s)
        dxo, ddo, duo, _, _ = O.lora_linear_backward(G, X, W, A, U, s)
        yv = n(y)
        if out_heads:
            h, d, D = out_heads
            assert np.all(yv.reshape(M, h, D)[:, :, d:] == 0)
            yv = yv.reshape(M, h, D)[:, :, :d].reshape(M, N)
        absy = np.abs(X) @ (np.abs(W) + s * np.abs(U) @ np.abs(A)).T
        assert np.all(np.abs(yv - yo) <= 2.0 ** -8 * absy + 2.0 ** -8 * np.abs(yo) + 1e-3)
        dx_sum = dx_sum + dxo
        absdx = absdx + np.abs(G) @ (np.abs(W) + s * np.abs(U) @ np.abs(A))
        close(n(m.lora_up.weight.grad), duo, s * (np.abs(G).T @ (np.abs(X) @ np.abs(A).T)), "f32", k=1e-4, msg="dUp")
        close(n(m.lora_down.weight.grad), ddo, (s * np.abs(G) @ np.abs(U)).T @ np.abs(X), "f32", k=1e-4, msg="dDown")
    # three bf16 roundings of the running sum (addmm_ accumulates in the output dtype)
    assert np.all(np.abs(n(x.grad) - dx_sum) <= 2.0 ** -8 * absdx + 3 * 2.0 ** -8 * np.abs(dx_sum) + 2e-3)


# ----------------------------------------------------------------------------- f1: every shape group in one launch
def test_ragged_svd_of_several_shape_groups_vs_exact_svd():
    """cli_svd.py:24-92 over a model = sites of several shapes.  ``topr_svd_ragged`` (one descriptor-table launch per
    step of the iteration for ALL groups) vs the exact SVD of every matrix: sign-free rank-r product, aligned vectors,
    singular values; then ``distill_model`` (one-launch residuals, f16 weights included) vs ``distill_group``."""
    from lora_amd import cli_svd as S
    from tests.test_cli_svd import _planted

    r = 8
    shapes = [(3, 320, 320), (1, 1280, 2880), (2, 640, 1280), (2, 2560, 320)]  # (sites, N, K); K = 2880: a 3x3 conv
    tuned, base = [], []
    for gi, (B, N, K) in enumerate(shapes):
        tb = [_planted(N, K, r + 4, 2e-3 / (N ** 0.5 + K ** 0.5), 10 * gi + i, "cpu") for i in range(B)]
        tuned.append([t.to(DEV) for t, _ in tb])
        base.append([b.to(DEV) for _, b in tb])
    deltas = [torch.stack([t - b for t, b in zip(ts, bs)]) for ts, bs in zip(tuned, base)]
    trip = S.topr_svd_ragged(deltas, r, generator=torch.Generator(device=DEV).manual_seed(0))
    assert len(trip) == len(shapes)
    for (B, N, K), d, (U, Sg, Vh) in zip(shapes, deltas, trip):
        assert U.shape == (B, N, r) and Sg.shape == (B, r) and Vh.shape == (B, r, K)
        for i in range(B):
            Ue, Se, Vhe = torch.linalg.svd(d[i].double().cpu(), full_matrices=False)
            ref = (Ue[:, :r] * Se[:r]) @ Vhe[:r]
            got = ((U[i] * Sg[i]) @ Vh[i]).double().cpu()
            assert (got - ref).norm() <= 2e-4 * ref.norm(), ((N, K, i), float((got - ref).norm() / ref.norm()))
            assert (Sg[i].double().cpu() - Se[:r]).abs().max() <= 1e-4 * float(Se[0])
            sgn = torch.sign((Vh[i].double().cpu() * Vhe[:r]).sum(1))
            assert (Vh[i].double().cpu() * sgn[:, None] - Vhe[:r]).abs().max() <= 5e-3 * float(Vhe[:r].abs().max())
            # the sign rule: the largest-magnitude entry of every down row is positive
            j = Vh[i].abs().argmax(dim=1)
            assert bool((Vh[i][torch.arange(r), j] > 0).all())
    # the model-level entry: grouped inputs, residuals formed by sub_ragged, clamp per site — vs the per-group path
    groups = list(zip(tuned, base))
    res = S.distill_model(groups, r, 0.99, torch.Generator(device=DEV).manual_seed(1))
    for (ts, bs), (up, down) in zip(groups, res):
        up1, down1 = S.distill_group(ts, bs, r, 0.99, torch.Generator(device=DEV).manual_seed(2))
        assert up.shape == up1.shape and down.shape == down1.shape
        assert (up - up1).abs().max() <= 2e-3 * up1.abs().max()
        assert (down - down1).abs().max() <= 2e-3 * down1.abs().max()
    # f16 weights of odd sizes through the one-launch subtraction
    a = [torch.randn(n_, device=DEV).half() for n_ in (4096 * 3 + 5, 8, 100000)]
    b = [torch.randn(n_, device=DEV).half() for n_ in (4096 * 3 + 5, 8, 100000)]
    o = [torch.empty(t.numel(), device=DEV) for t in a]
    _C.sub_ragged(list(zip(a, b)), o)
    for x, y, z in zip(a, b, o):
        assert torch.equal(z, x.float() - y.float())


# ----------------------------------------------------------------------------- the benchmarked configuration vs the oracle
def _sd15_twins(r=4, ref_device=DEV):
    """SD1.5-size UNet twice: bf16 on the device exactly as bench.py builds it, and the f32 oracle twin with the same
    (bf16-representable) frozen values and the reference-algorithm adapters; same factor values (up != 0).  The oracle twin
    lives on ``ref_device``: the GPU by default (its steps are then evaluated inside ``H.oracle_on_device()``: library
    kernels only, f32 — the host needs a minute per SD1.5-size step)."""
    sys.path.insert(0, H.REPO)
    from bench import build_unet

    dev_unet = build_unet(torch.device(DEV), torch.bfloat16, seed=0)
    with torch.device("meta"):
        ref = sd15_unet()
    ref.to_empty(device=ref_device)
    ref.load_state_dict({k: v.float().to(ref_device) for k, v in dev_unet.state_dict().items()})
    ref.requires_grad_(False)
    ref_params = TR.inject(ref, L.UNET_DEFAULT_TARGET_REPLACE, r=r)
    g = torch.Generator().manual_seed(11)
    for s_ in TR.sites_of(ref):
        s_.up.data.copy_((torch.randn(s_.up.shape, generator=g) * 0.02).to(ref_device))
        s_.down.data.copy_((torch.randn(s_.down.shape, generator=g) / r).to(ref_device))
    L.inject_trainable_lora(dev_unet, r=r)
    T.promote_lora_to_fp32(dev_unet)
    ours = [m for m in dev_unet.modules() if isinstance(m, L.LoraInjectedLinear)]
    theirs = TR.sites_of(ref)
    assert len(ours) == len(theirs) == 144
    for a, b in zip(ours, theirs):
        a.lora_up.weight.data.copy_(b.up.data.to(DEV))
        a.lora_down.weight.data.copy_(b.down.data.to(DEV))
    ref.train(), dev_unet.train()
    return ref, ref_params, dev_unet


@pytest.fixture(scope="module")
def sd15_reference_step():
    """One BATCH-4 512^2 step (BASELINE configs[1]'s batch) of the oracle's op sequence, twice on the same values: in f32, and
    under torch.autocast(bf16) — the reference's own arithmetic (ref train_lora_dreambooth.py:489-494) — both as plain torch ops
    on the GPU (``H.oracle_on_device``); loss and every LoRA gradient of each, computed once per module.  (Rounds 2-5 held the
    device step to cosine >= 0.99 against F32 per tensor, which a bf16 step of batch 4 does not meet on every tensor whoever
    computes it — 0.973-0.976 on one `up` tensor in all three configurations, the ATen-normalised one included — so that
    fixture had gone back to batch 1; round 6 judges batch 4 by the bracket rule instead, tests/helpers.bracket.)"""
    H.lap("fixture start")
    ref, ref_params, dev_unet = _sd15_twins()
    H.lap("fixture: twins built")
    g = torch.Generator().manual_seed(123)
    B = 4
    lat = torch.randn(B, 4, 64, 64, generator=g) * 0.18215
    ehs = torch.randn(B, 77, 768, generator=g)
    noise = torch.randn(B, 4, 64, 64, generator=g)
    ts = torch.randint(0, 1000, (B,), generator=g)
    # the device step sees bf16 inputs: give the oracle the same (bf16-representable) values
    lat, ehs, noise = (v.to(torch.bfloat16).float() for v in (lat, ehs, noise))
    with H.oracle_on_device():
        _, l32, g32 = H.oracle_step_on_device(ref, ref_params, lat.to(DEV), noise.to(DEV), ts.to(DEV), ehs.to(DEV), False)
        _, lbf, gbf = H.oracle_step_on_device(ref, ref_params, lat.to(DEV), noise.to(DEV), ts.to(DEV), ehs.to(DEV), True)
    H.lap("fixture: oracle steps (f32, bf16 autocast)")
    del ref
    torch.cuda.empty_cache()
    return dict(loss=l32, loss_bf=lbf, g32=g32, gbf=gbf, dev_unet=dev_unet, lat=lat, ehs=ehs, noise=noise, ts=ts)


def _compare_step(ref, loss_dev, st, label):
    """The bracket rule (tests/helpers.bracket / assert_bracket): the device step may be as far from the f32 step as the
    reference's own bf16 step is — loss within 1.5 x (+ 0.05 % of the loss: the reference's own loss error moves between 2e-5
    and 2e-4 with the library's attention picks), LoRA gradients in aggregate and for the median tensor within the measured
    ratio + 10 %, no tensor both 4 x the reference's error and 3 % off, none 5 % off; total gradient norm within 2 %."""
    l32, lbf = ref["loss"], ref["loss_bf"]
    assert abs(loss_dev - l32) <= 1.5 * abs(lbf - l32) + 5e-4 * abs(l32), (loss_dev, lbf, l32)
    H.assert_bracket(H.bracket(ref["g32"], ref["gbf"], st.flat_g, label))
    tot_r = float(torch.cat(ref["g32"]).norm())
    assert abs(float(st.flat_g.norm()) - tot_r) <= 0.02 * tot_r


@pytest.mark.parametrize("config", ["bench", "bench_fused_sites", "plain"])
def test_sd15_size_step_in_bench_configuration_matches_oracle(sd15_reference_step, monkeypatch, config):
    """VERDICT r2 item 1a.  "bench": what BENCH_rNN times (bf16, channels_last activations and conv weights, head-padded
    q/k/v/out projections, grouped q/k/v, the hostops passes, the step replayed from a hipGraph); "plain": NCHW, no head
    padding, one launch per projection, ATen normalisations, eager; "bench" additionally runs the adapters on the step's
    merged weight (``--merged 1``, bench.py's default since round 3), "bench_fused_sites" on the per-site fused MFMA
    kernels (``--merged 0``).  All at BATCH 4 against oracle/torch_ref.dreambooth_step by the bracket rule (``_compare_step``:
    as close to the f32 step as the reference's own bf16-autocast step is; the north star's "stated fp16 tolerance" made
    concrete as the reference's own mixed-precision error)."""
    from lora_amd.standin import fused

    ref = sd15_reference_step
    unet = ref["dev_unet"]
    bench_like = config.startswith("bench")
    monkeypatch.setenv("LORA_AMD_HEAD_PAD", "1" if bench_like else "0")
    monkeypatch.setenv("LORA_AMD_GROUP_QKV", "1" if bench_like else "0")
    monkeypatch.setattr(fused, "_ENABLED", bench_like)
    fmt = torch.channels_last if bench_like else torch.contiguous_format
    unet.to(memory_format=fmt)
    st = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-4, "weight_decay": 1e-2}], max_grad_norm=1.0,
                         device=torch.device(DEV))
    st.attach_direct_grads(unet)
    merged = st.enable_merged_weights(unet) if config == "bench" else None
    sched = DDPMScheduler()
    lat = ref["lat"].to(DEV).to(torch.bfloat16).contiguous(memory_format=fmt)
    ehs = ref["ehs"].to(DEV).to(torch.bfloat16)
    noise = ref["noise"].to(DEV).to(torch.bfloat16).contiguous(memory_format=fmt)
    ts = ref["ts"].to(DEV)

    def fwd_bwd(l_, c_):
        return T.forward_backward(unet, sched, l_, c_, T.StepConfig(), noise=noise, timesteps=ts, merged=merged)

    try:
        H.lap(f"{config}: state ready")
        for i in range(2):  # attention choices are timed on first use; the padded layout applies from the second call
            fwd_bwd(lat, ehs)
            st.zero_grad()
            H.lap(f"{config}: warm-up step {i}")
        if bench_like:
            graphed = T.GraphedForwardBackward(fwd_bwd, lat, ehs, st)
            st.zero_grad()
            H.lap(f"{config}: captured")
            loss = float(graphed(lat, ehs))
        else:
            loss = float(fwd_bwd(lat, ehs))
            st.reduce_pending()
        H.lap(f"{config}: measured step")
        _compare_step(ref, loss, st, config)
        H.lap(f"{config}: compared")
        if merged is not None:  # every one of the 144 sites took the merged path, in the layouts the host model uses
            assert len({id(e["module"]) for e in merged.entries.values()}) == 144 and merged.refreshes >= 3
    finally:
        for m in unet.modules():
            m.__dict__.pop("_grad_sink", None)
            m.__dict__.pop("_merged", None)


# ----------------------------------------------------------------------------- multi-GPU: 2 RCCL ranks (skips on a 1-GPU box)
_NCCL_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.environ["LORA_AMD_REPO"])
import lora_amd as L
from lora_amd import trainer as T
from lora_amd.standin import DDPMScheduler, tiny_unet
rank, local, world = T.init_distributed("cuda")
dev = torch.device("cuda", local)
torch.manual_seed(0)
unet = tiny_unet().to(dev)
unet.requires_grad_(False)
torch.manual_seed(100 + rank)  # replicas start different: FlatLoraState must broadcast rank 0's factors
L.inject_trainable_lora(unet, r=4)
for m in unet.modules():
    if isinstance(m, L.LoraInjectedLinear):
        m.lora_up.weight.data.normal_(0, 0.05)
st = T.FlatLoraState([{"params": T.lora_params(unet), "lr": 1e-3, "weight_decay": 1e-2}], max_grad_norm=1.0, device=dev)
st.attach_direct_grads(unet)
p0 = st.flat_p.clone()
g = torch.Generator().manual_seed(7)
lat, ehs = torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 7, 32, generator=g)
noise, ts = torch.randn(4, 4, 16, 16, generator=g), torch.randint(0, 1000, (4,), generator=g)
sl = slice(2 * rank, 2 * rank + 2)  # this rank's shard of the global batch of 4
T.forward_backward(unet, DDPMScheduler(), lat[sl].to(dev), ehs[sl].to(dev), T.StepConfig(), noise=noise[sl].to(dev),
                   timesteps=ts[sl].to(dev))
st.reduce_pending()
local_g = st.flat_g.clone()
scale = st.all_reduce()
summed = st.flat_g.clone()
st.step(scale)
gathered = [torch.empty_like(local_g) for _ in range(world)]
dist.all_gather(gathered, local_g)
ps = [torch.empty_like(st.flat_p) for _ in range(world)]
dist.all_gather(ps, st.flat_p)
p0s = [torch.empty_like(p0) for _ in range(world)]
dist.all_gather(p0s, p0)
if rank == 0:
    print(json.dumps({"world": world, "scale": scale, "backend": dist.get_backend(),
                      "sum_err": float((summed - sum(gathered)).abs().max()),
                      "gnorm": float(summed.norm()), "replica_diff": float((ps[0] - ps[1]).abs().max()),
                      "init_diff": float((p0s[0] - p0s[1]).abs().max()), "moved": float((ps[0] - p0s[0]).abs().max())}))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs on this node (the driver's 8-GPU box runs it)")
def test_two_rccl_ranks_allreduce_the_flat_gradient_and_step(tmp_path):
    """ref train_lora_dreambooth.py:744-757, 877: one process per GPU, backend nccl (= RCCL over xGMI): rank-0 broadcast
    of the flat LoRA state, ONE SUM all-reduce of flat_g, identical updates on both replicas."""
    import json
    import socket

    script = tmp_path / "w.py"
    script.write_text(_NCCL_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {**os.environ, "LORA_AMD_REPO": H.REPO, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["world"] == 2 and rec["backend"] == "nccl" and rec["scale"] == 0.5
    assert rec["sum_err"] <= 1e-6 * max(rec["gnorm"], 1e-6) and rec["gnorm"] > 0
    assert rec["init_diff"] == 0.0 and rec["replica_diff"] == 0.0 and rec["moved"] > 0
