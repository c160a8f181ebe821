"""Round-6 device parity: the dropout sites around the attention core (lora.py:53-58 called by CrossAttention's to_q / to_k /
to_v / to_out under ``dropout_p > 0``, the reference's default 0.1) in HEAD-PADDED rows on the weight-stationary kernel
(``lora_amd_linear_ws_heads``, ``lora_amd_ws_site.y_heads``): rounds 3-5 unpacked the input and packed the output with copies.

* kernel level, through the C-ABI: forward and input-gradient launches with a head-padded input / output against
  ``oracle/lora_numpy`` with the restated Philox mask, AND bit-equal to the dense launch around explicit copies; the input's
  pad (filled with NaN) is never read, the output's pad (pre-filled with NaN) leaves as zeros;
* the factor pass with dropout AND head-padded rows against the oracle (the two were only tested apart);
* module level: ``LoraInjectedLinear.forward_heads`` on a trainer's sink with the deferred factor pass, ``ops.WS_HEADS`` on
  against off: outputs, input gradients and the folded factor gradients are the same bits."""
import numpy as np
import pytest
import torch

import lora_amd as L
from lora_amd import _C, ops
from lora_amd import trainer as T
from oracle import lora_numpy as O
from tests import helpers as H
from tests.test_gpu_kernels import close, n, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pack(a: torch.Tensor, lay, fill: float) -> torch.Tensor:
    """[M, h*d] -> [M, h*D] with the pad columns set to ``fill`` (a kernel must not read them / must overwrite them)."""
    h, d, D = lay
    out = torch.full((a.shape[0], h, D), fill, dtype=a.dtype, device=a.device)
    out[:, :, :d] = a.view(a.shape[0], h, d)
    return out.view(a.shape[0], h * D)


def _unpack(a: torch.Tensor, lay) -> torch.Tensor:
    h, d, D = lay
    return a.view(a.shape[0], h, D)[:, :, :d].reshape(a.shape[0], h * d)


def _pads(a: torch.Tensor, lay) -> torch.Tensor:
    h, d, D = lay
    return a.view(a.shape[0], h, D)[:, :, d:]


CASES = [(1000, 320, 320, 16, (8, 40, 64), torch.bfloat16), (2304, 640, 640, 16, (8, 80, 128), torch.bfloat16),
         (576, 1280, 1280, 16, (8, 160, 256), torch.bfloat16), (1000, 320, 320, 4, (8, 40, 64), torch.bfloat16),
         (520, 640, 640, 6, (8, 80, 128), torch.float16), (333, 320, 320, 8, (8, 40, 64), torch.float16)]


@pytest.mark.parametrize("M,K,N,r,lay,dt", CASES)
@pytest.mark.parametrize("side", ["x", "y"])
def test_ws_forward_in_head_padded_rows(M, K, N, r, lay, dt, side):
    """Y = X W^T + b + s * mask .* ((X down^T) up^T) (lora.py:53-58 with nn.Dropout, lora.py:45) with X (side x) or Y (side y)
    head-padded: oracle within the fused kernels' tolerance, and the same bits as the dense launch between copies."""
    name = "bf16" if dt == torch.bfloat16 else "f16"
    p, seed, off, s_ = 0.1, 1234, 96, 0.7
    x, w, b = rnd((M, K), name, seed=1), rnd((N, K), name, 0.05, seed=2), rnd((N,), name, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    y_ref, t_ref = _C.linear_ws_fwd(x, w, b, down, up, s_, 0, p, seed, off)
    if side == "x":
        xp = _pack(x, lay, float("nan"))
        y, t = _C.linear_ws_fwd(xp, w, b, down, up, s_, 0, p, seed, off, x_heads=lay)
        assert torch.equal(y, y_ref) and torch.equal(t, t_ref)
    else:
        assert _C.ws_heads_ok(K, N, None, lay)
        ybuf = torch.full((M, lay[0] * lay[2]), float("nan"), dtype=dt, device=DEV)
        (y, t), = _C.linear_ws(x, [dict(wp=_C.ws_pack(w), N=N, bias=b, down=down, up=up, scale=s_, p=p, seed=seed, off=off,
                                         y=ybuf, y_heads=lay)])
        assert y.data_ptr() == ybuf.data_ptr()
        assert torch.equal(_unpack(y, lay), y_ref) and torch.equal(t, t_ref)
        assert float(_pads(y, lay).float().abs().max()) == 0.0 and not bool(torch.isnan(y).any())
    # ... and the dense launch is the reference's op sequence with the restated mask
    X, W, A, U = n(x), n(w), n(down), n(up)
    mask = H.philox_dropout_mask(M * N, p, seed, off).view(M, N).numpy()
    yo, _ = O.lora_linear_forward(X, W, n(b), A, U, s_, mask=mask)
    absy = np.abs(X) @ np.abs(W).T + np.abs(n(b)) + s_ / (1 - p) * (np.abs(X) @ np.abs(A).T) @ np.abs(U).T
    close(n(y_ref), yo, absy, name, k=2.0, msg="Y")


@pytest.mark.parametrize("M,K,N,r,lay,dt", CASES)
@pytest.mark.parametrize("side", ["g", "dx"])
def test_ws_input_gradient_in_head_padded_rows(M, K, N, r, lay, dt, side):
    """dX = G W + s ((mask .* G) up) down, Gt = s (mask .* G) up (the autograd of lora.py:53-58 for the input) with G (side g: the
    site's output was head-padded) or dX (side dx: its input was) in head-padded rows: the same bits as the dense launch."""
    name = "bf16" if dt == torch.bfloat16 else "f16"
    p, seed, off, s_ = 0.1, 77, 8, 0.9
    g, w = rnd((M, N), name, seed=11), rnd((N, K), name, 0.05, seed=12)
    down, up = rnd((r, K), "f32", 0.2, seed=14), rnd((N, r), "f32", 0.3, seed=15)
    dx_ref, gt_ref = _C.linear_ws_dx(g, w, down, up, s_, 0, p, seed, off)
    if side == "g":
        gp = _pack(g, lay, float("nan"))
        dx, gt = _C.linear_ws_dx(gp, w, down, up, s_, 0, p, seed, off, g_heads=lay)
        assert torch.equal(dx, dx_ref) and torch.equal(gt, gt_ref)
    else:
        assert _C.ws_heads_ok(N, K, None, lay)
        dx, gt = _C.linear_ws_dx(g, w, down, up, s_, 0, p, seed, off, dx_heads=lay)
        assert dx.shape == (M, lay[0] * lay[2])
        assert torch.equal(_unpack(dx, lay), dx_ref) and torch.equal(gt, gt_ref)
        assert float(_pads(dx, lay).float().abs().max()) == 0.0
    G, W, A, U = n(g), n(w), n(down), n(up)
    mask = H.philox_dropout_mask(M * N, p, seed, off).view(M, N).numpy()
    gto = s_ * (G * mask) @ U
    close(n(gt_ref), gto, s_ / (1 - p) * np.abs(G) @ np.abs(U), "f32", k=1e-4, msg="Gt")
    dxo = G @ W + gto @ A
    close(n(dx_ref), dxo, np.abs(G) @ np.abs(W) + np.abs(gto) @ np.abs(A), name, k=2.0, msg="dX")


@pytest.mark.parametrize("M,K,N,r,gh,xh", [(2304, 320, 320, 16, (8, 40, 64), None), (2304, 320, 320, 16, None, (8, 40, 64)),
                                           (1000, 640, 640, 8, (8, 80, 128), None), (576, 1280, 1280, 4, None, (8, 160, 256))])
def test_factor_pass_with_dropout_on_head_padded_rows(M, K, N, r, gh, xh):
    """dUp = s (mask .* G)^T (X down^T), dDown = (s (mask .* G) up)^T X through the one-launch matrix-core pass with the mask
    regenerated on a HEAD-PADDED G (or with a head-padded X beside a masked dense G) vs oracle.lora_linear_backward."""
    dt, p, seed, off, s_ = torch.bfloat16, 0.1, 4321, 40, 0.7
    x, g = rnd((M, K), "bf16", seed=21), rnd((M, N), "bf16", seed=22)
    down, up = rnd((r, K), "f32", 0.2, seed=23), rnd((N, r), "f32", 0.3, seed=24)
    gd = _pack(g, gh, 7.0) if gh else g
    xd = _pack(x, xh, -3.0) if xh else x
    plan = _C.factors_mfma_plan(M, K, N, r, dt)
    assert plan.supported
    up_part = torch.full((int(plan.up_part_floats),), float("nan"), device=DEV)
    down_part = torch.full((int(plan.down_part_floats),), float("nan"), device=DEV)
    pk_down = torch.empty(int(plan.pack_down_elems), dtype=dt, device=DEV)
    pk_up = torch.empty(int(plan.pack_up_elems), dtype=dt, device=DEV)
    arr, total = _C.factor_pack_table([(down, up, pk_down, pk_up)])
    _C.factor_pack(_C.table_to_device(arr, DEV), 1, total, dt)
    row = (gd, xd, pk_down, pk_up, up_part, down_part, s_, gh, xh, r, plan, (p, seed, off))
    arr, grid = _C.factors_mfma_table([row], dt, int(plan.lds_class))
    _C.linear_bwd_factors_mfma_ragged(_C.table_to_device(arr, DEV), 1, grid, int(plan.lds_class), dt, True,
                                      int(plan.rows_per_block))
    d_up, d_down = torch.empty(N, r, device=DEV), torch.empty(r, K, device=DEV)
    table, cnt, tot = _C.make_reduce_table(
        [(up_part, d_up, plan.nparts, plan.rank_tile, N, r, _C.FACTOR_KR, 1.0, 0.0),
         (down_part, d_down, plan.nparts, plan.rank_tile, K, r, _C.FACTOR_RK, 1.0, 0.0)], DEV)
    _C.reduce_batched(table, cnt, tot)
    G, X, A, U = n(g), n(x), n(down), n(up)
    mask = H.philox_dropout_mask(M * N, p, seed, off).view(M, N).numpy()
    _, ddo, duo, _, _ = O.lora_linear_backward(G, X, np.zeros((N, K), np.float32), A, U, s_, mask=mask)
    close(n(d_up), duo, s_ / (1 - p) * (np.abs(G).T @ (np.abs(X) @ np.abs(A).T)), "f32", k=1e-4, msg="dUp")
    close(n(d_down), ddo, (s_ / (1 - p) * np.abs(G) @ np.abs(U)).T @ np.abs(X), "f32", k=1e-4, msg="dDown")


@pytest.mark.parametrize("in_heads,out_heads,K,N", [(None, (8, 40, 64), 320, 320), ((8, 40, 64), None, 320, 320),
                                                    (None, (8, 80, 128), 640, 640), ((8, 160, 256), None, 1280, 1280)])
def test_dropout_site_forward_heads_with_and_without_the_copies(in_heads, out_heads, K, N, monkeypatch):
    """LoraInjectedLinear.forward_heads (dropout 0.1, rank 16) on a trainer's sink: ops.WS_HEADS (the kernels read / write the
    padded rows) against the unpack -> kernel -> pack copies of rounds 3-5 — same seed stream, same bits everywhere."""
    M, r, s = 2304, 16, 1.0

    def run(flag):
        monkeypatch.setattr(ops, "WS_HEADS", flag)
        torch.manual_seed(0)
        m = L.LoraInjectedLinear(K, N, in_heads is not None, r=r, dropout_p=0.1, scale=s).to(DEV).to(torch.bfloat16)
        m.linear.requires_grad_(False)
        T.promote_lora_to_fp32(m)
        m.lora_up.weight.data.normal_(0, 0.05)
        m.train()
        holder = torch.nn.ModuleList([m])
        st = T.FlatLoraState([{"params": T.lora_params(holder), "lr": 1e-3, "weight_decay": 0.0}], max_grad_norm=0.0, device=DEV)
        st.attach_direct_grads(holder)
        mw = st.enable_merged_weights(holder)
        x, gy = rnd((M, K), "bf16", seed=5), rnd((M, N), "bf16", seed=6)
        xd = (_pack(x, in_heads, 0.0) if in_heads else x).clone().requires_grad_(True)
        gd = _pack(gy, out_heads, 0.0) if out_heads else gy
        log = []
        monkeypatch.setattr(ops, "PATH_LOG", log)
        with ops.dropout_pool(DEV):
            torch.manual_seed(99)   # the dropout stream of both runs
            mw.refresh()
            y = m.forward_heads(xd, in_heads, out_heads)
            y.backward(gd)
            mw.flush_factors()
        st.all_reduce()
        g_flat = st.flat_g.clone()
        return y.detach().clone(), xd.grad.clone(), g_flat, [e[1] for e in log]

    y1, dx1, g1, paths1 = run(True)
    y0, dx0, g0, paths0 = run(False)
    assert any("ws_heads" in p_ for p_ in paths1), paths1
    assert not any("ws_heads" in p_ for p_ in paths0), paths0
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    assert float(g1.abs().max()) > 0 and torch.equal(g1, g0)
    if out_heads:
        assert float(_pads(y1, out_heads).float().abs().max()) == 0.0
    if in_heads:
        assert float(_pads(dx1, in_heads).float().abs().max()) == 0.0
