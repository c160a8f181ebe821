"""K1/K2 input-stationary kernel (scripts/gemm_xs/gemm_xs.hip, round 5; an experiment outside the product library since round 6 —
run with ``python -m pytest scripts/gemm_xs/check_gemm_xs.py -m gpu``): lora.py:53-58 and its input gradient at the short-contraction /
many-row sites, against the numpy oracle — plain product, LoRA branch, dropout with the kernels' own mask extracted, the
accumulate form, ragged M / N tails, f16.  The factors enter as hi + lo fragments (f32-grade), so the tolerance on the branch
is the output rounding alone.  Goes through ``scripts/gemm_xs/xs.py`` (ctypes)."""
import numpy as np
import pytest
import torch

from lora_amd import _C
from scripts.gemm_xs import xs as XS
from oracle import lora_numpy as O
from tests.test_gpu_kernels import close, n, rnd
from tests.test_gpu_parity_r3 import _mask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

XS_SHAPES = [(16384, 320, 320, 4), (4096, 640, 640, 4), (9216, 320, 2560, 16), (2304, 640, 5120, 8), (1000, 320, 324, 5),
             (77, 640, 20, 3), (130, 320, 1284, 1), (33, 320, 160, 16)]


@pytest.mark.parametrize("M,K,N,r", XS_SHAPES)
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_xs_forward_matches_oracle(M, K, N, r, dt):
    """Y = X W^T + b + s (X down^T) up^T and T through lora_amd_linear_xs vs oracle.lora_linear_forward (f64 products of the
    same 16-bit inputs): asymmetric operands, ragged M / N tails, ranks that are not multiples of 4."""
    s = 0.7
    x, w = rnd((M, K), dt, 1.0, seed=1), rnd((N, K), dt, 0.05, seed=2)
    b = rnd((N,), dt, 0.5, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    assert XS.xs_supported(x, K, N, r)
    y, t = XS.linear_xs_fwd(x, w, b, down, up, s)
    X, W, Bv, A, U = n(x), n(w), n(b), n(down), n(up)
    yo, to = O.lora_linear_forward(X, W, Bv, A, U, s)
    close(n(t), to, np.abs(X) @ np.abs(A).T, "f32", k=3e-5, msg="T")
    absref = np.abs(X) @ np.abs(W).T + np.abs(Bv) + np.abs(to) @ np.abs(s * U).T
    close(n(y), yo, absref, dt, k=3e-5, msg="Y")   # f32-grade accumulation + ONE rounding of the output


@pytest.mark.parametrize("M,K,N", [(16384, 320, 960), (4096, 640, 640), (300, 320, 100), (2304, 640, 5120)])
def test_xs_plain_product_on_a_packed_weight(M, K, N):
    """site.down == NULL: Y = X W^T + b (what a merged-weight site runs) from the packed operand AND from the row-major
    weight itself (site.reserved = 1); the accumulate flag of lora_amd_linear_ws is refused."""
    dt = "bf16"
    x, w, b = rnd((M, K), dt, 1.0, seed=1), rnd((N, K), dt, 0.05, seed=2), rnd((N,), dt, 0.5, seed=3)
    y, t = XS.linear_xs(x, dict(wp=_C.ws_pack(w), N=N, bias=b))
    assert t is None
    X, W, Bv = n(x), n(w), n(b)
    ref = X @ W.T + Bv
    absref = np.abs(X) @ np.abs(W).T + np.abs(Bv)
    close(n(y), ref, absref, dt, k=3e-5, msg="Y")
    y_rm, _ = XS.linear_xs(x, dict(wp=w, N=N, bias=b, rowmajor=True))
    assert torch.equal(y_rm, y)
    with pytest.raises(RuntimeError, match="no accumulate form"):
        XS.linear_xs(x, dict(wp=_C.ws_pack(w), N=N, y=y.clone(), flayout=4))


@pytest.mark.parametrize("M,K,N,r", [(4096, 640, 640, 4), (9216, 2560, 320, 16), (1000, 1280, 640, 8), (130, 768, 320, 3),
                                     (16384, 320, 320, 4)])
def test_xs_input_gradient(M, K, N, r):
    """dX = G W + s (G up) down and Gt = s G up through the same kernel on W packed in the transposed orientation (contraction
    over N in {320, 640}), factors read in place in their [N, r] / [r, K] layouts."""
    dt, s = "bf16", 0.6
    g, w = rnd((M, N), dt, 1.0, seed=2), rnd((N, K), dt, 0.05, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    assert XS.xs_supported(g, N, K, r)
    dx, gt = XS.linear_xs_dx(g, w, down, up, s)
    G, W, A, U = n(g), n(w), n(down), n(up)
    dxo, _, _, _, _ = O.lora_linear_backward(G, np.zeros((M, K), np.float32), W, A, U, s)
    gt_ref = s * (G @ U)
    close(n(gt), gt_ref, s * (np.abs(G) @ np.abs(U)), "f32", k=3e-5, msg="Gt")
    close(n(dx), dxo, np.abs(G) @ np.abs(W) + np.abs(gt_ref) @ np.abs(A), dt, k=3e-5, msg="dX")


@pytest.mark.parametrize("M,K,N,r,p", [(4096, 320, 320, 16, 0.1), (2048, 640, 640, 4, 0.25), (9216, 320, 2560, 16, 0.1),
                                       (1000, 320, 2560, 8, 0.1)])
def test_xs_dropout_forward_vs_oracle_with_extracted_mask(M, K, N, r, p):
    """nn.Dropout(p) on the branch (lora.py:45, 56) inside the launch, the mask regenerated from (seed, offset) exactly as the
    other kernels index it (extracted here with a rank-1 update of zeros), vs oracle.lora_linear_forward(mask=)."""
    dt, s, seed = "bf16", 0.9, 4321
    x, w, b = rnd((M, K), dt, seed=1), rnd((N, K), dt, 0.05, seed=2), rnd((N,), dt, seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.5, seed=5)
    off = torch.tensor([977], dtype=torch.int64, device=DEV)
    mask = _mask(M, N, p, seed, off)
    y, t = XS.linear_xs_fwd(x, w, b, down, up, s, p, seed, off)
    X, W, Bv, A, U = n(x), n(w), n(b), n(down), n(up)
    yo, to = O.lora_linear_forward(X, W, Bv, A, U, s, None, mask)
    close(n(t), to, np.abs(X) @ np.abs(A).T, "f32", k=3e-5, msg="T")
    absref = np.abs(X) @ np.abs(W).T + np.abs(Bv) + (np.abs(to) @ np.abs(s * U).T) * mask
    close(n(y), yo, absref, dt, k=3e-5, msg="Y")
    frozen = X @ W.T + Bv            # dropped elements carry the frozen product only
    dropped = mask == 0
    assert np.abs(n(y) - frozen)[dropped].max() <= 2.0 ** -8 * np.abs(frozen)[dropped].max() + 1e-4 * absref[dropped].max()


@pytest.mark.parametrize("M,K,N,r,p", [(4096, 320, 320, 16, 0.1), (2048, 1280, 640, 4, 0.25), (9216, 2560, 320, 16, 0.1)])
def test_xs_dropout_input_gradient_vs_oracle_with_extracted_mask(M, K, N, r, p):
    """The input gradient with the forward's mask on G (only on the branch): Gt = s (mask o G) up, dX = G W + Gt down."""
    dt, s, seed, off = "bf16", 0.8, 99, 31337
    g, w = rnd((M, N), dt, seed=1), rnd((N, K), dt, 0.05, seed=2)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    mask = _mask(M, N, p, seed, off)
    G, W, A, U = n(g), n(w), n(down), n(up)
    dxo, _, _, _, _ = O.lora_linear_backward(G, np.zeros((M, K), np.float32), W, A, U, s, None, mask)
    gt_o = s * ((G * mask) @ U)
    dx, gt = XS.linear_xs_dx(g, w, down, up, s, p, seed, off)
    close(n(gt), gt_o, s * ((np.abs(G) * mask) @ np.abs(U)), "f32", k=3e-5, msg="Gt")
    close(n(dx), dxo, np.abs(G) @ np.abs(W) + np.abs(gt_o) @ np.abs(A), dt, k=3e-5, msg="dX")


@pytest.mark.parametrize("sl,pg", [(1, 1), (2, 3), (4, 2), (4, 64), (2, 1)])
def test_xs_block_geometries_agree(sl, pg):
    """Slabs per wave / panels per workgroup are launch geometry only: every choice gives the same bits (one accumulation
    order per output element), through the panel loop's double-buffered LDS images and counted waits."""
    M, K, N, r, s = 5000, 320, 2560, 8, 0.7
    x, w, b = rnd((M, K), "bf16", seed=1), rnd((N, K), "bf16", 0.05, seed=2), rnd((N,), "bf16", seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    y0, t0 = XS.linear_xs_fwd(x, w, b, down, up, s, 0.1, 5, 9)
    try:
        XS.xs_set_tuning(sl, pg)
        y1, t1 = XS.linear_xs_fwd(x, w, b, down, up, s, 0.1, 5, 9)
    finally:
        XS.xs_set_tuning(0, 0)
    assert torch.equal(y0, y1) and torch.equal(t0, t1)


def test_xs_and_ws_kernels_agree_on_the_same_site():
    """The two routes of one site (weight-stationary / input-stationary) on the same packed weight: the same result to the
    rounding of T and s up that only the weight-stationary kernel performs."""
    M, K, N, r, s = 4096, 320, 640, 4, 0.9
    x, w, b = rnd((M, K), "bf16", seed=1), rnd((N, K), "bf16", 0.05, seed=2), rnd((N,), "bf16", seed=3)
    down, up = rnd((r, K), "f32", 0.2, seed=4), rnd((N, r), "f32", 0.3, seed=5)
    y1, t1 = XS.linear_xs_fwd(x, w, b, down, up, s)
    y2, t2 = _C.linear_ws_fwd(x, w, b, down, up, s)
    assert (t1 - t2).abs().max() <= 1e-4 * t2.abs().max()
    assert (y1.float() - y2.float()).abs().max() <= 2.0 ** -6 * y2.float().abs().max()
