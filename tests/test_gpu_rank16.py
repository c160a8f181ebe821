"""Matrix-core forms of the rank-9..16 kernels (csrc/rank16_mfma.hip) through the C ABI: against the float64 restatement of
lora.py:53-58 / its autograd (same tolerances as tests/test_gpu_kernels.py), against the VALU kernels they replace (the
`lora_amd_rank16_mfma` hook turned off), with the forward's dropout mask rebuilt by tests/helpers.philox_dropout_mask."""
import numpy as np
import pytest
import torch

from lora_amd import _C
from tests import helpers as H
from tests.test_gpu_kernels import DEV, DT, close, n, rnd

pytestmark = pytest.mark.gpu


class valu:
    """The VALU kernels for the block (what ran before round 4)."""

    def __enter__(self):
        self.prev = _C.rank16_mfma(0)

    def __exit__(self, *a):
        _C.rank16_mfma(self.prev)


def mask_of(M, N, p, seed, off):
    if p == 0.0:
        return np.ones((M, N), np.float64)
    return H.philox_dropout_mask(M * N, p, seed, off).view(M, N).double().numpy()


@pytest.mark.parametrize("M,K,r,layout,p,dt", [
    (9216, 320, 16, "KR", 0.1, "bf16"), (577, 1280, 16, "RK", 0.0, "bf16"), (144, 2560, 12, "KR", 0.1, "bf16"),
    (33, 640, 9, "RK", 0.0, "f16"), (1, 1280, 16, "KR", 0.1, "bf16"), (77, 768, 16, "RK", 0.25, "f16"), (2304, 960, 16, "RK", 0.0, "bf16")])
def test_rowdot16_vs_float64_and_vs_valu(M, K, r, layout, p, dt):
    """T = s (mask . X) F^T: lora.py:56's lora_down (RK, no mask) and the autograd's dT = s (mask . G) up (KR, masked)."""
    x = rnd((M, K), dt, seed=1)
    f = rnd((r, K) if layout == "RK" else (K, r), "f32", 0.3, seed=2)
    lay = _C.FACTOR_RK if layout == "RK" else _C.FACTOR_KR
    seed, off, s = 0x51ED5EED, 12345, 0.7
    assert _C.rank16_mfma(-1) == 1
    got = _C.rowdot(x, f, lay, s, None, False, p, seed, off)
    with valu():
        old = _C.rowdot(x, f, lay, s, None, False, p, seed, off)
    F = n(f).astype(np.float64) if layout == "RK" else n(f).astype(np.float64).T
    xm = n(x).astype(np.float64) * mask_of(M, K, p, seed, off)
    want = s * (xm @ F.T)
    bound = s * (np.abs(xm) @ np.abs(F.T))
    close(n(got), want, bound, msg="rowdot16")
    close(n(old), want, bound, msg="rowdot valu")
    assert np.abs(n(got) - n(old)).max() <= 4e-5 * bound.max()


@pytest.mark.parametrize("M,N,r,layout,p,dt", [
    (9216, 320, 16, "KR", 0.1, "bf16"), (2304, 640, 16, "KR", 0.0, "bf16"), (576, 1280, 12, "RK", 0.1, "bf16"),
    (144, 10240, 16, "KR", 0.1, "bf16"), (33, 648, 9, "RK", 0.0, "f16"), (1, 320, 16, "KR", 0.2, "bf16"), (100, 2560, 16, "RK", 0.1, "f16")])
def test_rank_update16_vs_float64_and_vs_valu(M, N, r, layout, p, dt):
    """Y += s mask . (T F): lora.py:56-57 (lora_up, dropout, scale, +) and the autograd's dX += Gt down."""
    y0, t = rnd((M, N), dt, seed=1), rnd((M, r), "f32", 0.5, seed=2)
    f = rnd((r, N) if layout == "RK" else (N, r), "f32", 0.3, seed=3)
    lay = _C.FACTOR_RK if layout == "RK" else _C.FACTOR_KR
    seed, off, s = 77, (1 << 33) + 5, 1.3
    y = y0.clone()
    _C.rank_update_(y, t, f, lay, s, p, seed, off)
    yv = y0.clone()
    with valu():
        _C.rank_update_(yv, t, f, lay, s, p, seed, off)
    F = n(f).astype(np.float64) if layout == "RK" else n(f).astype(np.float64).T
    prod = n(t).astype(np.float64) @ F
    want = n(y0).astype(np.float64) + s * mask_of(M, N, p, seed, off) * prod
    bound = np.abs(n(y0)) + s / (1 - p) * (np.abs(n(t)) @ np.abs(F))
    close(n(y), want, bound, dt, msg="rank_update16")
    close(n(yv), want, bound, dt, msg="rank_update valu")
    # the two kernels round the same exact value: they differ by at most one unit in the last place, on few elements
    diff = (y.float() - yv.float()).abs()
    assert float((diff > 0).float().mean()) < 0.02
    assert float(diff.max()) <= 2.0 ** (-7 if dt == "bf16" else -10) * float(np.abs(want).max())


@pytest.mark.parametrize("M,N,r,p,dt", [
    (9216, 320, 16, 0.1, "bf16"), (2304, 640, 16, 0.1, "bf16"), (576, 1280, 16, 0.0, "bf16"), (144, 1280, 12, 0.1, "bf16"),
    (9216, 2560, 16, 0.1, "bf16"), (100, 5120, 9, 0.1, "f16"), (61, 64, 16, 0.0, "bf16"), (1, 1280, 16, 0.1, "bf16"),
    (300, 96, 16, 0.2, "f16"), (1000, 10240, 16, 0.1, "bf16")])
def test_bwd_g16_partials_vs_float64_and_vs_valu(M, N, r, p, dt):
    """One pass over G (lora.py:53-58's autograd): gt_part[ct] sums to Gt = s (mask . G) up, up_part[rb] to
    dUp = s (mask . G)^T T — the same partial buffers and launch geometry as the VALU kernel, which must agree."""
    K = 320
    plan = _C.linear_plan(M, K, N, r)
    assert plan.fused == 1 and plan.rank_tile == 16
    g, t, up = rnd((M, N), dt, seed=1), rnd((M, r), "f32", 0.5, seed=2), rnd((N, r), "f32", 0.3, seed=3)
    seed, off, s = 991, 3, 0.9
    res = {}
    for which in ("mfma", "valu"):
        gt_part = torch.full((plan.gt_part_floats,), 7.0, device=DEV)
        up_part = torch.full((plan.up_part_floats,), 7.0, device=DEV)
        if which == "valu":
            with valu():
                _C.linear_bwd_g(g, t, up, gt_part, up_part, s, p, seed, off)
        else:
            _C.linear_bwd_g(g, t, up, gt_part, up_part, s, p, seed, off)
        res[which] = (n(gt_part).reshape(plan.nct_g, M, r).sum(0),
                      n(up_part).reshape(plan.nparts_up, 16, N).sum(0)[:r].T.copy(),
                      n(up_part).reshape(plan.nparts_up, 16, N).sum(0)[r:])
    gm = n(g).astype(np.float64) * mask_of(M, N, p, seed, off)
    Un, Tn = n(up).astype(np.float64), n(t).astype(np.float64)
    Gt, dUp = s * (gm @ Un), s * (gm.T @ Tn)
    bG, bU = s * (np.abs(gm) @ np.abs(Un)), s * (np.abs(gm.T) @ np.abs(Tn))
    for which in ("mfma", "valu"):
        close(res[which][0], Gt, bG, msg=f"Gt {which}")
        close(res[which][1], dUp, bU, msg=f"dUp {which}")
        assert not res[which][2].any()   # the rank tile's rows beyond r stay zero


@pytest.mark.parametrize("M,K,N,r,p,dt", [(2304, 960, 640, 16, 0.1, "bf16"), (576, 1280, 10240, 16, 0.1, "bf16"),
                                           (144, 2560, 1280, 12, 0.0, "f16"), (9216, 960, 320, 16, 0.1, "bf16")])
def test_linear_fwd16_vs_float64_and_vs_valu(M, K, N, r, p, dt):
    """lora.py:53-58 on top of a given frozen product: T = X down^T (saved for the backward), Y += s dropout(T up^T)."""
    x, y0 = rnd((M, K), dt, seed=1), rnd((M, N), dt, seed=2)
    A, B = rnd((r, K), "f32", 0.1, seed=3), rnd((N, r), "f32", 0.2, seed=4)
    seed, off, s = 5, 1 << 20, 0.6
    y = y0.clone()
    t = _C.linear_fwd_(x, y, A, B, s, None, p, seed, off)
    yv = y0.clone()
    with valu():
        tv = _C.linear_fwd_(x, yv, A, B, s, None, p, seed, off)
    Xn, An, Bn = n(x).astype(np.float64), n(A).astype(np.float64), n(B).astype(np.float64)
    T = Xn @ An.T
    close(n(t), T, np.abs(Xn) @ np.abs(An.T), msg="T")
    close(n(tv), T, np.abs(Xn) @ np.abs(An.T), msg="T valu")
    want = n(y0).astype(np.float64) + s * mask_of(M, N, p, seed, off) * (T @ Bn.T)
    bound = np.abs(n(y0)) + s / (1 - p) * ((np.abs(Xn) @ np.abs(An.T)) @ np.abs(Bn.T))
    close(n(y), want, bound, dt, msg="y")
    close(n(yv), want, bound, dt, msg="y valu")


@pytest.mark.parametrize("B,M,C,r", [(3, 320, 320, 16), (1, 1280, 2880, 16), (2, 77, 10240, 16), (2, 2560, 320, 12), (1, 17, 64, 16)])
def test_rowdot16_planes_vs_float64(B, M, C, r):
    """cli_svd's skinny products on the matrix cores: out[b] = X[b] F[b] for f32 stacks held as (hi, lo) bf16 planes
    (lora_amd_split16_ragged) — the planes reproduce X to 2^-17 relative, the product the float64 one within the f32-grade
    tolerance of the other kernels of this file."""
    g = torch.Generator().manual_seed(B * 1000 + M + C)
    x = (torch.randn(B, M, C, generator=g) * 0.1).to(DEV)
    f = torch.randn(B, C, r, generator=g).to(DEV)
    hi, lo = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    _C.split16_ragged([x.view(-1)], [hi.view(-1)], [lo.view(-1)])
    back = hi.float() + lo.float()
    assert float((back - x).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    assert torch.equal(hi, x.to(torch.bfloat16))
    # the same planes, and those of the transposed matrices, from the one-read launch of the SVD path
    if M % 8 == 0:
        h2, l2 = torch.empty_like(hi), torch.empty_like(lo)
        th, tl = (torch.empty(B, C, M, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        _C.split16_transpose([x], [h2], [l2], [th], [tl])
        assert torch.equal(h2, hi) and torch.equal(l2, lo)
        assert torch.equal(th, hi.transpose(1, 2)) and torch.equal(tl, lo.transpose(1, 2))
    out = torch.full((B, M, r), 7.0, device=DEV)
    prog = _C.PlanesProgram(torch.device(DEV), r)
    h = prog.table([(hi, lo, f, out)])
    prog.upload()
    prog.run(h)
    xn, fn = n(x).astype(np.float64), n(f).astype(np.float64)
    want = np.einsum("bmc,bcr->bmr", xn, fn)
    bound = np.einsum("bmc,bcr->bmr", np.abs(xn), np.abs(fn))
    close(n(out), want, bound, k=3e-5, msg="rowdot16 over planes")


def test_svd_iteration_on_planes_matches_the_f32_passes():
    """topr_svd_ragged with the matrix-core products over (hi, lo) planes (cli_svd.PLANES) against the same iteration on the
    f32 column-reduction passes, same random sketch: the rank-r products agree far inside the test tolerance of either
    against the exact SVD (tests/test_gpu_parity_r3.py::test_ragged_svd_of_several_shape_groups_vs_exact_svd runs the
    default, i.e. the planes)."""
    from lora_amd import cli_svd as S
    from tests.test_cli_svd import _planted

    r = 8
    shapes = [(2, 320, 320), (1, 640, 2880), (2, 1280, 320)]
    deltas = []
    for gi, (B, N, K) in enumerate(shapes):
        tb = [_planted(N, K, r + 4, 2e-3 / (N ** 0.5 + K ** 0.5), 7 * gi + i, "cpu") for i in range(B)]
        deltas.append(torch.stack([(t - b).to(DEV) for t, b in tb]))
    res = {}
    for planes in (True, False):
        prev, S.PLANES = S.PLANES, planes
        try:
            res[planes] = S.topr_svd_ragged([d.clone() for d in deltas], r, generator=torch.Generator(device=DEV).manual_seed(0))
        finally:
            S.PLANES = prev
    for (U1, S1, V1), (U0, S0, V0), d in zip(res[True], res[False], deltas):
        p1, p0 = (U1 * S1[:, None, :]) @ V1, (U0 * S0[:, None, :]) @ V0
        assert float((p1 - p0).norm()) <= 5e-5 * float(p0.norm())
        assert float((S1 - S0).abs().max()) <= 2e-5 * float(S0.max())


def test_clamp_quantile_from_the_top_order_statistics_equals_torch_quantile():
    """cli_svd._quantile_rows (ref cli_svd.py:39-47's torch.quantile of the joint factor values) on the device: bit for bit."""
    from lora_amd import cli_svd as S

    g = torch.Generator().manual_seed(4)
    for B, nn_ in ((3, 5120), (2, 102400), (1, 99999), (4, 640), (2, 63)):
        x = torch.randn(B, nn_, generator=g).to(DEV)
        for q in (0.99, 0.9, 1.0, 0.5):
            assert torch.equal(S._quantile_rows(x, q), torch.quantile(x, q, dim=1)), (B, nn_, q)
