"""K5 round 5: the fused small steps of the batched subspace iteration (csrc/svd_small.hip) against numpy / torch f64, the
selection kernel against torch.quantile bit for bit, and the distillation against the reference recipe (cli_svd.py:30-47: exact
``torch.linalg.svd``, top r) on spectra that do NOT decay: sigma_i ~ i^-0.5 and a fine-tuning-like delta with a flat tail.
Everything goes through the C-ABI (``lora_amd/_C.py``)."""
import numpy as np
import pytest
import torch

from lora_amd import _C
from lora_amd import cli_svd as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _thin_sites(rows_list, seed=0, cond=1.0):
    """Flat buffer of thin matrices [rows][16] (sites 64-element aligned) + table."""
    g = torch.Generator().manual_seed(seed)
    offs, total = [], 0
    for rows in rows_list:
        offs.append(total)
        total += -(-(rows * 16) // 64) * 64
    flat = torch.zeros(total)
    mats = []
    for off, rows in zip(offs, rows_list):
        m = torch.randn(rows, 16, generator=g) * torch.logspace(0, -np.log10(cond), 16)[None, :]
        m = m @ torch.linalg.qr(torch.randn(16, 16, generator=g))[0]
        flat[off:off + rows * 16] = m.flatten()
        mats.append(m.double())
    tab = _C.ThinTable(list(zip(offs, rows_list)), DEV)
    return flat.to(DEV), tab, mats, offs


ROWS = [320, 16, 1, 255, 256, 257, 1280, 23040, 4096, 77]


def test_thin_gram_cholesky_inverse_vs_numpy_and_is_deterministic():
    flat, tab, mats, _ = _thin_sites(ROWS, cond=30.0)
    linv = torch.full((len(ROWS), 16, 16), float("nan"), device=DEV)
    ritz = torch.full((len(ROWS), 2), float("nan"), device=DEV)
    fin = _C.thin_finish(tab, 1, 8, 1e-4, linv_out=linv, ritz_out=ritz)
    _C.thin_gram(tab, flat, None, fin)
    first = linv.clone()
    _C.thin_gram(tab, flat, None, fin)  # the counters were reset by the last arrivers; fixed summation order
    assert torch.equal(first, linv)
    assert int(tab.counters.abs().sum()) == 0
    for i, m in enumerate(mats):
        G = (m.T @ m).numpy()
        ev = np.sort(np.linalg.eigvalsh(G))[::-1]
        assert abs(float(ritz[i, 0]) - ev[:8].sum()) <= 2e-5 * ev[:8].sum(), i
        assert abs(float(ritz[i, 1]) - ev[8:].sum()) <= 2e-5 * ev[:8].sum(), i
        if m.shape[0] < 16:
            continue  # rank-deficient: dropped pivots, checked in the orthonormalisation test
        Gs = G + 1e-4 * np.trace(G) / 16 * np.eye(16)
        want = np.linalg.inv(np.linalg.cholesky(Gs))
        got = linv[i].double().cpu().numpy()
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), i
        assert np.abs(np.triu(got, 1)).max() == 0.0


@pytest.mark.parametrize("cond", [1.0, 1e3])
def test_choleskyqr3_in_four_launches_orthonormalises_every_site(cond):
    rows = [r for r in ROWS if r >= 16]
    flat, tab, mats, offs = _thin_sites(rows, seed=1, cond=cond)
    a, b = flat.clone(), torch.zeros_like(flat)
    lin = [torch.empty(len(rows), 16, 16, device=DEV) for _ in range(2)]
    f = lambda sh, out: _C.thin_finish(tab, 1, 8, sh, linv_out=out)  # noqa: E731
    _C.thin_gram(tab, a, None, f(1e-4, lin[0]))
    _C.thin_apply(tab, a, lin[0], b, f(0.0, lin[1]))
    _C.thin_apply(tab, b, lin[1], a, f(0.0, lin[0]))
    _C.thin_apply(tab, a, lin[0], b)
    for off, r_, m in zip(offs, rows, mats):
        q = b[off:off + r_ * 16].view(r_, 16).double().cpu()
        assert (q.T @ q - torch.eye(16, dtype=torch.float64)).abs().max() < 5e-6, (r_, cond)
        # same column space: projecting the input onto Q loses nothing
        assert (m - q @ (q.T @ m)).norm() <= 1e-5 * m.norm()
        # and it IS Y R^-1 with R upper triangular (CholeskyQR): Q^T Y is upper triangular
        R = q.T @ m
        assert np.abs(np.tril(R.numpy(), -1)).max() <= 1e-4 * R.abs().max()


def test_core_svd_inside_the_launch_vs_numpy():
    rows = [320, 2560, 23040, 64]
    fa, tab, ma, _ = _thin_sites(rows, seed=2, cond=100.0)
    fb, _, mb, _ = _thin_sites(rows, seed=3)
    n, r = len(rows), 8
    ubt = torch.empty(n, r, 16, device=DEV)
    vb = torch.empty(n, r, 16, device=DEV)
    s = torch.empty(n, 16, device=DEV)
    _C.thin_gram(tab, fa, fb, _C.thin_finish(tab, 2, r, ubt=ubt, vb=vb, s_out=s))
    for i in range(n):
        core = (ma[i].T @ mb[i]).numpy()
        U, Sg, Vh = np.linalg.svd(core)
        assert np.abs(s[i].double().cpu().numpy() - Sg).max() <= 2e-6 * Sg[0], i
        got = (ubt[i].double().cpu().numpy().T * s[i, :r].double().cpu().numpy()) @ vb[i].double().cpu().numpy()
        want = (U[:, :r] * Sg[:r]) @ Vh[:r]
        assert np.abs(got - want).max() <= 5e-6 * np.abs(want).max(), i
        u_ = ubt[i].double().cpu().numpy()
        assert np.abs(u_ @ u_.T - np.eye(r)).max() < 1e-5


@pytest.mark.parametrize("r", [4, 8, 5, 16])
def test_thin_rotate_and_the_sign_rule(r):
    rows = [320, 1000, 23040, 17]
    flat, tab, mats, offs = _thin_sites(rows, seed=4)
    g = torch.Generator().manual_seed(9)
    M = torch.randn(len(rows), r, 16, generator=g)
    sa = torch.rand(len(rows), 16, generator=g) + 0.5
    out = torch.full((flat.numel() // 16 * r,), float("nan"), device=DEV)
    sign = torch.zeros(len(rows), 16, device=DEV)
    ws = (torch.empty(tab.total_blocks * 32, device=DEV), torch.empty(tab.total_blocks * 16, dtype=torch.int32, device=DEV))
    _C.thin_rotate(tab, flat, M.to(DEV), r, out, sign_ws=ws, sign_out=sign)
    out2 = torch.empty_like(out)
    _C.thin_rotate(tab, flat, M.to(DEV), r, out2, scale_a=sa.to(DEV), scale_b=sign)
    for i, (off, r_, m) in enumerate(zip(offs, rows, mats)):
        want = m @ M[i].double().T
        got = out[off // 16 * r: off // 16 * r + r_ * r].view(r_, r).double().cpu()
        assert (got - want).abs().max() <= 1e-5 * want.abs().max()
        j = want.abs().argmax(dim=0)
        sg = torch.sign(want[j, torch.arange(r)])
        assert torch.equal(sign[i, :r].double().cpu(), sg), i
        got2 = out2[off // 16 * r: off // 16 * r + r_ * r].view(r_, r).double().cpu()
        assert (got2 - want * (sa[i, :r].double() * sg)).abs().max() <= 1e-5 * want.abs().max()


def _select(us, vs, signs, r, q):
    """Run the three selection passes + lerp on per-site (u [N r], v [K r]) pairs; returns hi [n]."""
    offs_u, offs_v, tu, tv = [], [], 0, 0
    for u, v in zip(us, vs):
        offs_u.append(tu)
        offs_v.append(tv)
        tu += u.numel()
        tv += v.numel()
    U, V = torch.cat([u.flatten() for u in us]).to(DEV), torch.cat([v.flatten() for v in vs]).to(DEV)
    qt = _C.ThinQTable([(ou, u.numel(), ov, v.numel()) for ou, ov, u, v in zip(offs_u, offs_v, us, vs)], DEV)
    state = torch.zeros(len(us), 8, dtype=torch.int32)
    ws = []
    for i, (u, v) in enumerate(zip(us, vs)):
        n = u.numel() + v.numel()
        ranks = torch.tensor(q, dtype=torch.float32) * (n - 1)
        state[i, 1] = int(ranks.floor().item())
        ws.append(float((ranks - ranks.floor()).item()))
    state[:, 3] = -1
    state = state.to(DEV)
    sg = torch.ones(len(us), 16)
    for i, s_ in enumerate(signs):
        sg[i, :r] = s_
    sg = sg.to(DEV)
    out2 = torch.empty(len(us), 2, device=DEV)
    for p in range(3):
        _C.thin_select(qt, U, V, sg, r, p, state, out2)
    return torch.lerp(out2[:, 0], out2[:, 1], torch.tensor(ws, device=DEV)), (qt, U, V, sg, offs_u, offs_v)


@pytest.mark.parametrize("q", [0.99, 0.5, 1.0, 0.999])
def test_selection_equals_torch_quantile_bit_for_bit(q):
    r = 8
    g = torch.Generator().manual_seed(5)
    shapes = [(320, 320), (1280, 23040), (64, 32), (10240, 1280), (32, 32)]
    us = [torch.randn(N, r, generator=g) * 0.3 for N, K in shapes]
    vs = [torch.randn(K, r, generator=g) * 0.05 for N, K in shapes]
    us[2][:] = us[2].round()                      # heavy ties
    vs[2][:] = 0.0
    us[4][:], vs[4][:] = 1.5, 1.5                 # one value only
    signs = [torch.sign(torch.randn(r, generator=g)) for _ in shapes]
    hi, _ = _select(us, vs, signs, r, q)
    for i, (u, v, s_) in enumerate(zip(us, vs, signs)):
        want = torch.quantile(torch.cat([u.flatten(), (v * s_).flatten()]).to(DEV), q)
        assert torch.equal(hi[i], want), (i, float(hi[i]), float(want))


def test_selection_with_a_nan_is_nan_and_the_clamp_propagates_it():
    r = 4
    g = torch.Generator().manual_seed(6)
    us = [torch.randn(64, r, generator=g), torch.randn(96, r, generator=g)]
    vs = [torch.randn(32, r, generator=g), torch.randn(160, r, generator=g)]
    us[1][7, 2] = float("nan")
    signs = [torch.tensor([1., -1., 1., -1.]), torch.ones(r)]
    hi, (qt, U, V, sg, ou, ov) = _select(us, vs, signs, r, 0.99)
    assert torch.equal(hi[0], torch.quantile(torch.cat([us[0].flatten(), (vs[0] * signs[0]).flatten()]).to(DEV), 0.99))
    assert bool(torch.isnan(hi[1]))     # torch.quantile's rule
    down = torch.full_like(V, 7.0)
    _C.thin_clamp(qt, U, V, sg, hi, down, r)
    h0 = hi[0].cpu()
    assert torch.equal(U[:64 * r].cpu().view(64, r), torch.minimum(torch.maximum(us[0], -h0), h0))
    want_down = torch.minimum(torch.maximum(vs[0] * signs[0], -h0), h0).T.contiguous()
    assert torch.equal(down[:32 * r].cpu().view(r, 32), want_down)
    assert bool(torch.isnan(U[64 * r:]).all()) and bool(torch.isnan(down[32 * r:]).all())


# ----------------------------------------------------------------------------- the distillation on spectra that do not decay
def _power_law(N, K, seed, p=0.5):
    g = torch.Generator().manual_seed(seed)
    n = min(N, K)
    U = torch.linalg.qr(torch.randn(N, n, generator=g, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(K, n, generator=g, dtype=torch.float64))[0]
    s = torch.arange(1, n + 1, dtype=torch.float64) ** -p
    return ((U * s) @ V.T).float() * 0.05


def _finetune_like(N, K, seed):
    """A delta like W_tuned - W_base of a full fine-tune: a few strong directions over a FLAT tail (sigma_tail ~ 0.3 sigma_8)."""
    g = torch.Generator().manual_seed(seed)
    n = min(N, K)
    U = torch.linalg.qr(torch.randn(N, n, generator=g, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(K, n, generator=g, dtype=torch.float64))[0]
    s = torch.full((n,), 0.3, dtype=torch.float64) * (1.0 + 0.2 * torch.rand(n, generator=g, dtype=torch.float64))
    s[:8] = torch.tensor([6., 5., 4., 3., 2.2, 1.7, 1.3, 1.0], dtype=torch.float64)
    s, _ = torch.sort(s, descending=True)
    return ((U * s) @ V.T).float() * 0.01


@pytest.mark.parametrize("kind", ["power_law", "finetune_like"])
def test_distillation_on_a_flat_spectrum_is_within_2_percent_of_the_exact_truncation(kind):
    """``|| dW - up down ||_F <= 1.02 || dW - dW_r ||_F`` with dW_r the exact rank-r truncation (what cli_svd.py:35-47's full
    ``torch.linalg.svd`` yields), clamp off (quantile 1.0).  The captured-subspace error of a FIXED iteration count does not
    vanish here ((sigma_17 / sigma_8)^(2 n + 1) with a ratio near 1): the iteration count adapts (Ritz energy settled to
     settled, cli_svd.RITZ_TOL / TAIL_TOL)."""
    r = 8
    mk = _power_law if kind == "power_law" else _finetune_like
    shapes = [(2, 320, 320), (1, 640, 2880), (1, 1280, 320)]
    deltas = [torch.stack([mk(N, K, 100 * gi + i) for i in range(B)]) for gi, (B, N, K) in enumerate(shapes)]
    base = [[torch.zeros(N, K, device=DEV) for _ in range(B)] for B, N, K in shapes]
    tuned = [[d[i].to(DEV) for i in range(d.shape[0])] for d in deltas]
    res = S.distill_model(list(zip(tuned, base)), r, 1.0, torch.Generator(device=DEV).manual_seed(0))
    for (B, N, K), d, (up, down) in zip(shapes, deltas, res):
        for i in range(B):
            dd = d[i].double()
            U, Sg, Vh = torch.linalg.svd(dd, full_matrices=False)
            best = (dd - (U[:, :r] * Sg[:r]) @ Vh[:r]).norm()
            got = (dd - up[i].double().cpu() @ down[i].double().cpu()).norm()
            assert got <= 1.02 * best, (kind, (N, K, i), float(got / best))


def test_fused_small_steps_equal_the_ragged_launches_of_rounds_2_to_4(monkeypatch):
    """Same generator, same fixed iteration count: THIN on / off give the same factors (to rounding), thresholds included.
    (Both on BOTH planes of the residuals: the hi-plane-only power iterations of round 6 steer the subspace at 2^-9 and move
    the factors by a few 1e-4 — their effect on the result is bounded by the flat-spectrum test above and the one below.)"""
    from tests.test_cli_svd import _planted

    monkeypatch.setattr(S, "HI_ONLY_ITERATIONS", False)

    r = 8
    shapes = [(3, 320, 320), (1, 1280, 2880), (2, 2560, 320)]
    tuned, base = [], []
    for gi, (B, N, K) in enumerate(shapes):
        tb = [_planted(N, K, r + 4, 2e-3 / (N ** 0.5 + K ** 0.5), 10 * gi + i, "cpu") for i in range(B)]
        tuned.append([t.to(DEV) for t, _ in tb])
        base.append([b.to(DEV) for _, b in tb])
    groups = list(zip(tuned, base))
    a = S.distill_model(groups, r, 0.99, torch.Generator(device=DEV).manual_seed(1), n_iter=4)
    monkeypatch.setattr(S, "THIN", False)
    b = S.distill_model(groups, r, 0.99, torch.Generator(device=DEV).manual_seed(1), n_iter=4)
    for (ua, da), (ub, db) in zip(a, b):
        assert ua.shape == ub.shape and da.shape == db.shape
        assert (ua - ub).abs().max() <= 1e-4 * ub.abs().max()
        assert (da - db).abs().max() <= 1e-4 * db.abs().max()


def test_adaptive_iteration_stops_early_on_a_decaying_spectrum_and_runs_longer_on_a_flat_one(monkeypatch):
    """``n_iter=None``: never fewer than MIN_ITER = 4 (rounds 2-4's fixed count); a planted (decaying) spectrum stops there; a
    power-law one stops there too at the default tolerance (1e-3 of a LARGE tail energy) and runs on when the tolerance asks for
    more — the geometric-tail estimate, not a single small gain, decides (one iteration ahead: the loop's last pass is known
    before it runs and reads both planes of the residuals)."""
    from tests.test_cli_svd import _planted

    t, b = _planted(640, 640, 12, 1e-5, 3, "cpu")
    st = S._subspace_thin([(t - b)[None].to(DEV)], 8, None, torch.Generator(device=DEV).manual_seed(0))
    assert st.iterations == S.MIN_ITER, st.iterations
    flat = _power_law(640, 640, 4)[None].to(DEV)
    st = S._subspace_thin([flat], 8, None, torch.Generator(device=DEV).manual_seed(0))
    assert S.MIN_ITER <= st.iterations <= S.MAX_ITER
    loose = st.iterations
    monkeypatch.setattr(S, "RES_TOL", 1e-6)
    st = S._subspace_thin([flat], 8, None, torch.Generator(device=DEV).manual_seed(0))
    assert loose < st.iterations <= S.MAX_ITER, (loose, st.iterations)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_residual_planes_and_norms_in_one_launch_equal_sub_then_split(dt):
    """lora_amd_split16_residual vs lora_amd_sub_ragged + lora_amd_split16_transpose: identical planes (of dW and dW^T), and
    |dW|_F^2 per site vs f64."""
    g = torch.Generator().manual_seed(8)
    dims = [(2, 320, 320), (1, 64, 2880), (3, 640, 96)]
    pairs = [([torch.randn(N, K, generator=g).to(dt).to(DEV) for _ in range(B)],
              [torch.randn(N, K, generator=g).to(dt).to(DEV) for _ in range(B)]) for B, N, K in dims]
    mk = lambda sh: [torch.full(s_, 7.0, dtype=torch.bfloat16, device=DEV) for s_ in sh]  # noqa: E731
    a = [mk([(B, N, K) for B, N, K in dims]), mk([(B, N, K) for B, N, K in dims]),
         mk([(B, K, N) for B, N, K in dims]), mk([(B, K, N) for B, N, K in dims])]
    b = [mk([(B, N, K) for B, N, K in dims]), mk([(B, N, K) for B, N, K in dims]),
         mk([(B, K, N) for B, N, K in dims]), mk([(B, K, N) for B, N, K in dims])]
    norm2 = _C.split16_residual(pairs, dims, *a)
    # the residual in the weights' OWN dtype, then .float(): the reference's order (cli_svd.py:30-33, 57-60)
    deltas = [torch.stack([(t - b_).float() for t, b_ in zip(ts, bs)]) for ts, bs in pairs]
    _C.split16_transpose(deltas, *b)
    for pa, pb in zip(a, b):
        for x, y in zip(pa, pb):
            assert torch.equal(x, y)
    want = torch.cat([d.double().square().sum((1, 2)) for d in deltas])
    assert (norm2.double() - want).abs().max() <= 2e-6 * want.max()


def test_packed_factor_fragments_give_the_same_product_as_the_f32_factor():
    """lora_amd_thin_pack + lora_amd_rowdot16_planes_packed vs lora_amd_rowdot16_planes on the f32 factor: the same hi / lo
    fragments reach the matrix cores either way — bit-equal outputs."""
    g = torch.Generator().manual_seed(12)
    dims = [(2, 320, 640), (1, 64, 2880), (3, 640, 96)]     # (B, M, C): X [B, M, C] planes, F [B, C, 16]
    X = [torch.randn(B, M, Cc, generator=g).to(DEV) for B, M, Cc in dims]
    hi = [x.to(torch.bfloat16) for x in X]
    lo = [(x - h.float()).to(torch.bfloat16) for x, h in zip(X, hi)]
    offs, tot = [], 0
    for B, M, Cc in dims:
        offs.append(tot)
        tot += B * Cc * 16
    flat = torch.randn(tot, generator=g).to(DEV)
    F = [flat[o:o + B * Cc * 16].view(B, Cc, 16) for o, (B, M, Cc) in zip(offs, dims)]
    out_a = [torch.empty(B, M, 16, device=DEV) for B, M, Cc in dims]
    out_b = [torch.empty(B, M, 16, device=DEV) for B, M, Cc in dims]
    pa = _C.PlanesProgram(DEV, 16)
    ha = pa.table(list(zip(hi, lo, F, out_a)))
    pa.upload()
    pa.run(ha)
    tab = _C.ThinTable([(o + b * Cc * 16, Cc) for o, (B, M, Cc) in zip(offs, dims) for b in range(B)], DEV)
    pk = torch.full((tot * 2,), 7.0, dtype=torch.bfloat16, device=DEV)
    _C.thin_pack(tab, flat, pk)
    PK = [pk[2 * o: 2 * o + B * Cc * 32].view(B, Cc * 32) for o, (B, M, Cc) in zip(offs, dims)]
    pb = _C.PlanesProgram(DEV, 16, packed=True)
    hb = pb.table(list(zip(hi, lo, PK, out_b)))
    pb.upload()
    pb.run(hb)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)


def test_a_model_with_shapes_outside_the_fused_path_is_split_between_the_two_paths():
    """``distill_model``: the groups the fused path takes (multiples of 32) and the ones it does not (40 x 72) in ONE call:
    every group vs the exact truncation, clamp off."""
    from tests.test_cli_svd import _planted

    r = 8
    shapes = [(2, 320, 320), (1, 40, 72), (1, 640, 2880), (2, 72, 40)]
    tuned, base = [], []
    for gi, (B, N, K) in enumerate(shapes):
        tb = [_planted(N, K, r + 4, 2e-3 / (N ** 0.5 + K ** 0.5), 10 * gi + i, "cpu") for i in range(B)]
        tuned.append([t.to(DEV) for t, _ in tb])
        base.append([b.to(DEV) for _, b in tb])
    res = S.distill_model(list(zip(tuned, base)), r, 1.0, torch.Generator(device=DEV).manual_seed(3))
    assert all(x is not None for x in res)
    for (B, N, K), ts, bs, (up, down) in zip(shapes, tuned, base, res):
        assert up.shape == (B, N, r) and down.shape == (B, r, K)
        for i in range(B):
            dd = (ts[i] - bs[i]).double().cpu()
            U, Sg, Vh = torch.linalg.svd(dd, full_matrices=False)
            ref = (U[:, :r] * Sg[:r]) @ Vh[:r]
            got = up[i].double().cpu() @ down[i].double().cpu()
            assert (got - ref).norm() <= 5e-4 * ref.norm(), ((N, K, i), float((got - ref).norm() / ref.norm()))


def test_the_in_launch_hand_off_survives_thousands_of_repeats():
    """the last-arriver hand-off of csrc/svd_small.hip (write-through slabs, vmcnt(0), relaxed agent-scope arrival,
    one acquire fence in the last arriver) under stress — sites of up to 90 blocks, 3000 launches back to back, every result
    bit-equal to the first and the counters back at zero."""
    flat, tab, _, _ = _thin_sites([23040, 4096, 1280, 257, 23040], cond=30.0)
    linv = torch.full((5, 16, 16), float("nan"), device=DEV)
    fin = _C.thin_finish(tab, 1, 8, 1e-4, linv_out=linv)
    _C.thin_gram(tab, flat, None, fin)
    first = linv.clone()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(3000):
        _C.thin_gram(tab, flat, None, fin)
        bad += (linv != first).sum()
    assert int(bad) == 0 and int(tab.counters.abs().sum()) == 0


def test_hi_plane_only_power_iterations_cost_the_rank_r_error_nothing_measurable(monkeypatch):
    """cli_svd.HI_ONLY_ITERATIONS (round 6: the sketch and the power iterations read the hi plane of dW only, the pass that forms
    the factors both planes): the Frobenius error of up @ down against dW is within 0.1 % of the both-planes run's on a decaying
    AND on a flat spectrum (an O(2^-9) perturbation of range(Q) enters the error to second order)."""
    from tests.test_cli_svd import _planted

    r = 8
    t, b = _planted(1280, 640, r + 4, 1e-4, 5, "cpu")
    cases = [(t.to(DEV), b.to(DEV)), (_power_law(1280, 640, 2).to(DEV), torch.zeros(1280, 640, device=DEV))]
    for tuned, base in cases:
        errs = []
        for hi in (True, False):
            monkeypatch.setattr(S, "HI_ONLY_ITERATIONS", hi)
            (up, down), = S.distill_model([([tuned], [base])], r, 1.0, torch.Generator(device=DEV).manual_seed(1), n_iter=4)
            errs.append(float(((tuned - base).double() - up[0].double() @ down[0].double()).norm()))
        assert abs(errs[0] - errs[1]) <= 1e-3 * errs[1], errs
