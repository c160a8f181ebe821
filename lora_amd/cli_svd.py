"""Drop-in for the reference's ``lora_distill`` console script (``lora_diffusion/cli_svd.py``): distil the difference
between a fine-tuned model and its base into rank-r LoRA factors, site by site — BASELINE configs[4] / SURVEY §8f-1.

Reference recipe per site (cli_svd.py:30-53 Linear, :55-92 Conv2d flattened from dim 1): full ``torch.linalg.svd`` of
``dW = W_tuned - W_base`` in f32, keep the top r triplets, ``up = U_r diag(S_r)``, ``down = Vh_r``, clamp both at the
0.99-quantile of their joint value distribution.

MI355X path (device tensors): only r of min(N, K) singular triplets are wanted, so the full SVD (O(N K min(N,K)) flops,
one LAPACK-style call per site, 224 of them) is replaced by randomized subspace iteration (Halko-Martinsson-Tropp)
BATCHED over every same-shape site of the model (``distill_group``): the residuals of a group are stacked [B, N, K] and
each step of the iteration is ONE launch for the whole stack —

    Y  = dW Omega            ``lora_amd_colreduce_batched`` over the transposed stack   [B, N, l]
    Z  = dW^T Q              ``lora_amd_colreduce_batched`` over the stack              [B, K, l]   (l = 2r rounded up to 8)
    Q  = orth(Y)             shifted CholeskyQR3 on the device: Gram = Y^T Y (colreduce_batched of Y with itself),
                             L^{-1} of the small Gram (``lora_amd_chol_inverse_batched``, one wave per matrix),
                             Q = Y L^{-T} (rowdot_batched); shift on the first pass, two clean passes after it

— all HBM-streaming passes over dW with a skinny factor (``csrc/linear.hip``), plus one batched l x l SVD (torch) of
the triangular-factor-sized core at the end.  Model-level entry (``distill_model`` / ``topr_svd_ragged``): every step is ONE
launch over all shape groups, and with a 16-wide sketch the products with dW / dW^T run on the matrix cores over (hi, lo)
bf16 planes of the residuals (``PLANES``; ``csrc/rank16_mfma.hip``: ``lora_amd_rowdot16_planes[_packed]``, 4.1 TB/s per pass).
Round 5 (``THIN``, ``csrc/svd_small.hip``): the residuals, their planes and |dW|^2 come out of ONE launch
(``lora_amd_split16_residual``); every small dense step between two passes — Gram / Cholesky / apply (CholeskyQR3 = 4 launches),
the 16 x 16 core SVDs (one-sided Jacobi inside the launch that sums the cores), the sign rule, the quantile (exact radix
selection, bit-equal to ``torch.quantile``) and the clamp — is a launch over a per-site table whose reduction is finished by the
last-arriving workgroup of each site; the iteration count adapts to the spectrum (``n_iter=None``): 224 sites in ~13 ms.  With ``n_iter`` power iterations the captured subspace error decays like
(s_{l+1}/s_r)^(2 n_iter + 1); the defaults reproduce ``up @ down`` of the reference to ~1e-4 relative on
distillation-like spectra.  The signs of singular vectors are arbitrary (LAPACK's are too, and the reference's clamp
threshold - a quantile of SIGNED entries - inherits that arbitrariness); the device path fixes them by making the
largest-magnitude entry of every ``down`` row positive, so results are reproducible.  Device parity
(tests/test_gpu_parity_r2.py) compares, after aligning each pair's sign to the reference's, the factors, the clamp
threshold and the clamped factors with the reference recipe.

CPU tensors take the reference's exact full-SVD path (plumbing).
"""
from __future__ import annotations

import sys
from typing import Optional, Tuple

import torch

from . import _C
from .lora import (LoraInjectedConv2d, LoraInjectedLinear, inject_trainable_lora, inject_trainable_lora_extended,
                   save_all)


# The four products with dW / dW^T of the ragged iteration on the matrix cores over (hi, lo) bf16 planes of the residuals
# (csrc/rank16_mfma.hip: lora_amd_split16_ragged + lora_amd_rowdot16_planes) when the sketch is 16 wide and every shape is a
# multiple of 32 both ways; False = the f32 column-reduction passes of rounds 2-3 (tests run both).
PLANES = True
# The small dense steps between the passes (Gram / Cholesky / apply, the 16 x 16 core SVDs, sign rule, quantile, clamp) as
# fused launches over a per-site table (csrc/svd_small.hip) when the sketch is 16 wide; False = rounds 2-4's ragged launches +
# torch.linalg.svd / torch.topk per shape group (tests run both).
THIN = True
# adaptive iteration count (``n_iter=None``): the squared error of the rank-r product is |dW|_F^2 - E_r with E_r the top-r Ritz
# energy (sum of the r largest eigenvalues of the Gram matrix of dW Qz), which only grows with the iterations.  Iterate until,
# for EVERY site, one more power iteration improved that squared error by less than RES_TOL of itself (floor: 1e-6 |dW|^2, the
# resolution of the f32 sums; an exactly low-rank residual).  |dW|_F^2 comes out of the launch that forms the residual.  At
# least MIN_ITER = 4 (the fixed count of rounds 2-4), at most MAX_ITER iterations; "improved by" = the geometric-tail estimate
# from the ratio rho of two successive gains, evaluated ONE ITERATION AHEAD (after iteration i: would everything after i + 1,
# gain rho^2 / (1 - rho), be below the tolerance?) so that the loop knows its last iteration before it runs it — that one
# reads both planes of the residuals (HI_ONLY_ITERATIONS below), all others the hi plane.
RES_TOL, MIN_ITER, MAX_ITER = 1e-3, 4, 12   # MIN_ITER = the fixed count of rounds 2-4: the adaptive rule only ever ADDS iterations
# the passes that only steer the subspace (the sketch, every `dW^T Q`, and `dW Qz` of the iterations before the one that may be
# the last) read the hi plane of the residuals alone: half the bytes; the products the returned factors are formed from (the last
# iterations' `dW Qz` and b = Q^T dW) read both planes.  False: every pass on both planes (rounds 4-5).
HI_ONLY_ITERATIONS = True
LAST_ITERATIONS = None  # power iterations the last fused distillation ran (evidence for bench.py)


def _iter_lora(model):  # ref :16-21
    for module in model.modules():
        if isinstance(module, (LoraInjectedConv2d, LoraInjectedLinear)):
            yield module


def _fix_signs(U: torch.Tensor, Vh: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    idx = Vh.abs().argmax(dim=1)
    sgn = torch.sign(Vh.gather(1, idx[:, None]).squeeze(1))
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
    return U * sgn[None, :], Vh * sgn[:, None]


def _orth(y: torch.Tensor) -> torch.Tensor:
    """Columns of every matrix of the stack ``y`` [B, M, l] orthonormalised on the device: shifted CholeskyQR3
    (Fukaya et al.): the first pass adds a shift to the Gram matrix so that the f32 Cholesky cannot break down on an
    ill-conditioned block (power iteration drives the columns towards each other), the two clean passes restore
    orthonormality to f32 precision.  9 small batched launches; no host round trip."""
    for shift in (1e-4, 0.0, 0.0):
        gram = _C.colreduce_batched(y, y, _C.FACTOR_RK)                  # [B, l, l] = Y^T Y
        linv = _C.chol_inverse_batched(gram, shift)                       # L^{-1}
        y = _C.rowdot_batched(y, linv, _C.FACTOR_RK)                      # Y L^{-T}
    return y


def _sketch_width(rank: int, oversample: int, N: int, K: int) -> int:
    """Sketch columns: rank + max(oversample, rank) rounded up to the kernels' 8-column granule, at most 32."""
    l = -(-(rank + max(oversample, rank)) // 8) * 8
    return min(l, 32)


def group_supported(N: int, K: int, rank: int, oversample: int = 8) -> bool:
    """Shapes the batched device path takes: 16-byte-friendly rows and a sketch that fits the small dense kernels."""
    l = _sketch_width(rank, oversample, N, K)
    return K % 8 == 0 and rank <= l <= min(N, K)


def topr_svd_batched(delta: torch.Tensor, rank: int, oversample: int = 8, n_iter: Optional[int] = 4,
                     generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Top-``rank`` singular triplets (U [B,N,r], S [B,r], Vh [B,r,K]) of a stack of f32 matrices [B, N, K] on the
    device; every step is one launch for the whole stack (see module docstring)."""
    _C.require()
    n_iter = 4 if n_iter is None else n_iter
    B, N, K = delta.shape
    l = _sketch_width(rank, oversample, N, K)
    if not group_supported(N, K, rank, oversample):
        raise ValueError(f"topr_svd_batched: shape {N}x{K} rank {rank} is outside the batched device path")
    delta = delta.float().contiguous()
    # both products of the iteration as column-reduction passes (the faster primitive: ~2 TB/s on f32 rows, the row-dot
    # form re-stages its factor tile per few rows of a tall-and-wide weight): dW^T Q streams dW, dW Qz streams the
    # resident transposed stack (a second f32 layout of the residuals; 288 GB of HBM)
    delta_t = delta.transpose(1, 2).contiguous()
    omega = torch.randn(B, K, l, device=delta.device, dtype=torch.float32, generator=generator)
    q = _orth(_C.colreduce_batched(delta_t, omega, _C.FACTOR_KR))        # [B, N, l] = orth(dW Omega)
    for _ in range(n_iter):
        qz = _orth(_C.colreduce_batched(delta, q, _C.FACTOR_KR))         # [B, K, l] = orth(dW^T Q)
        q = _orth(_C.colreduce_batched(delta_t, qz, _C.FACTOR_KR))       # [B, N, l] = orth(dW Qz)
    b = _C.colreduce_batched(delta, q, _C.FACTOR_RK)                     # [B, l, K] = Q^T dW
    # small SVD of b [l, K]: b^T = Qb Rb with Qb = orth(b^T) (the same CholeskyQR3), SVD of the l x l factor Rb^T = b Qb
    # (batched, tiny), b = Ub S (Qb Vb)^T.  (A batched gesvd on [B, l, K] is 10-20x slower here; the Gram/eigh route
    # loses the orthonormality of the weak directions.)
    qb = _orth(b.transpose(1, 2).contiguous())                           # [B, K, l]
    ub, s, vbh = torch.linalg.svd(torch.bmm(b, qb), full_matrices=False)  # [B, l, l]
    u = torch.bmm(q, ub[:, :, :rank])
    vh = torch.bmm(vbh[:, :rank], qb.transpose(1, 2))
    s = s[:, :rank]
    idx = vh.abs().argmax(dim=2, keepdim=True)
    sgn = torch.sign(vh.gather(2, idx)).squeeze(2)
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
    return u * sgn[:, None, :], s, vh * sgn[:, :, None]


def topr_svd(delta: torch.Tensor, rank: int, oversample: int = 8, n_iter: Optional[int] = 4,
             generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Top-``rank`` singular triplets (U [N,r], S [r], Vh [r,K]) of a 2-D f32 matrix.

    Device tensors: the batched randomized path with a stack of one (shapes it does not cover: exact SVD on the
    device).  CPU tensors: exact ``torch.linalg.svd`` (what the reference runs)."""
    N, K = delta.shape
    if not delta.is_cuda or not group_supported(N, K, rank, oversample):
        U, S, Vh = torch.linalg.svd(delta.float(), full_matrices=False)
        U, S, Vh = U[:, :rank], S[:rank], Vh[:rank]
        if delta.is_cuda:  # same deterministic sign rule as the batched path
            U, Vh = _fix_signs(U, Vh)
        return U, S, Vh  # CPU: LAPACK's signs, as in the reference
    U, S, Vh = topr_svd_batched(delta[None], rank, oversample, n_iter, generator)
    return U[0], S[0], Vh[0]


def _quantile_rows(x: torch.Tensor, q: float) -> torch.Tensor:
    """``torch.quantile(x, q, dim=1)`` (linear interpolation) bit for bit, from the top (1 - q) n order statistics only: for
    the clamp's q = 0.99 a top-k of 1 % of a row instead of a full segmented sort of it (the sorts were ~3.5 ms of a 30 ms
    distillation of 224 sites).  Same rank arithmetic as ATen's quantile: ranks and weights in the input dtype."""
    n = x.shape[1]
    if not x.is_cuda or not (0.5 <= q <= 1.0) or n < 64:
        return torch.quantile(x, q, dim=1)
    bad = torch.isnan(x).any(dim=1)   # torch.quantile: a row holding a NaN gives NaN (top-k would sort it as the largest value)
    ranks = torch.tensor(q, dtype=x.dtype) * (n - 1)
    below = ranks.floor()
    lo = int(below.item())
    hi = min(lo + 1, n - 1)
    vals = torch.topk(x, n - lo, dim=1, largest=True, sorted=True).values   # descending: column j = order statistic n - 1 - j
    out = torch.lerp(vals[:, n - 1 - lo], vals[:, n - 1 - hi], float((ranks - below).item()))   # an exact f32 value
    return torch.where(bad, torch.full_like(out, float("nan")), out)


def _clamp_pairs(U: torch.Tensor, Vh: torch.Tensor, clamp_quantile: float):
    """ref :39-47 for a stack: per site, clamp both factors at the quantile of their joint (signed) values."""
    dist = torch.cat([U.flatten(1), Vh.flatten(1)], dim=1)
    hi = _quantile_rows(dist, clamp_quantile)
    return torch.minimum(torch.maximum(U, -hi[:, None, None]), hi[:, None, None]), \
        torch.minimum(torch.maximum(Vh, -hi[:, None, None]), hi[:, None, None])


def distill_group(w_tuned, w_base, rank: int, clamp_quantile: float = 0.99,
                  generator: Optional[torch.Generator] = None, **svd_kw):
    """(up [B, N, r], down [B, r, K]) of a group of same-shape sites: lists of tuned / base weights (any trailing
    dims are flattened from dim 1, ref :57-60).  Device tensors of supported shapes: ONE batched iteration."""
    res = torch.stack([(t.float() - b.float()).flatten(start_dim=1) for t, b in zip(w_tuned, w_base)])
    B, N, K = res.shape
    if res.is_cuda and group_supported(N, K, rank, svd_kw.get("oversample", 8)):
        U, S, Vh = topr_svd_batched(res, rank, generator=generator, **svd_kw)
    else:
        trip = [topr_svd(res[i], rank, generator=generator, **svd_kw) for i in range(B)]
        U, S, Vh = (torch.stack([t[k] for t in trip]) for k in range(3))
    return _clamp_pairs(U * S[:, None, :], Vh, clamp_quantile)


def _flat_stacks(shapes, device):
    """One f32 allocation holding a [B, R, C] stack per entry of ``shapes``; every stack starts on a 256-byte boundary."""
    offs, total = [], 0
    for B, R, Cc in shapes:
        offs.append(total)
        total += -(-(B * R * Cc) // 64) * 64
    flat = torch.empty(max(total, 64), dtype=torch.float32, device=device)
    return flat, [flat[o:o + B * R * Cc].view(B, R, Cc) for o, (B, R, Cc) in zip(offs, shapes)]


def _elem_off(view: torch.Tensor, flat: torch.Tensor) -> int:
    return (view.data_ptr() - flat.data_ptr()) // 4


class _ThinState:
    """Buffers and tables of one fused subspace iteration over a list of stacks (``_subspace_thin``)."""


def _subspace_thin(deltas, rank: int, n_iter, generator, pairs=None):
    """The subspace iteration of ``topr_svd_ragged`` with a 16-wide sketch on the matrix cores end to end: the products with
    dW / dW^T over (hi, lo) bf16 planes (``lora_amd_rowdot16_planes``), every CholeskyQR3 as 4 fused launches over a per-site
    table (``lora_amd_thin_gram`` / ``lora_amd_thin_apply``: the Gram matrix of a pass's output is accumulated by the pass and
    inverted by its last-arriving workgroup), the 16 x 16 core SVDs by a one-sided Jacobi inside the launch that sums the cores.
    ``n_iter=None``: power iterations until the top-r Ritz energy of every site (eigenvalues of the Gram matrix of dW Qz, from
    the same launch) settles to RITZ_TOL — a flat spectrum needs more than the 4 iterations a decaying one does.
    ``pairs`` = [(tuned list, base list)] per group instead of ``deltas`` = [B, N, K] f32 stacks: the residuals are formed
    inside the launch that writes the planes (``lora_amd_split16_residual``), never in f32.
    Leaves Q in ``st.yb``, Qb in ``st.za``, the core factors in ``st.ubt / st.vb / st.s``."""
    if pairs is not None:
        dev = pairs[0][0][0].device
        dims = [(len(t), t[0].shape[0], t[0].numel() // t[0].shape[0]) for t, _ in pairs]
    else:
        dev = deltas[0].device
        dims = [tuple(d.shape) for d in deltas]
    l = 16
    yshape = [(B, N, l) for B, N, K in dims]
    zshape = [(B, K, l) for B, N, K in dims]
    st = _ThinState()
    st.dims, st.rank = dims, rank
    st.ya, Ya = _flat_stacks(yshape, dev)
    st.yb, Yb = _flat_stacks(yshape, dev)
    st.za, Za = _flat_stacks(zshape, dev)
    st.zb, Zb = _flat_stacks(zshape, dev)
    st.zc, Zc = _flat_stacks(zshape, dev)
    ysites, zsites = [], []
    for (B, N, K), y, z in zip(dims, Ya, Za):
        oy, oz = _elem_off(y, st.ya), _elem_off(z, st.za)
        ysites += [(oy + b * N * l, N) for b in range(B)]
        zsites += [(oz + b * K * l, K) for b in range(B)]
    st.ysites, st.zsites = ysites, zsites
    st.ty, st.tz = _C.ThinTable(ysites, dev), _C.ThinTable(zsites, dev)
    nb = len(ysites)
    st.nb = nb
    lin = [torch.empty(nb * 256, dtype=torch.float32, device=dev) for _ in range(2)]
    st.ritz = torch.zeros(nb, 2, dtype=torch.float32, device=dev)
    st.ubt = torch.empty(nb * rank * l, dtype=torch.float32, device=dev)
    st.vb = torch.empty(nb * rank * l, dtype=torch.float32, device=dev)
    st.s = torch.empty(nb, l, dtype=torch.float32, device=dev)

    def planes_of(shapes_):
        return ([torch.empty(sh, dtype=torch.bfloat16, device=dev) for sh in shapes_],
                [torch.empty(sh, dtype=torch.bfloat16, device=dev) for sh in shapes_])
    dh, dl = planes_of([(B, N, K) for B, N, K in dims])
    th, tl = planes_of([(B, K, N) for B, N, K in dims])
    if pairs is not None:
        norm2 = _C.split16_residual(pairs, dims, dh, dl, th, tl)
    else:
        _C.split16_transpose([d.contiguous() for d in deltas], dh, dl, th, tl)
        norm2 = torch.cat([d.square().sum((1, 2)) for d in deltas]) if n_iter is None else None
    # the factor of a pass enters the matrix pipe as (hi, lo) 16-bit fragments, packed ONCE per pass (lora_amd_thin_pack) instead
    # of split per 16-row slab inside it: two coalesced 16-byte loads per k-step in place of eight strided ones + the split
    st.pky = torch.empty(st.yb.numel() * 2, dtype=torch.bfloat16, device=dev)
    st.pkz = torch.empty(st.zb.numel() * 2, dtype=torch.bfloat16, device=dev)
    PY = [st.pky[2 * _elem_off(y, st.ya): 2 * _elem_off(y, st.ya) + B * N * 32].view(B, N * 32) for (B, N, K), y in zip(dims, Ya)]
    PZ = [st.pkz[2 * _elem_off(z, st.za): 2 * _elem_off(z, st.za) + B * K * 32].view(B, K * 32) for (B, N, K), z in zip(dims, Za)]
    pprog = _C.PlanesProgram(dev, l, packed=True)
    p_y = pprog.table(list(zip(dh, dl, PZ, Ya)))          # Y = dW F, F [K, 16] packed in pkz (Omega, then Qz)
    p_z = pprog.table(list(zip(th, tl, PY, Za)))          # Z = dW^T Q, Q [N, 16] packed in pky
    p_b = pprog.table(list(zip(th, tl, PY, Zc)))          # b^T = dW^T Q
    pprog.upload()

    def orth(tab, a, b, ritz=None):
        """CholeskyQR3 a -> b -> a -> b (result in ``b``): shifted first pass, two clean ones; 4 launches."""
        fin = lambda shift, out, rz=None: _C.thin_finish(tab, 1, rank, shift, linv_out=out, ritz_out=rz)  # noqa: E731
        _C.thin_gram(tab, a, None, fin(1e-4, lin[0], ritz))
        _C.thin_apply(tab, a, lin[0], b, fin(0.0, lin[1]))
        _C.thin_apply(tab, b, lin[1], a, fin(0.0, lin[0]))
        _C.thin_apply(tab, a, lin[0], b)

    st.zc.normal_(generator=generator)                            # Omega, every site at once
    hi = HI_ONLY_ITERATIONS
    _C.thin_pack(st.tz, st.zc, st.pkz)
    pprog.run(p_y, hi)
    orth(st.ty, st.ya, st.yb)                                     # q in yb
    it, prev, prev_gain, final_next = 0, None, None, False
    while True:
        _C.thin_pack(st.ty, st.yb, st.pky)
        pprog.run(p_z, hi)
        orth(st.tz, st.za, st.zb)                                 # qz in zb
        _C.thin_pack(st.tz, st.zb, st.pkz)
        # the Q that is RETURNED must come from a product with dW itself — one multiplication by the exact matrix contracts the
        # hi-plane iterations' O(2^-9) subspace error by the spectral gap (exactly low-rank deltas: to nothing): the LAST `dW Qz`
        # reads both planes.  A fixed count knows its last iteration; the adaptive loop decides ONE ITERATION AHEAD (below), so
        # it knows too — rounds 5-6a stopped after the fact and repeated the pass on both planes (one pass and one
        # orthonormalisation more per distillation)
        last = (it + 1 >= n_iter) if n_iter is not None else (final_next or it + 1 >= MAX_ITER)
        pprog.run(p_y, hi and not last)
        orth(st.ty, st.ya, st.yb, st.ritz if (n_iter is None and not last) else None)
        it += 1
        if last:
            break
        if n_iter is not None:
            continue
        gained = (st.ritz[:, 0] - prev).abs() if prev is not None else None
        if it >= MIN_ITER - 1 and prev_gain is not None:
            # what will still be to come AFTER THE NEXT iteration, from the convergence ratio of two successive gains (a
            # geometric tail: the next gain is gained * rho, everything after it gained * rho^2 / (1 - rho), rho capped at 0.95):
            # a small gain alone does not stop a site whose gains are not shrinking.  (A previous gain already
            # inside the rounding noise of the Ritz energies, 1e-6 |dW|^2, says nothing about a ratio.)  Compares like with
            # like: every Ritz energy that enters comes from a hi-plane iteration.
            floor = 1e-6 * norm2
            rho = torch.where(prev_gain > floor, gained / prev_gain.clamp_min(1e-30), torch.zeros_like(gained)).clamp(0.0, 0.95)
            remaining = gained * rho * rho / (1.0 - rho)
            err2 = (norm2 - st.ritz[:, 0] - gained * rho).clamp_min(0.0)      # the squared error expected after the next one
            if bool((remaining <= torch.maximum(RES_TOL * err2, floor)).all()):  # one host sync per iteration >= MIN_ITER - 1
                final_next = True
        prev_gain = gained
        prev = st.ritz[:, 0].clone()
    st.iterations = it
    global LAST_ITERATIONS
    LAST_ITERATIONS = it
    _C.thin_pack(st.ty, st.yb, st.pky)
    pprog.run(p_b)                                                # b^T [K, l] in zc
    _C.thin_gram(st.tz, st.zc, None, _C.thin_finish(st.tz, 1, rank, 1e-4, linv_out=lin[0]))
    _C.thin_apply(st.tz, st.zc, lin[0], st.za, _C.thin_finish(st.tz, 1, rank, 0.0, linv_out=lin[1]))
    _C.thin_apply(st.tz, st.za, lin[1], st.zb, _C.thin_finish(st.tz, 1, rank, 0.0, linv_out=lin[0]))
    _C.thin_apply(st.tz, st.zb, lin[0], st.za)                    # qb in za
    # core = b Qb = zc^T za, its SVD inside the same launch
    _C.thin_gram(st.tz, st.zc, st.za, _C.thin_finish(st.tz, 2, rank, ubt=st.ubt, vb=st.vb, s_out=st.s))
    st.sign = torch.empty(nb, l, dtype=torch.float32, device=dev)
    st.sign_ws = (torch.empty(st.tz.total_blocks * 32, dtype=torch.float32, device=dev),
                  torch.empty(st.tz.total_blocks * 16, dtype=torch.int32, device=dev))
    st.uo = torch.empty(st.ya.numel() // l * rank, dtype=torch.float32, device=dev)
    st.vo = torch.empty(st.za.numel() // l * rank, dtype=torch.float32, device=dev)
    # V = Qb Vb[:r]^T [K, r] and the sign rule (largest-|.| entry of every down row positive) in one launch
    _C.thin_rotate(st.tz, st.za, st.vb, rank, st.vo, sign_ws=st.sign_ws, sign_out=st.sign)
    return st


def _group_views(st, flat, rows_of, cols: int):
    """Per shape group the [B, rows, cols] view of a flat per-site buffer laid out like the iteration's stacks."""
    out, i = [], 0
    sites = st.ysites if rows_of == "N" else st.zsites
    for (B, N, K) in st.dims:
        rows = N if rows_of == "N" else K
        o = sites[i][0] // 16 * cols
        out.append(flat[o:o + B * rows * cols].view(B, rows, cols))
        i += B
    return out


_QTABLES = {}


def _distill_thin(pairs, rank: int, clamp_quantile: float, n_iter, generator):
    """``topr_svd_ragged`` + ``_clamp_pairs`` with the small steps fused, from (tuned, base) weight lists per shape group:
    returns per group (up [B, N, r], down [B, r, K])."""
    st = _subspace_thin(None, rank, n_iter, generator, pairs=pairs)
    dev = st.ya.device
    # up = Q Ub[:, :r] diag(S) diag(sign)
    _C.thin_rotate(st.ty, st.yb, st.ubt, rank, st.uo, scale_a=st.s, scale_b=st.sign)
    key = (tuple(st.ysites), tuple(st.zsites), rank, clamp_quantile, str(dev))
    ent = _QTABLES.get(key)
    if ent is None:
        qs, ks, ws = [], [], []
        q32 = torch.tensor(clamp_quantile, dtype=torch.float32)
        for (oy, N), (oz, K) in zip(st.ysites, st.zsites):
            n = (N + K) * rank
            qs.append((oy // 16 * rank, N * rank, oz // 16 * rank, K * rank))
            ranks = q32 * (n - 1)          # ATen's quantile: ranks and weights in the input dtype
            lo = ranks.floor()
            ks.append(int(lo.item()))
            ws.append(float((ranks - lo).item()))
        tmpl = torch.zeros(len(qs), 8, dtype=torch.int32)
        tmpl[:, 1] = torch.tensor(ks, dtype=torch.int32)
        tmpl[:, 3] = -1
        if len(_QTABLES) > 8:
            _QTABLES.clear()
        ent = _QTABLES[key] = (_C.ThinQTable(qs, dev), tmpl.to(dev), torch.tensor(ws, dtype=torch.float32, device=dev))
    qt, tmpl, w = ent
    qt.counters.zero_()   # the table is cached across calls: a launch that died half-way must not poison the next one
    qt.hist.zero_()
    state = tmpl.clone()
    out2 = torch.empty(st.nb, 2, dtype=torch.float32, device=dev)
    for p in range(3):
        _C.thin_select(qt, st.uo, st.vo, st.sign, rank, p, state, out2)
    hi = torch.lerp(out2[:, 0], out2[:, 1], w)      # torch.quantile's interpolation, bit for bit
    down = torch.empty_like(st.vo)
    _C.thin_clamp(qt, st.uo, st.vo, st.sign, hi, down, rank)
    ups = _group_views(st, st.uo, "N", rank)
    downs = [d.view(B, rank, K) for d, (B, N, K) in zip(_group_views(st, down, "K", rank), st.dims)]
    return list(zip(ups, downs)), st


def _thin_ok(dims, rank: int, oversample: int) -> bool:
    return (THIN and PLANES and all(_sketch_width(rank, oversample, N, K) == 16 and N % 32 == 0 and K % 32 == 0
                                    for _, N, K in dims))


def topr_svd_ragged(deltas, rank: int, oversample: int = 8, n_iter: Optional[int] = 4,
                    generator: Optional[torch.Generator] = None):
    """``topr_svd_batched`` for a LIST of stacks of different shapes ([B_g, N_g, K_g] f32, contiguous) run in lock-step:
    every step of the iteration is ONE launch (pair) for the whole model — ``lora_amd_colreduce_ragged`` /
    ``lora_amd_rowdot_ragged`` over a descriptor table with one entry per shape group, ``chol_inverse_batched`` and the
    l x l core SVD over the concatenation of all groups.  SD1.5 extended (224 sites, 31 shapes): ~110 launches instead
    of ~3100.  Returns per group (U [B,N,r], S [B,r], Vh [B,r,K]) with the deterministic sign rule of the batched path."""
    _C.require()
    dev = deltas[0].device
    dims = [tuple(d.shape) for d in deltas]
    l = _sketch_width(rank, oversample, dims[0][1], dims[0][2])
    for (B, N, K) in dims:
        if not group_supported(N, K, rank, oversample) or _sketch_width(rank, oversample, N, K) != l:
            raise ValueError(f"topr_svd_ragged: shape {N}x{K} rank {rank} is outside the batched device path")
    nb = sum(B for B, _, _ in dims)
    planes = PLANES and l == 16 and all(N % 32 == 0 and K % 32 == 0 for _, N, K in dims)
    if planes and THIN:
        st = _subspace_thin([d.contiguous() for d in deltas], rank, n_iter, generator)
        _C.thin_rotate(st.ty, st.yb, st.ubt, rank, st.uo, scale_a=st.sign)        # U = Q Ub[:, :r] diag(sign)
        vt = torch.empty_like(st.vo)
        _C.thin_rotate(st.tz, st.za, st.vb, rank, vt, scale_a=st.sign)            # Vh^T = Qb Vb[:r]^T diag(sign)
        Us, Vs, o = _group_views(st, st.uo, "N", rank), _group_views(st, vt, "K", rank), 0
        out = []
        for (B, N, K), u, v in zip(dims, Us, Vs):
            out.append((u, st.s[o:o + B, :rank], v.transpose(1, 2).contiguous()))
            o += B
        return out
    if n_iter is None:
        n_iter = 4
    # second resident layout of the residuals (f32 passes only: the planes of dW^T come out of the split launch)
    delta_t = None if planes else [d.transpose(1, 2).contiguous() for d in deltas]
    yshape = [(B, N, l) for B, N, K in dims]
    zshape = [(B, K, l) for B, N, K in dims]
    _ya, Ya = _flat_stacks(yshape, dev)
    _yb, Yb = _flat_stacks(yshape, dev)
    _za, Za = _flat_stacks(zshape, dev)
    _zb, Zb = _flat_stacks(zshape, dev)
    zc_flat, Zc = _flat_stacks(zshape, dev)
    _u, Uo = _flat_stacks([(B, N, rank) for B, N, K in dims], dev)
    _v, Vo = _flat_stacks([(B, K, rank) for B, N, K in dims], dev)
    gram = torch.empty(nb, l, l, dtype=torch.float32, device=dev)
    linv = torch.empty(nb, l, l, dtype=torch.float32, device=dev)
    core = torch.empty(nb, l, l, dtype=torch.float32, device=dev)
    ubt = torch.empty(nb, rank, l, dtype=torch.float32, device=dev)
    vb = torch.empty(nb, rank, l, dtype=torch.float32, device=dev)
    boffs, o = [], 0
    for B, _, _ in dims:
        boffs.append(o)
        o += B
    per = lambda t: [t[o:o + B] for o, (B, _, _) in zip(boffs, dims)]  # noqa: E731
    G, L, Cr, Ub, Vb = per(gram), per(linv), per(core), per(ubt), per(vb)
    ws = [torch.empty(B * max(_C.colreduce_workspace_floats(N, K, l), _C.colreduce_workspace_floats(K, N, l)),
                      dtype=torch.float32, device=dev) for B, N, K in dims]

    prog = _C.RaggedProgram(dev)
    cr = lambda xs, fs, outs: prog.table(_C.RAGGED_COLREDUCE, l, list(zip(xs, fs, outs, ws)))  # noqa: E731
    rd = lambda xs, fs, outs, r=l: prog.table(_C.RAGGED_ROWDOT, r, [(x, f, o_, None) for x, f, o_ in zip(xs, fs, outs)])  # noqa: E731
    if planes:
        # dW and dW^T as (hi, lo) bf16 planes — the bytes of two f32 stacks, all four written by ONE launch that reads dW once
        # (lora_amd_split16_transpose); every product with them is then a row product on the matrix cores: Y = dW F streams
        # the planes of dW, Z = dW^T F those of dW^T
        def planes_of(shapes_):
            return ([torch.empty(sh, dtype=torch.bfloat16, device=dev) for sh in shapes_],
                    [torch.empty(sh, dtype=torch.bfloat16, device=dev) for sh in shapes_])
        dh, dl = planes_of([(B, N, K) for B, N, K in dims])
        th, tl = planes_of([(B, K, N) for B, N, K in dims])
        _C.split16_transpose([d.contiguous() for d in deltas], dh, dl, th, tl)
        pprog = _C.PlanesProgram(dev, l)
        p_sketch = pprog.table(list(zip(dh, dl, Zc, Ya)))     # Y = dW Omega
        p_fwd = pprog.table(list(zip(th, tl, Yb, Za)))        # Z = dW^T Q
        p_back = pprog.table(list(zip(dh, dl, Zb, Ya)))       # Y = dW Qz
        p_b = pprog.table(list(zip(th, tl, Yb, Zc)))          # b^T = dW^T Q
        pprog.upload()
        big = {"sketch": lambda: pprog.run(p_sketch), "fwd": lambda: pprog.run(p_fwd), "back": lambda: pprog.run(p_back),
               "b": lambda: pprog.run(p_b)}
    else:
        t_sketch = cr(delta_t, Zc, Ya)          # Y = dW Omega
        t_fwd = cr(deltas, Yb, Za)              # Z = dW^T Q
        t_back = cr(delta_t, Zb, Ya)            # Y = dW Qz
        t_b = cr(deltas, Yb, Zc)                # b^T = dW^T Q
        big = {"sketch": lambda: prog.run(t_sketch, _C.FACTOR_KR), "fwd": lambda: prog.run(t_fwd, _C.FACTOR_KR),
               "back": lambda: prog.run(t_back, _C.FACTOR_KR), "b": lambda: prog.run(t_b, _C.FACTOR_KR)}
    t_core = cr(Za, Zc, Cr)                 # b Qb  [l, l]
    gram_of = {id(t[0]): cr(t, t, G) for t in (Ya, Yb, Za, Zb, Zc)}
    apply_l = {(id(a[0]), id(b[0])): rd(a, L, b) for a, b in ((Ya, Yb), (Yb, Ya), (Za, Zb), (Zb, Za), (Zc, Za))}
    t_u = rd(Yb, Ub, Uo, rank)
    t_v = rd(Za, Vb, Vo, rank)
    prog.upload()
    lib_chol = _C.chol_inverse_batched

    def orth(chain):
        """CholeskyQR3 along ``chain`` = (src, tmp, ..., dst): shifted first pass, two clean ones (see ``_orth``)."""
        for shift, (a, b) in zip((1e-4, 0.0, 0.0), zip(chain[:-1], chain[1:])):
            prog.run(gram_of[id(a[0])], _C.FACTOR_RK)
            lib_chol(gram, shift, out=linv)
            prog.run(apply_l[(id(a[0]), id(b[0]))], _C.FACTOR_RK)

    zc_flat.normal_(generator=generator)                          # Omega, every group at once
    big["sketch"]()
    orth((Ya, Yb, Ya, Yb))                                        # q in Yb
    for _ in range(n_iter):
        big["fwd"]()
        orth((Za, Zb, Za, Zb))                                    # qz in Zb
        big["back"]()
        orth((Ya, Yb, Ya, Yb))
    big["b"]()                                                    # b^T [K, l] in Zc
    orth((Zc, Za, Zb, Za))                                        # qb in Za
    prog.run(t_core, _C.FACTOR_RK)
    ub, s, vbh = torch.linalg.svd(core, full_matrices=False)      # [sum B, l, l]: one batched call for the model
    ubt.copy_(ub[:, :, :rank].transpose(1, 2))
    vb.copy_(vbh[:, :rank])
    prog.run(t_u, _C.FACTOR_RK)                                   # U = Q Ub[:, :r]
    prog.run(t_v, _C.FACTOR_RK)                                   # Vh^T = Qb Vb[:r]^T
    out = []
    for (o, (B, N, K)), u, vt in zip(zip(boffs, dims), Uo, Vo):
        idx = vt.abs().argmax(dim=1, keepdim=True)                # [B, 1, r]: the largest-magnitude entry of each down row
        sgn = torch.sign(vt.gather(1, idx))
        sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)
        out.append((u * sgn, s[o:o + B, :rank], (vt * sgn).transpose(1, 2).contiguous()))
    return out


def distill_model(groups, rank: int, clamp_quantile: float = 0.99, generator: Optional[torch.Generator] = None,
                  **svd_kw):
    """``distill_group`` for every shape group of a model at once: ``groups`` = [(tuned weights, base weights), ...],
    one entry per shape.  Device groups of supported shapes share ONE ragged iteration (``topr_svd_ragged``), with the
    residuals of all sites formed by one launch; anything else falls back to ``distill_group``.  Returns a list of
    (up [B, N, r], down [B, r, K]) in the order of ``groups``."""
    over = svd_kw.get("oversample", 8)
    fast, results = [], [None] * len(groups)
    for gi, (tuned, base) in enumerate(groups):
        N = tuned[0].shape[0]
        K = tuned[0].numel() // N
        if (tuned[0].is_cuda and group_supported(N, K, rank, over) and tuned[0].dtype == base[0].dtype
                and all(t.is_contiguous() for t in tuned) and all(b.is_contiguous() for b in base)):
            fast.append((gi, len(tuned), N, K))
        else:
            results[gi] = distill_group(tuned, base, rank, clamp_quantile, generator, **svd_kw)
    by_dtype = {}
    for ent in fast:
        by_dtype.setdefault(groups[ent[0]][0][0].dtype, []).append(ent)
    for dt, all_ents in by_dtype.items():
        dev = groups[all_ents[0][0]][0][0].device
        # the groups the fused path takes (16-wide sketch, shapes multiples of 32) go through it together; the rest (if any)
        # through the ragged launches of rounds 2-4
        thin = [e for e in all_ents if _thin_ok([e[1:]], rank, over)]
        ents = [e for e in all_ents if e not in thin]
        if thin:
            # residuals, planes and |dW|^2 in one launch; the small steps fused; adaptive iteration count unless the caller
            # fixes one (``n_iter=``)
            flat2 = [([t.reshape(t.shape[0], -1) for t in groups[gi][0]], [b.reshape(b.shape[0], -1) for b in groups[gi][1]])
                     for gi, _, _, _ in thin]
            res, _ = _distill_thin(flat2, rank, clamp_quantile, svd_kw.get("n_iter"), generator)
            for (gi, _, _, _), ud in zip(thin, res):
                results[gi] = ud
        if not ents:
            continue
        _, deltas = _flat_stacks([(B, N, K) for _, B, N, K in ents], dev)
        pairs, outs = [], []
        for (gi, B, N, K), d in zip(ents, deltas):
            for i, (t, b) in enumerate(zip(*groups[gi])):
                pairs.append((t, b))
                outs.append(d[i])
        _C.sub_ragged(pairs, outs)                                 # every residual W_tuned - W_base: one launch
        trip = topr_svd_ragged(deltas, rank, generator=generator, **svd_kw)
        for (gi, _, _, _), (U, S, Vh) in zip(ents, trip):
            results[gi] = _clamp_pairs(U * S[:, None, :], Vh, clamp_quantile)
    return results


def distill_pair(w_tuned: torch.Tensor, w_base: torch.Tensor, rank: int, clamp_quantile: float = 0.99,
                 generator: Optional[torch.Generator] = None, **svd_kw) -> Tuple[torch.Tensor, torch.Tensor]:
    """(up [N, r], down [r, K]) of one site from its tuned / base weights flattened to 2-D (ref :30-47, :57-74)."""
    up, down = distill_group([w_tuned], [w_base], rank, clamp_quantile, generator, **svd_kw)
    return up[0], down[0]


def overwrite_base(base_model, tuned_model, rank, clamp_quantile, seed: int = 0, **svd_kw):
    """ref :24-92 — walk both models' adapters pairwise and overwrite the BASE model's LoRA factors with the rank-r
    distillation of (tuned - base) frozen weights.  Sites are grouped by (kind, weight shape): one batched device
    iteration per group (SD1.5 extended: 224 sites in ~20 groups) instead of 224 LAPACK calls."""
    groups, gens = {}, {}
    for lor_base, lor_tune in zip(_iter_lora(base_model), _iter_lora(tuned_model)):
        fb, ft = lor_base._frozen(), lor_tune._frozen()
        print(("Distill Linear shape " if isinstance(lor_base, LoraInjectedLinear) else "Distill Conv shape "),
              tuple(fb.weight.shape))
        key = (type(lor_base).__name__, tuple(fb.weight.shape), fb.weight.device, fb.weight.dtype)
        groups.setdefault(key, []).append((lor_base, fb, ft))
    by_dev = {}
    for key in groups:
        by_dev.setdefault(key[2], []).append(key)
    done = {}
    for dev, keys in by_dev.items():
        if dev.type == "cuda" and dev not in gens:
            gens[dev] = torch.Generator(device=dev).manual_seed(seed)
        res = distill_model([([ft.weight.data for _, _, ft in groups[k]], [fb.weight.data for _, fb, _ in groups[k]])
                             for k in keys], rank, clamp_quantile, gens.get(dev), **svd_kw)
        done.update(zip(keys, res))
    for (kind, shape, dev, dtype), sites in groups.items():
        ups, downs = done[(kind, shape, dev, dtype)]
        for (lor_base, fb, _), up, down in zip(sites, ups, downs):
            if isinstance(lor_base, LoraInjectedConv2d):
                up = up.reshape(up.shape[0], up.shape[1], 1, 1)
                down = down.reshape(down.shape[0], fb.in_channels, fb.kernel_size[0], fb.kernel_size[1])
            assert lor_base.lora_up.weight.shape == up.shape
            assert lor_base.lora_down.weight.shape == down.shape
            lor_base.lora_up.weight.data = up.to(device=dev, dtype=dtype)
            lor_base.lora_down.weight.data = down.to(device=dev, dtype=dtype)


def _load_pipe(path: str, device: str):
    """(unet, text_encoder) of a checkpoint; ``standin[:seed]`` builds random-init stand-ins (no diffusers here)."""
    if path.startswith("standin"):
        from .standin import clip_text_model, tiny_unet

        seed = int(path.split(":", 1)[1]) if ":" in path else 0
        torch.manual_seed(seed)
        return tiny_unet().to(device), clip_text_model(hidden=32, layers=2, heads=2).to(device)
    from diffusers import StableDiffusionPipeline

    pipe = StableDiffusionPipeline.from_pretrained(path, torch_dtype=torch.float16).to(device)
    return pipe.unet, pipe.text_encoder


def svd_distill(target_model: str, base_model: str, rank: int = 4, clamp_quantile: float = 0.99,
                device: str = "cuda:0", save_path: str = "svd_distill.safetensors"):
    """ref :95-142."""
    unet_b, te_b = _load_pipe(base_model, device)
    unet_t, te_t = _load_pipe(target_model, device)
    inject_trainable_lora_extended(unet_b, r=rank)
    inject_trainable_lora_extended(unet_t, r=rank)
    overwrite_base(unet_b, unet_t, rank=rank, clamp_quantile=clamp_quantile)
    inject_trainable_lora(te_b, r=rank, target_replace_module={"CLIPAttention"})
    inject_trainable_lora(te_t, r=rank, target_replace_module={"CLIPAttention"})
    overwrite_base(te_b, te_t, rank=rank, clamp_quantile=clamp_quantile)
    save_all(unet=unet_b, text_encoder=te_b, placeholder_token_ids=None, placeholder_tokens=None, save_path=save_path,
             save_lora=True, save_ti=False)


def main():
    from .cli_lora_pti import _parse_cli

    svd_distill(**_parse_cli(sys.argv[1:]))


if __name__ == "__main__":
    main()
