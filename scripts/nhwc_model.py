#!/usr/bin/env python
"""Lane-level numpy model of csrc/conv_nhwc.hip (index arithmetic only, f64 values).

Every function below follows its kernel statement by statement — same fragment layouts, same slot -> (tap, rank)
mapping, same clamps and masks — with the 64 lanes of a wave as a numpy axis and v_mfma_f32_16x16x32 modelled as
    D[lane][e] = sum_k A(row = (lane >> 4)*4 + e, k) * B(col = lane & 15, k),
    A(i, k) = a[lane with (l & 15) == i, (l >> 4) == k // 8][k % 8],   B likewise
(the layout gemm_ws.hip is verified with on hardware).  `python scripts/nhwc_model.py` checks the three contractions
against torch's conv2d and its autograd on CPU, for square / ragged / tiny maps; tests/test_conv_nhwc_model.py runs
the same at one small size.  It is a design check (a wrong shift sign or slot mapping shows up here, without a GPU),
not a product path.
"""
from __future__ import annotations

import numpy as np

LANES = np.arange(64)
L15, LG = LANES & 15, LANES >> 4


def mfma(a, b, acc):
    """a, b: [64, 8]; acc: [64, 4]."""
    A = np.zeros((16, 32))
    Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = a[l]
        Bm[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = b[l]
    D = A @ Bm.T  # [row, col]
    out = acc.copy()
    for l in range(64):
        for e in range(4):
            out[l, e] += D[(l >> 4) * 4 + e, l & 15]
    return out


def clamp(v, hi):
    return np.minimum(np.maximum(v, 0), hi)


def pack(down, r, C, lo_model=False):
    """conv3_pack_kernel.  down [r, C, 3, 3].  Returns pf [9, KC, 64, 8], pd [C/16, KS, 64, 8]."""
    KC, KS = C // 32, (9 * r + 31) // 32
    d = down.reshape(r, C, 9)
    pf = np.zeros((9, KC, 64, 8))
    for tap in range(9):
        for kc in range(KC):
            for l in range(64):
                row, c0 = l & 15, kc * 32 + (l >> 4) * 8
                lo = r <= 8 and row >= 8
                j = row - 8 if lo else row
                for e in range(8):
                    v = d[j, c0 + e, tap] if j < r else 0.0
                    # the model keeps exact values: the "low part" of an exactly representable value is 0
                    pf[tap, kc, l, e] = 0.0 if lo else v
    pd = np.zeros((C // 16, KS, 64, 8))
    for ct in range(C // 16):
        for ks in range(KS):
            for l in range(64):
                ch, s0 = ct * 16 + (l & 15), ks * 32 + (l >> 4) * 8
                for e in range(8):
                    s = s0 + e
                    tap, j = s // r, s % r
                    pd[ct, ks, l, e] = d[j, ch, tap] if s < 9 * r else 0.0
    return pf, pd


def down_fwd(x, pf, B, H, W, C, r, PT, ksplit=1):
    """conv3_down_nhwc_kernel (+ the sum over the `ksplit` channel shares).  x [B, H, W, C] -> T [B*H*W, r]."""
    KC = C // 32
    ntc, nrg = (W + 15) // 16, (H + PT - 1) // PT
    xf = x.reshape(-1)
    Tp = np.full((ksplit, B * H * W, r), np.nan)
    for bid, kz in ((bb, kk) for bb in range(B * nrg * ntc) for kk in range(ksplit)):
        T = Tp[kz]
        tc = bid % ntc
        rgi = (bid // ntc) % nrg
        b = bid // ntc // nrg
        y0, xx = rgi * PT, tc * 16 + L15
        red = np.zeros((4, PT, 64, 4))
        for wave in range(4):
            rowoff = [((b * H + int(clamp(y0 + q - 1, H - 1))) * W) * C for q in range(PT + 2)]
            rowmask = [0 <= y0 + q - 1 < H for q in range(PT + 2)]
            coloff = [clamp(xx + d - 1, W - 1) * C + LG * 8 for d in range(3)]
            colmask = [(xx + d - 1 >= 0) & (xx + d - 1 < W) for d in range(3)]
            acc = [np.zeros((64, 4)) for _ in range(PT)]
            for kc in range(kz * 4 + wave, KC, 4 * ksplit):
                for d in range(3):
                    pa = [pf[dy * 3 + d, kc] for dy in range(3)]
                    xr = []
                    for q in range(PT + 2):
                        base = kc * 32 + rowoff[q] + coloff[d]
                        v = xf[base[:, None] + np.arange(8)[None, :]]
                        m = (colmask[d] & rowmask[q])[:, None]
                        xr.append(np.where(m, v, 0.0))
                    for t in range(PT):
                        for dy in range(3):
                            acc[t] = mfma(pa[dy], xr[t + dy], acc[t])
            for t in range(PT):
                red[wave, t] = acc[t]
        for wave in range(4):
            for t in range(wave, PT, 4):
                v = red[:, t].sum(0)
                if r <= 8:
                    add = np.zeros_like(v)
                    add[:32] = red[:, t, 32:].sum(0)
                    v = v + np.where((LG < 2)[:, None], add, 0.0)
                yy = y0 + t
                for l in range(64):
                    if yy < H and xx[l] < W and LG[l] * 4 < r:
                        p = (b * H + yy) * W + xx[l]
                        for e in range(4):
                            if LG[l] * 4 + e < r:
                                T[p, LG[l] * 4 + e] = v[l, e]
    return Tp.sum(0)


def bwd_dx(dx, gt, pd, B, H, W, C, r, PT, csplit=1):
    """conv3_dx_nhwc_kernel.  dx [B, H, W, C] (modified in place), gt [B*H*W, r]."""
    KS = (9 * r + 31) // 32
    ntc, nrg = (W + 15) // 16, (H + PT - 1) // PT
    gf = gt.reshape(-1)
    dxf = dx.reshape(-1)
    for bid in range(B * nrg * ntc):
        tc = bid % ntc
        rgi = (bid // ntc) % nrg
        b = bid // ntc // nrg
        y0, xx = rgi * PT, tc * 16 + L15
        bg = [[None] * KS for _ in range(PT)]
        for ks in range(KS):
            halves = [[None, None] for _ in range(PT)]
            for h in range(2):
                s = ks * 32 + LG * 8 + 4 * h
                live = s < 9 * r
                tap = np.where(live, s // r, 0)
                j = np.where(live, s - tap * r, 0)
                dy, dxx = tap // 3 - 1, tap % 3 - 1
                xs = xx - dxx
                xok = live & (xs >= 0) & (xs < W)
                xc = clamp(xs, W - 1)
                for t in range(PT):
                    ys = y0 + t - dy
                    src = ((b * H + clamp(ys, H - 1)) * W + xc) * r + j
                    v = gf[src[:, None] + np.arange(4)[None, :]]
                    m = (xok & (ys >= 0) & (ys < H))[:, None]
                    halves[t][h] = np.where(m, v, 0.0)
            for t in range(PT):
                bg[t][ks] = np.concatenate(halves[t], axis=1)
        for wave, cz in ((w, z) for w in range(4) for z in range(csplit)):
            for cb in range(cz, C // 64, csplit):
                if True:
                    ct = cb * 4 + wave
                    for t in range(PT):
                        yy = y0 + t
                        a = np.zeros((64, 4))
                        for ks in range(KS):
                            a = mfma(pd[ct, ks], bg[t][ks], a)
                        for l in range(64):
                            if yy < H and xx[l] < W:
                                off = ((b * H + yy) * W + xx[l]) * C + LG[l] * 4 + ct * 16
                                dxf[off:off + 4] += a[l]
    return dx


def pick_pr(H, W):
    """nh_pick_pr: image rows per dDown strip."""
    pr = 0
    for p in range(1, H + 1):
        if (p + 2) * W * 8 <= 12 * 256 and (p + 2) * (W + 2) <= 432 and ((p * (W + 2) + 31) // 32) * 32 <= 456:
            pr = p
    return pr


def bwd_down(x, gt, B, H, W, C, r, nsplit, PR=None):
    """conv3_ddown_nhwc_kernel.  Returns part [nsplit, rank_pad, C*9]."""
    PR = PR or pick_pr(H, W)
    rank_pad = 4 if r <= 4 else 8 if r <= 8 else 16
    WP = W + 2
    spi = (H + PR - 1) // PR
    nstrips = B * spi
    SPP = PR * WP
    KSP = (SPP + 31) // 32
    nsplit = min(nsplit, nstrips)
    part = np.full((nsplit, rank_pad, C * 9), np.nan)
    for cc in range(C // 64):
        for sid in range(nsplit):
            acc = [[np.zeros((64, 4)) for _ in range(9)] for _ in range(4)]
            xs = np.zeros((432 + 41, 64))  # zeroed once: margins and tail are never written
            gT = np.zeros((16, 456))
            per, extra = divmod(nstrips, nsplit)
            s_begin = sid * per + min(sid, extra)
            for s in range(s_begin, s_begin + per + (1 if sid < extra else 0)):
                b = s // spi
                y0 = (s - b * spi) * PR
                rows_valid = min(PR, H - y0)
                for q in range((PR + 2) * W * 8):
                    f, c16 = q >> 3, q & 7
                    ryp = f // W
                    xq = f - ryp * W
                    y = y0 + ryp - 1
                    val = x[b, y, xq, cc * 64 + c16 * 8: cc * 64 + c16 * 8 + 8] if 0 <= y < H else 0.0
                    xs[ryp * WP + xq + 2, c16 * 8: c16 * 8 + 8] = val
                for sp in range(KSP * 32):
                    ry = sp // WP
                    xx = sp - ry * WP
                    ok = sp < SPP and 1 <= xx <= W and ry < rows_valid
                    for j in range(r):
                        gT[j, sp] = gt[(b * H + y0) * W + ry * W + xx - 1, j] if ok else 0.0
                for wave in range(4):
                    col = wave * 16 + L15
                    for ks in range(KSP):
                        sp0 = ks * 32 + LG * 8
                        ga = np.stack([gT[L15[l], sp0[l]:sp0[l] + 8] for l in range(64)])
                        for d in range(3):
                            v = np.stack([xs[sp0 + WP * d + i, col] for i in range(10)], axis=1)  # [64, 10]
                            acc[wave][d * 3 + 0] = mfma(ga, v[:, 0:8], acc[wave][d * 3 + 0])
                            acc[wave][d * 3 + 1] = mfma(ga, v[:, 1:9], acc[wave][d * 3 + 1])
                            acc[wave][d * 3 + 2] = mfma(ga, v[:, 2:10], acc[wave][d * 3 + 2])
            for wave in range(4):
                for t in range(9):
                    for l in range(64):
                        c = cc * 64 + wave * 16 + L15[l]
                        for e in range(4):
                            j = LG[l] * 4 + e
                            if j < r:
                                part[sid, j, c * 9 + t] = acc[wave][t][l, e]
    return part


def check(B, H, W, C, r, PT, nsplit, seed=0, ksplit=1, csplit=1):
    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    down = torch.randn(r, C, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    gt = torch.randn(B, r, H, W, generator=g, dtype=torch.float64)
    t = F.conv2d(x, down, padding=1)
    t.backward(gt)
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous().numpy()
    gt_rows = gt.permute(0, 2, 3, 1).reshape(-1, r).contiguous().numpy()
    pf, pd = pack(down.detach().numpy(), r, C)
    T = down_fwd(x_nhwc, pf, B, H, W, C, r, PT, ksplit)
    e_t = np.abs(T - t.detach().permute(0, 2, 3, 1).reshape(-1, r).numpy()).max()
    dx = np.zeros((B, H, W, C))
    bwd_dx(dx, gt_rows, pd, B, H, W, C, r, min(PT, 2), csplit)
    e_dx = np.abs(dx - x.grad.permute(0, 2, 3, 1).numpy()).max()
    part = bwd_down(x_nhwc, gt_rows, B, H, W, C, r, nsplit)
    dd = part[:, :r].sum(0).reshape(r, C, 3, 3)
    e_dd = np.abs(dd - down.grad.numpy()).max()
    return e_t, e_dx, e_dd


if __name__ == "__main__":
    for cfg in [(1, 5, 20, 64, 4, 2, 1), (2, 4, 7, 64, 8, 1, 2), (1, 3, 3, 64, 16, 4, 1), (1, 6, 16, 128, 12, 2, 3)]:
        errs = check(*cfg, ksplit=1 if cfg[3] == 64 else 2, csplit=1 if cfg[3] == 64 else 2)
        print(cfg, ["%.2e" % e for e in errs])
        assert max(errs) < 1e-9, errs
    print("ok")
