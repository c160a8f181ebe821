#!/usr/bin/env python
"""Lane-level numpy model of csrc/rank16_mfma.hip's fragment index arithmetic (v_mfma_f32_16x16x32 operand layouts of
csrc/mfma16.hpp, ds_read_b64_tr_b16 as verified on hardware by tests/test_gpu_parity_r4.py::test_ds_read_tr16_b64_semantics):
rank_update16's interleaved transposed product and bwd_g16's two phases must reproduce plain matrix products."""
import numpy as np

L = np.arange(64)
ROW, KQ = L & 15, L >> 4


def mfma(a, b, c=None):
    """a[lane][e]: A[row = lane & 15][k = 8 (lane >> 4) + e]; b[lane][e]: B[k][col = lane & 15]; returns d[lane][reg] =
    D[row = 4 (lane >> 4) + reg][col = lane & 15]."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        for e in range(8):
            A[l & 15, 8 * (l >> 4) + e] = a[l][e]
            B[8 * (l >> 4) + e, l & 15] = b[l][e]
    D = A @ B
    d = np.zeros((64, 4))
    for l in range(64):
        for reg in range(4):
            d[l][reg] = D[4 * (l >> 4) + reg, l & 15]
    return d if c is None else c + d


def rank_update16(T, U):
    """one 16-row slab x one 32-column group: out[m, n] = sum_j T[m, j] U[n, j]"""
    r = T.shape[1]
    out = np.zeros((16, 32))
    th = np.zeros((64, 8))
    for l in range(64):
        mm, q = l & 15, l >> 4
        for e in range(8):
            j = 8 * (q & 1) + e
            th[l][e] = T[mm, j] if j < r else 0.0
    d = []
    for tile in range(2):
        ua = np.zeros((64, 8))
        for l in range(64):
            mm, q = l & 15, l >> 4
            n = 8 * (mm >> 2) + 4 * tile + (mm & 3)
            for e in range(8):
                j = 8 * (q & 1) + e
                ua[l][e] = (U[n, j] if j < r else 0.0) if q < 2 else 0.0   # hi part only (lo = 0 in exact arithmetic)
        d.append(mfma(ua, th))
    for l in range(64):
        mm, q = l & 15, l >> 4
        for reg in range(4):
            out[mm, 8 * q + reg] = d[0][l][reg]
            out[mm, 8 * q + 4 + reg] = d[1][l][reg]
    return out


def tr_read(stage, pitch, q, i, row_off, c0):
    """lane (q, i): supplies the address of row 4 q + i / 4, columns c0 + 4 (i % 4); receives rows 4 q .. 4 q + 3 of column
    c0 + i of the 16-row block starting at row_off"""
    return [stage[row_off + 4 * q + e, c0 + i] for e in range(4)]


def bwd_g16_unit(G, T, U):
    """one 32-row x 32-column unit: Gt partial [32, 16] = G U, dUp partial [32 cols, 16] = G^T T"""
    r = T.shape[1]
    # phase 1
    uh = np.zeros((64, 8))
    for l in range(64):
        jj, kq = l & 15, l >> 4
        for e in range(8):
            uh[l][e] = U[8 * kq + e, jj] if jj < r else 0.0
    gt = np.zeros((32, 16))
    for rg in range(2):
        cur = np.zeros((64, 8))
        for l in range(64):
            jj, q = l & 15, l >> 4
            cur[l] = G[rg * 16 + jj, 8 * q:8 * q + 8]
        d1 = mfma(cur, uh)
        for l in range(64):
            jj, q = l & 15, l >> 4
            for reg in range(4):
                gt[rg * 16 + 4 * q + reg, jj] = d1[l][reg]
    # phase 2: the stage tile is the unit itself (row jj / 16 + jj, columns 8 q ..)
    stage = G.copy()
    tf = np.zeros((64, 8))
    for l in range(64):
        jj, q = l & 15, l >> 4
        for e in range(8):
            row = 4 * q + e if e < 4 else 16 + 4 * q + (e - 4)
            tf[l][e] = T[row, jj] if jj < r else 0.0
    dup = np.zeros((32, 16))
    for nt in range(2):
        a = np.zeros((64, 8))
        for l in range(64):
            jj, q = l & 15, l >> 4
            a[l][:4] = tr_read(stage, None, q, jj, 0, 16 * nt)
            a[l][4:] = tr_read(stage, None, q, jj, 16, 16 * nt)
        d2 = mfma(a, tf)
        for l in range(64):
            jj, q = l & 15, l >> 4
            for reg in range(4):
                dup[16 * nt + 4 * q + reg, jj] = d2[l][reg]
    return gt, dup


def rowdot16_step(X, F):
    """one 16-row slab x one 32-column k-step: T[m, j] = sum_c X[m, c] F[j, c]"""
    r = F.shape[0]
    fh, cur = np.zeros((64, 8)), np.zeros((64, 8))
    for l in range(64):
        jj, kq = l & 15, l >> 4
        cur[l] = X[jj, 8 * kq:8 * kq + 8]
        if jj < r:
            fh[l] = F[jj, 8 * kq:8 * kq + 8]
    d = mfma(cur, fh)
    t = np.zeros((16, 16))
    for l in range(64):
        jj, q = l & 15, l >> 4
        for reg in range(4):
            t[4 * q + reg, jj] = d[l][reg]
    return t


def check():
    rng = np.random.default_rng(0)
    for r in (16, 12, 9):
        T, U = rng.standard_normal((16, r)), rng.standard_normal((32, r))
        assert np.allclose(rank_update16(T, U), T @ U.T)
        G, T32, U32 = rng.standard_normal((32, 32)), rng.standard_normal((32, r)), rng.standard_normal((32, r))
        gt, dup = bwd_g16_unit(G, T32, U32)
        assert np.allclose(gt[:, :r], G @ U32) and not gt[:, r:].any()
        assert np.allclose(dup[:, :r], G.T @ T32) and not dup[:, r:].any()
        X, F = rng.standard_normal((16, 32)), rng.standard_normal((r, 32))
        t = rowdot16_step(X, F)
        assert np.allclose(t[:, :r], X @ F.T) and not t[:, r:].any()
    return True


if __name__ == "__main__":
    print("ok" if check() else "mismatch")
