#!/usr/bin/env python
"""Lane-level numpy model of csrc/factor_mfma.hip (index arithmetic only, f64 values, no hi / lo rounding).

Follows the kernel statement by statement — LDS images with the padded pitch, the packed factor fragments, the k-step
split of phase 1 over the four waves, the position permutation of T inside a 32-row k-step, the transpose-read address
pattern of phase 2, the chunk loop over the streamed operand, head-padded rows, tail rows — with the 64 lanes of a wave as
a Python loop and v_mfma_f32_16x16x32 modelled as in scripts/nhwc_model.py (the layout gemm_ws.hip is verified with on
hardware).  ``ds_read_b64_tr_b16`` is modelled as the microarchitecture guide states it: inside a 16-lane group, lane i
receives element i % 4 of the four 8-byte words addressed by lanes i / 4 + 4 e (e = 0..3).  `python scripts/fm_model.py`
checks both partial slabs against the dense formula for several geometries; tests/test_fm_model.py runs a small one.
A design check (a wrong pitch, position or chunk offset shows up here, without a GPU), not a product path.
"""
from __future__ import annotations

import numpy as np

from nhwc_model import mfma  # noqa: E402  (same directory)


def fm_pitch(cols):
    return ((cols * 2 + 31) // 64) * 64 + 32


def fm_tpitch(R):
    return R * 2 + 16


def hchunk(c, hc, hp):
    return (c // hc) * hp + (c % hc) if hc else c


def pack(factor_jc, r):
    """factor(jj, c) [r, C] -> pk[c8][16][8] (hi split only: the model keeps exact values)."""
    C = factor_jc.shape[1]
    pk = np.zeros((C // 8, 16, 8))
    for c8 in range(C // 8):
        for jj in range(r):
            pk[c8, jj] = factor_jc[jj, c8 * 8:c8 * 8 + 8]
    return pk.reshape(-1)  # element offsets as the kernel computes them


class Lds:
    """Byte-addressed LDS holding 2-byte elements (modelled as f64 per element slot) and 4-byte floats."""

    def __init__(self, nbytes):
        self.h = np.full(nbytes // 2, np.nan)   # 16-bit element view
        self.f = np.full(nbytes // 4, np.nan)   # f32 view (scratch)

    def w16(self, byte, vals):
        assert byte % 2 == 0
        self.h[byte // 2:byte // 2 + len(vals)] = vals

    def r16(self, byte, n):
        assert byte % 2 == 0
        v = self.h[byte // 2:byte // 2 + n]
        assert not np.isnan(v).any(), f"LDS read of unwritten bytes at {byte}"
        return v.copy()


def stage(lds, base, pitch, data, ld_rows, m0, nrows, R, c8w, c8_0, hc, hp):
    """fm_issue + fm_write for every thread and batch: the [R, c8w] tile of 16-byte pieces."""
    for p in range(R * c8w):
        row, c = divmod(p, c8w)
        ok = row < nrows
        if ok:
            src = hchunk(c8_0 + c, hc, hp) * 8
            v = data[m0 + row, src:src + 8]
        else:
            v = np.zeros(8)
        lds.w16(base + row * pitch + c * 16, v)


def phase1(lds, base, pitch, nrt, nks, pk, pk_off, acc):
    """acc[wave][t] += ...; k-steps dealt to the waves."""
    for wave in range(4):
        for ks in range(wave, nks, 4):
            b = np.zeros((64, 8))
            for lane in range(64):
                off = pk_off + ks * 512 + lane * 8
                b[lane] = pk[off:off + 8]
            for t in range(nrt):
                a = np.zeros((64, 8))
                for lane in range(64):
                    a[lane] = lds.r16(base + (lane & 15) * pitch + (lane >> 4) * 16 + t * 16 * pitch + ks * 64, 8)
                acc[wave][t] = mfma(a, b, acc[wave][t])


def combine(lds, acc, tt_base, nrt, R, scale):
    """fm_combine: scratch round trip left out (a register image copy); T -> [jj][position] (hi split)."""
    tp = fm_tpitch(R)
    for t in range(nrt):
        tot = sum(acc[w][t] for w in range(4)) * scale  # [64, 4]
        for lane in range(64):
            jj, q = lane & 15, lane >> 4
            pos = (t >> 1) * 32 + 8 * q + 4 * (t & 1)
            lds.w16(tt_base + jj * tp + pos * 2, tot[lane])


def tr_read(lds, addr_of_lane):
    """ds_read_b64_tr_b16: out[lane][e] = word of lane (16 * (lane // 16) + (lane % 16) // 4 + 4 e), element lane % 4."""
    words = [lds.r16(addr_of_lane[l], 4) for l in range(64)]
    out = np.zeros((64, 4))
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for e in range(4):
            out[lane, e] = words[16 * g + 4 * e + (i >> 2)][i & 3]
    return out


def colfrag(lds, base, pitch, ks, c0):
    addr0, addr1 = {}, {}
    for lane in range(64):
        q, i = lane >> 4, lane & 15
        p = base + (ks * 32 + 4 * q + (i >> 2)) * pitch + (c0 + 4 * (i & 3)) * 2
        addr0[lane], addr1[lane] = p, p + 16 * pitch
    return np.concatenate([tr_read(lds, addr0), tr_read(lds, addr1)], axis=1)  # [64, 8]


def load_tfrags(lds, tt_base, R, nk2):
    tp = fm_tpitch(R)
    tf = []
    for k2 in range(nk2):
        f = np.zeros((64, 8))
        for lane in range(64):
            jj, q = lane & 15, lane >> 4
            f[lane] = lds.r16(tt_base + jj * tp + (k2 * 32 + 8 * q) * 2, 8)
        tf.append(f)
    return tf


def phase2(lds, base, pitch, nk2, ncols, tf, out, col_off, RT):
    """out [RT, C]: columns col_off .. col_off + ncols."""
    for ct in range(ncols // 16):  # dealt to waves ct % 4: no interaction
        d = np.zeros((64, 4))
        for k2 in range(nk2):
            d = mfma(colfrag(lds, base, pitch, k2, ct * 16), tf[k2], d)
        for lane in range(64):
            jj, q = lane & 15, lane >> 4
            if jj < RT:
                out[jj, col_off + ct * 16 + 4 * q:col_off + ct * 16 + 4 * q + 4] = d[lane]


def run_block(g, x, down, up, scale, rb, geom, g_heads=None, x_heads=None):
    """One workgroup of factors_mfma_kernel.  g [M, N'], x [M, K'] (possibly head-padded physical rows), returns
    (up_part [RT, N], down_part [RT, K]) of row block rb."""
    R, cw_full, nch, pa, pb, ax = geom["R"], geom["cw"], geom["nchunk"], geom["pitch_a"], geom["pitch_b"], geom["ax"]
    r = down.shape[0]
    K, N = down.shape[1], up.shape[0]
    RT = 4 if r <= 4 else 8 if r <= 16 and r <= 8 else 16
    M = g.shape[0]
    m0 = rb * R
    nrows = min(R, M - m0)
    nrt, nk2 = R // 16, R // 32
    pk_down, pk_up = pack(down, r), pack(up.T.copy(), r)
    da, db = (x, g) if ax else (g, x)
    Ca, Cb = (K, N) if ax else (N, K)
    ha = x_heads if ax else g_heads
    hb = g_heads if ax else x_heads
    hca, hpa = (ha[1] // 8, ha[2] // 8) if ha else (0, 0)
    hcb, hpb = (hb[1] // 8, hb[2] // 8) if hb else (0, 0)
    pka, pkb = (pk_down, pk_up) if ax else (pk_up, pk_down)
    outa, outb = np.zeros((RT, Ca)), np.zeros((RT, Cb))
    bufA, bufB = 0, R * pa
    ttA = bufB + R * pb
    ttB = ttA + 32 * fm_tpitch(R)
    lds = Lds(ttB + 32 * fm_tpitch(R))
    assert geom["lds"] == ttB + 32 * fm_tpitch(R)
    stage(lds, bufA, pa, da, None, m0, nrows, R, Ca // 8, 0, hca, hpa)
    acc = [[np.zeros((64, 4)) for _ in range(nrt)] for _ in range(4)]
    phase1(lds, bufA, pa, nrt, Ca // 32, pka, 0, acc)
    combine(lds, acc, ttA, nrt, R, scale)
    tf = load_tfrags(lds, ttA, R, nk2)
    acc = [[np.zeros((64, 4)) for _ in range(nrt)] for _ in range(4)]
    for c in range(nch):
        col0 = c * cw_full
        cw = min(cw_full, Cb - col0)
        stage(lds, bufB, pb, db, None, m0, nrows, R, cw // 8, col0 // 8, hcb, hpb)
        phase1(lds, bufB, pb, nrt, cw // 32, pkb, (col0 // 8) * 128, acc)
        phase2(lds, bufB, pb, nk2, cw, tf, outb, col0, RT)
    combine(lds, acc, ttB, nrt, R, scale)
    tf = load_tfrags(lds, ttB, R, nk2)
    phase2(lds, bufA, pa, nk2, Ca, tf, outa, 0, RT)
    return (outb, outa) if ax else (outa, outb)


def geometry(M, K, N, R, lds_cap):
    """fm_fit of the host planner."""
    Ca, Cb = min(K, N), max(K, N)
    pa = fm_pitch(Ca)
    fixed = R * pa + 2 * 32 * fm_tpitch(R)
    cwmax = min(8 * 256 * 8 // R, 512)
    while cwmax >= 32:
        nch = -(-Cb // cwmax)
        cw = min((-(-Cb // nch) + 31) // 32 * 32, cwmax)
        pb = max(fm_pitch(cw), 288)
        if fixed + R * pb <= lds_cap:
            return dict(R=R, cw=cw, nchunk=-(-Cb // cw), pitch_a=pa, pitch_b=pb, ax=K <= N, lds=fixed + R * pb)
        cwmax -= 32
    return None


def heads_pack(a, lay, fill):
    h, d, D = lay
    out = np.full(a.shape[:-1] + (h * D,), fill, dtype=a.dtype)
    out.reshape(a.shape[:-1] + (h, D))[..., :d] = a.reshape(a.shape[:-1] + (h, d))
    return out


def check(M, K, N, r, R, gh=None, xh=None, cap=81920, seed=0):
    rng = np.random.default_rng(seed)
    g, x = rng.standard_normal((M, N)), rng.standard_normal((M, K))
    down, up = rng.standard_normal((r, K)), rng.standard_normal((N, r))
    s = 0.7
    geom = geometry(M, K, N, R, cap)
    assert geom is not None, "does not fit"
    gp = heads_pack(g, gh, 7.0) if gh else g
    xp = heads_pack(x, xh, -3.0) if xh else x
    nparts = -(-M // R)
    d_up, d_down = np.zeros((N, r)), np.zeros((r, K))
    for rb in range(nparts):
        up_part, down_part = run_block(gp, xp, down, up, s, rb, geom, gh, xh)
        d_up += up_part[:r].T
        d_down += down_part[:r]
    t, gt = s * x @ down.T, s * g @ up
    assert np.allclose(d_up, g.T @ t, rtol=1e-9, atol=1e-9), "dUp"
    assert np.allclose(d_down, gt.T @ x, rtol=1e-9, atol=1e-9), "dDown"
    return geom


if __name__ == "__main__":
    for args in [dict(M=70, K=64, N=64, r=4, R=64), dict(M=100, K=64, N=320, r=4, R=32),
                 dict(M=40, K=320, N=96, r=8, R=32), dict(M=64, K=320, N=320, r=16, R=64),
                 dict(M=33, K=320, N=640, r=4, R=64, gh=None, xh=(8, 40, 64)),
                 dict(M=50, K=320, N=320, r=4, R=32, gh=(8, 40, 64))]:
        geom = check(**args)
        print("ok", args, geom)
