"""Thread-level emulation (numpy) of the gn_nhwc_* kernels' index math: same geometry, same (block, thread) -> address
mapping, same partial layouts.  Validates the algorithm and indexing against torch.group_norm on CPU."""
import numpy as np, torch, torch.nn.functional as F

kHT, kNU = 256, 4

def geo(B, C, HW, G):
    c8 = C // 8
    best, best_act = 1, 0
    for d in range(1, min(c8, 128) + 1):
        if c8 % d: continue
        act = (kHT // d) * d
        if act >= best_act: best_act, best = act, d
    cw = best; tiles = c8 // cw; nslots = kHT // cw
    px = max(2 * nslots, HW * B * tiles // 768); px = min(px, HW)
    S = (HW + px - 1) // px
    return dict(c8=c8, cw=cw, tiles=tiles, nslots=nslots, px=px, S=S)

def block(q, bid, tid, C, HW):
    bs, tile = divmod(bid, q['tiles'])
    b, s = divmod(bs, q['S'])
    slot, cl = divmod(tid, q['cw'])
    active = slot < q['nslots']
    col = tile * q['cw'] + cl
    p0 = s * q['px']; np_ = min(q['px'], HW - p0)
    base = (b * HW + p0) * C + col * 8
    return b, s, slot, active, col, np_, base

def run(B, C, H, W, G, act, addend=None, seed=0):
    rng = np.random.default_rng(seed)
    HW = H * W
    x = (rng.standard_normal((B, HW, C)) * 1.5 + rng.standard_normal((1, 1, C)) * 3).astype(np.float32)  # NHWC memory
    gamma = (rng.standard_normal(C) * 0.5 + 1).astype(np.float32); beta = (rng.standard_normal(C) * 0.3).astype(np.float32)
    go = rng.standard_normal((B, HW, C)).astype(np.float32)
    xf, gof = x.reshape(-1), go.reshape(-1)
    q = geo(B, C, HW, G)
    nblocks = B * q['S'] * q['tiles']
    part = np.zeros(B * q['S'] * 2 * C, np.float32)
    # ---- stats
    for bid in range(nblocks):
        s_red = np.zeros((2, kHT, 8), np.float32)
        regs = {}
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            s1 = np.zeros(8, np.float32); s2 = np.zeros(8, np.float32); sh = None
            if active:
                sh = xf[base:base + 8].copy()
                p = slot
                while p < np_:
                    for u in range(kNU):
                        pp = p + u * q['nslots']
                        if pp < np_:
                            v = xf[base + pp * C: base + pp * C + 8]
                            d = v - sh; s1 += d; s2 += d * d
                    p += q['nslots'] * kNU
                s_red[0, tid] = s1; s_red[1, tid] = s2
            regs[tid] = (s1, s2, sh)
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            if active and slot == 0:
                s1, s2, sh = regs[tid]
                s1 = s1.copy(); s2 = s2.copy()
                for sl in range(1, q['nslots']):
                    s1 += s_red[0, sl * q['cw'] + tid]; s2 += s_red[1, sl * q['cw'] + tid]
                mean = sh + s1 / np_; m2 = np.maximum(s2 - s1 * s1 / np_, 0)
                o = ((b * q['S'] + s) * 2) * C + col * 8
                part[o:o + 8] = mean; part[o + C:o + C + 8] = m2
    # ---- finalize
    aff = np.zeros((B, 4, C), np.float32); cpg = C // G
    for bg in range(B * G):
        b, g = divmod(bg, G)
        a = 0.0
        for i in range(q['S'] * cpg):
            s, cc = divmod(i, cpg); c = g * cpg + cc
            add = 0.0 if addend is None else addend[b, c]
            a += min(q['px'], HW - s * q['px']) * (part[((b * q['S'] + s) * 2) * C + c] + add)
        n_all = HW * cpg; mean = a / n_all; m2 = 0.0
        for i in range(q['S'] * cpg):
            s, cc = divmod(i, cpg); c = g * cpg + cc
            add = 0.0 if addend is None else addend[b, c]
            o = ((b * q['S'] + s) * 2) * C + c
            dm = part[o] + add - mean
            m2 += part[o + C] + min(q['px'], HW - s * q['px']) * dm * dm
        rstd = 1 / np.sqrt(m2 / n_all + 1e-5)
        for cc in range(cpg):
            c = g * cpg + cc; add = 0.0 if addend is None else addend[b, c]
            ga = gamma[c] * rstd
            aff[b, 0, c] = ga; aff[b, 1, c] = beta[c] + (add - mean) * ga; aff[b, 2, c] = mean - add; aff[b, 3, c] = rstd
    # ---- apply
    y = np.zeros_like(xf)
    for bid in range(nblocks):
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            if not active: continue
            ga = aff[b, 0, col * 8:col * 8 + 8]; be = aff[b, 1, col * 8:col * 8 + 8]
            p = slot
            while p < np_:
                for u in range(kNU):
                    pp = p + u * q['nslots']
                    if pp < np_:
                        z = xf[base + pp * C: base + pp * C + 8] * ga + be
                        y[base + pp * C: base + pp * C + 8] = z / (1 + np.exp(-z)) if act else z
                p += q['nslots'] * kNU
    # ---- backward stats / finalize / apply
    def terms(v, g_, b, col):
        ga = aff[b, 0, col*8:col*8+8]; be = aff[b, 1, col*8:col*8+8]; mean = aff[b, 2, col*8:col*8+8]; rstd = aff[b, 3, col*8:col*8+8]
        xh = (v - mean) * rstd
        dz = g_.copy()
        if act:
            z = v * ga + be; sg = 1 / (1 + np.exp(-z)); dz = dz * (sg * (1 + z * (1 - sg)))
        return gamma[col*8:col*8+8] * dz, xh, rstd
    partb = np.zeros_like(part)
    for bid in range(nblocks):
        s_red = np.zeros((2, kHT, 8), np.float32); regs = {}
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            s1 = np.zeros(8, np.float32); s2 = np.zeros(8, np.float32)
            if active:
                p = slot
                while p < np_:
                    for u in range(kNU):
                        pp = p + u * q['nslots']
                        if pp < np_:
                            t, xh, _ = terms(xf[base + pp*C: base + pp*C + 8], gof[base + pp*C: base + pp*C + 8], b, col)
                            s1 += t; s2 += t * xh
                    p += q['nslots'] * kNU
                s_red[0, tid] = s1; s_red[1, tid] = s2
            regs[tid] = (s1, s2)
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            if active and slot == 0:
                s1, s2 = regs[tid]; s1 = s1.copy(); s2 = s2.copy()
                for sl in range(1, q['nslots']):
                    s1 += s_red[0, sl * q['cw'] + tid]; s2 += s_red[1, sl * q['cw'] + tid]
                o = ((b * q['S'] + s) * 2) * C + col * 8
                partb[o:o+8] = s1; partb[o+C:o+C+8] = s2
    cvec = np.zeros((B, 2, C), np.float32)
    for bg in range(B * G):
        b, g = divmod(bg, G); s1 = s2 = 0.0
        for i in range(q['S'] * cpg):
            s, cc = divmod(i, cpg); c = g * cpg + cc
            o = ((b * q['S'] + s) * 2) * C + c
            s1 += partb[o]; s2 += partb[o + C]
        cvec[b, 0, g*cpg:(g+1)*cpg] = s1 / (HW * cpg); cvec[b, 1, g*cpg:(g+1)*cpg] = s2 / (HW * cpg)
    dx = np.zeros_like(xf)
    for bid in range(nblocks):
        for tid in range(kHT):
            b, s, slot, active, col, np_, base = block(q, bid, tid, C, HW)
            if not active: continue
            p = slot
            while p < np_:
                for u in range(kNU):
                    pp = p + u * q['nslots']
                    if pp < np_:
                        t, xh, rstd = terms(xf[base + pp*C: base + pp*C + 8], gof[base + pp*C: base + pp*C + 8], b, col)
                        dx[base + pp*C: base + pp*C + 8] = rstd * (t - cvec[b, 0, col*8:col*8+8] - xh * cvec[b, 1, col*8:col*8+8])
                p += q['nslots'] * kNU
    # ---- reference
    xt = torch.from_numpy(x.reshape(B, H, W, C)).permute(0, 3, 1, 2).double().requires_grad_(True)
    xin = xt if addend is None else xt + torch.from_numpy(addend).double()[:, :, None, None]
    yr = F.group_norm(xin, G, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5)
    if act: yr = F.silu(yr)
    yr.backward(torch.from_numpy(go.reshape(B, H, W, C)).permute(0, 3, 1, 2).double())
    yr_n = yr.detach().permute(0, 2, 3, 1).reshape(-1).numpy(); dxr = xt.grad.permute(0, 2, 3, 1).reshape(-1).numpy()
    e1 = np.abs(y - yr_n).max(); e2 = np.abs(dx - dxr).max() / (np.abs(dxr).max() + 1e-12)
    print((B, C, H, W, G, act, addend is not None), q, "fwd err %.2e  bwd rel err %.2e" % (e1, e2))
    assert e1 < 2e-4 and e2 < 2e-4

run(2, 64, 4, 6, 8, True)
run(1, 88, 5, 7, 11, False)
run(2, 96, 3, 5, 3, True)
run(1, 320, 8, 8, 32, True)
rng = np.random.default_rng(5)
run(2, 64, 4, 6, 8, True, addend=rng.standard_normal((2, 64)).astype(np.float32) * 2)
print("emulation ok")
run(1, 64, 32, 32, 8, True)
run(2, 16, 10, 9, 2, False, addend=rng.standard_normal((2, 16)).astype(np.float32))
print("multi-slice ok")
