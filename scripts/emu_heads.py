"""Coverage check of the head-padded epilogue of linear_gemm_fwd_kernel (index math only): every physical 16-byte
chunk of every row of the padded output is written exactly once — logical chunks by the tile store loop, pad chunks by
the zero loop — and the operand remap of stage_slab reads exactly the logical chunks."""
import numpy as np

def head_chunk(c, hc, hp):
    return (c // hc) * hp + (c % hc) if hc else c

def check(N, d, D, BN, BM=64, M=130, kGT=256):
    yhc, yhp, CH = d // 8, D // 8, BN // 8
    assert CH % yhc == 0
    heads = N // d
    phys_chunks = heads * yhp
    hits = np.zeros((M, phys_chunks), int)
    val = np.full((M, phys_chunks), -1, int)   # logical chunk stored there (-2 = zero pad)
    for m0 in range(0, M, BM):
        for n0 in range(0, N, BN):
            for tid in range(kGT):
                idx = tid
                while idx < BM * CH:
                    rl, cc = divmod(idx, CH)
                    if m0 + rl < M and n0 + cc * 8 < N:
                        pc = head_chunk((n0 >> 3) + cc, yhc, yhp)
                        hits[m0 + rl, pc] += 1; val[m0 + rl, pc] = (n0 >> 3) + cc
                    idx += kGT
                head0 = (n0 >> 3) // yhc
                nheads = min(CH // yhc, (N >> 3) // yhc - head0); padc = yhp - yhc
                idx = tid
                while idx < BM * nheads * padc:
                    rl, rem = divmod(idx, nheads * padc); hh, pc = divmod(rem, padc)
                    if m0 + rl < M:
                        p = (head0 + hh) * yhp + yhc + pc
                        hits[m0 + rl, p] += 1; val[m0 + rl, p] = -2
                    idx += kGT
    assert (hits == 1).all(), (N, d, D, BN, np.argwhere(hits != 1)[:5])
    # logical chunk c must sit at head (c // yhc), offset c % yhc; pad chunks are the rest
    for p in range(phys_chunks):
        h, o = divmod(p, yhp)
        want = h * yhc + o if o < yhc else -2
        assert (val[:, p] == want).all(), (p, want, val[0, p])

for N, d, D in [(320, 40, 64), (320, 40, 48), (640, 80, 128), (640, 80, 96), (1280, 160, 192), (640, 40, 64), (2560, 160, 256)]:
    for BN in (160, 320):
        check(N, d, D, BN)
# operand remap: the chunks a K-step slab fetches (8 chunks per row, swizzled) are the logical chunks' physical homes
for K, d, D in [(320, 40, 64), (640, 80, 128), (1280, 160, 192)]:
    hc, hp = d // 8, D // 8
    for k0 in range(0, K, 64):
        for rl in range(16):
            got = sorted(head_chunk((k0 >> 3) + ((lane & 7) ^ (rl & 7)), hc, hp) for lane in range(8))
            want = sorted(head_chunk(c, hc, hp) for c in range(k0 >> 3, (k0 >> 3) + 8))
            assert got == want
print("head-layout index math ok")
