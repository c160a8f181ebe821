#!/usr/bin/env python
"""LDS bank-conflict check of csrc/factor_mfma.hip's three access patterns for a given row pitch, with the lane groups
and bank functions of /opt/skills/guides/MI355X_MICROARCH.md (LDS section): a wave64 access is serviced in fixed lane
groups, one LDS cycle per group when no two lanes of a group touch different addresses on one bank.
    python scripts/lds_banks.py [cols ...]     -> worst n-way conflict per pattern (1 = conflict-free)
"""
import sys

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(0, 32)), list(range(32, 64))]
OCTETS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def fm_pitch(cols):
    return ((cols * 2 + 31) // 64) * 64 + 32


def worst(groups, addr, nbytes, nbanks):
    w = 1
    for grp in groups:
        per_bank = {}
        for l in grp:
            for b in range(addr[l] // 4, (addr[l] + nbytes) // 4):
                per_bank.setdefault(b % nbanks, set()).add(b)
        w = max(w, max(len(v) for v in per_bank.values()))
    return w


def check(cols, pitch=None):
    pitch = pitch or fm_pitch(cols)
    out = {}
    # phase 1: ds_read_b128, lane -> row lane & 15, 16-byte slot lane >> 4 of the k-step
    out["phase1_b128"] = max(worst(B128_GROUPS, [(l & 15) * pitch + (l >> 4) * 16 + ks * 64 for l in range(64)], 16, 64)
                             for ks in range(cols // 32))
    # phase 2: ds_read_b64_tr_b16, lane (q, i) -> row 4q + i / 4, columns c0 + 4 (i % 4)
    out["phase2_tr_b64"] = max(worst(HALVES, [(4 * (l >> 4) + ((l & 15) >> 2)) * pitch + (c0 + 4 * (l & 3)) * 2
                                              for l in range(64)], 8, 64) for c0 in range(0, cols, 16))
    # staging: ds_write_b128, consecutive lanes = consecutive 16-byte pieces of a row
    c8 = cols // 8
    out["stage_w128"] = max(worst(OCTETS, [((p0 + l) // c8) * pitch + ((p0 + l) % c8) * 16 for l in range(64)], 16, 32)
                            for p0 in range(0, 4 * c8, 64))
    return pitch, out


if __name__ == "__main__":
    for cols in [int(a) for a in sys.argv[1:]] or [64, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1280]:
        pitch, res = check(cols)
        unpadded = check(cols, cols * 2)[1]
        print(f"cols {cols:5d} pitch {pitch:5d}: {res}   (unpadded pitch {cols * 2}: {unpadded})")
