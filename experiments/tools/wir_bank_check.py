#!/usr/bin/env python3
"""Brute-force LDS bank-conflict check of the activation ring image of csrc/sed_conv_wir.hip (host only, no GPU).
For every supported (W, CIN) it enumerates the ds_read_b128 fragment reads of all nine taps and both k-halves and
counts, per 16-lane service group (MI355X_MICROARCH.md, LDS table), how many lanes share a 16-byte bank slot."""
import itertools

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def z(W, slots, col_lds, v):
    if slots == 16:
        return (col_lds + (8 * (v & 1) if W == 8 else 0)) & 15
    return ((col_lds >> 1) + (4 * (v & 1) if W == 8 else 0)) & 7


def check(W, CIN, R):
    slots, pix, wp = CIN // 8, CIN * 2, W + 2
    rowb = wp * pix
    worst = 1
    for v0, ti, tj, q, kh in itertools.product(range(4), range(3), range(3), range(CIN // 32), range(2)):
        addr = []
        for lane in range(64):
            n, hh = lane & 31, lane >> 5
            prow, pcol = n // W, n % W
            vin = v0 + prow + ti - 1
            col = pcol + tj
            slot = kh * (slots // 2) + 2 * q + hh
            addr.append((vin % R) * rowb + col * pix + ((slot ^ z(W, slots, col, vin)) << 4))
        for g in GROUPS:
            banks = {}
            for l in g:
                banks.setdefault((addr[l] // 16) % 16, set()).add(addr[l])
            worst = max(worst, max(len(s) for s in banks.values()))
    return worst


if __name__ == "__main__":
    for W, CIN, R in ((16, 128, 8), (16, 128, 16), (8, 128, 16), (32, 64, 8), (16, 64, 8), (16, 64, 16), (8, 64, 16),
                      (32, 128, 8), (32, 32, 8), (16, 32, 8)):
        print(f"W={W:2d} CIN={CIN:3d} R={R:2d}: worst {check(W, CIN, R)}-way")
