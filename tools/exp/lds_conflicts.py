#!/usr/bin/env python3
"""LDS bank-conflict model for the 16-byte exchanges of the transform kernels (MI355X_MICROARCH.md, LDS table):
ds_read_b128 is served in four groups of 16 lanes, ds_write_b128 in eight groups of 8 contiguous lanes; within a group
every extra distinct 16-byte slot on a busy bank quad (slot mod 16) costs one more LDS cycle.  Prints, per pass of a
length-L transform with V values per thread, the cycles per wave-instruction of the get (read) and put (write) side for
candidate slot functions.   python tools/exp/lds_conflicts.py [L] [V] [C columns interleaved]"""
import sys

RD = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
RD = RD + [[x + 32 for x in g] for g in RD]
WR = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def plan(lgL, lgV):
    rem, n = (lgL - lgV) % lgV, (lgL - lgV) // lgV
    lgs = [lgV] + ([rem] if rem else []) + [lgV] * n
    out, left = [], lgL
    for lg in lgs:
        left -= lg
        out.append((lg, left))
    return out


def reg_pos(pl, tpf, i, b, idx):
    lg, lgS = pl[i]
    bb, q = b + tpf * (idx >> lg), idx & ((1 << lg) - 1)
    j, block = bb & ((1 << lgS) - 1), bb >> lgS
    return (block << (lgS + lg)) + j + (q << lgS)


def cycles(slots, groups):
    tot = 0
    for g in groups:
        quad = {}
        for lane in g:
            quad.setdefault(slots[lane] % 16, set()).add(slots[lane])
        tot += max(len(v) for v in quad.values())
    return tot


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lgL, lgV = L.bit_length() - 1, V.bit_length() - 1
    pl, tpf = plan(lgL, lgV), L // V
    cands = {
        "pos": lambda p: p,
        "pos + pos>>4": lambda p: p + (p >> 4),
        "pos + pos>>3": lambda p: p + (p >> 3),
        "pos + pos>>5": lambda p: p + (p >> 5),
        "pos + pos>>6": lambda p: p + (p >> 6),
        "pos ^ (pos>>4 & 3)": lambda p: p ^ ((p >> 4) & 3),
        "pos + (pos>>4) + (pos>>8)": lambda p: p + (p >> 4) + (p >> 8),
        "pos + (pos>>3) + (pos>>6)": lambda p: p + (p >> 3) + (p >> 6),
        "pos + (pos>>4)*1 ^ ": lambda p: (p + (p >> 4)) ^ ((p >> 8) & 3),
    }
    print("L", L, "V", V, "C", C, "passes (lg r, lg stride):", pl, "threads per transform", tpf)
    for name, f in cands.items():
        stride = f(L - 1) + 1
        if C > 1:                      # C transforms side by side: lane l -> column l % C, butterfly l // C
            want = 16 // C if C <= 16 else 1
            stride += (want - stride) & 15
        row = []
        for i in range(len(pl)):
            rd = wr = n = 0
            for w in range(max(1, tpf * C // 64)):
                for idx in range(V):
                    slots = []
                    for lane in range(64):
                        t = w * 64 + lane
                        c, b = t % C, t // C
                        if b >= tpf:
                            b = tpf - 1
                        slots.append(c * stride + f(reg_pos(pl, tpf, i, b, idx)))
                    rd += cycles(slots, RD)
                    wr += cycles(slots, WR)
                    n += 1
            row.append("p%d rd %.2f wr %.2f" % (i, rd / n / 4, wr / n / 8))
        print("%-30s" % name, " | ".join(row))


if __name__ == "__main__":
    main()
