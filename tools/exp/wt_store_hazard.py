#!/usr/bin/env python3
"""Store-data hazard check of the row kernels' write-through stores in a gfx950 assembly listing.

    python tools/exp/wt_store_hazard.py listing.s [kernel-name substring ...]

gfx940+ needs two wait states between a VMEM store of more than 64 bits and a VALU instruction that overwrites the store's data
registers (LLVM GCNHazardRecognizer: VMEM store data hazard).  The compiler keeps them for stores it emits itself; it does not
look into inline assembly.  Round 3 issued the write-through stores (sc0 sc1) by inline assembly: in the packed complex64 row
kernel the values went through ONE temporary register tuple that the next value's instructions overwrote immediately (12 of
16 stores) -- rel-L2 0.67 "at every size"; the double-precision kernel's values sat in tuples of their own (0 of 16).  Round 4
stores through __builtin_amdgcn_raw_buffer_store_b128 with the policy in the aux bits: the compiler sees a store.
"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    keys = sys.argv[2:] or ["k_rowI"]
    for m in re.finditer(r"\n(_Z\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end", txt, re.S):
        name, body = m.group(1), m.group(2).split("\n")
        if not any(k in name for k in keys):
            continue
        idx = [i for i, ln in enumerate(body) if re.search(r"(global|buffer)_store_dwordx4.*sc0 sc1", ln)]
        if not idx:
            continue
        haz = 0
        for i in idx:
            mm = re.search(r"_store_dwordx4 (?:v\[\d+:\d+\], )?v\[(\d+):(\d+)\]", body[i])
            if "buffer_store" in body[i]:
                mm = re.search(r"buffer_store_dwordx4 v\[(\d+):(\d+)\]", body[i])
            lo, hi = int(mm.group(1)), int(mm.group(2))
            if "global_store" in body[i]:
                mm = re.search(r"global_store_dwordx4 v\[\d+:\d+\], v\[(\d+):(\d+)\]", body[i])
                lo, hi = int(mm.group(1)), int(mm.group(2))
            ws = 0
            for ln in body[i + 1:i + 6]:
                ln = ln.strip()
                if not ln or ln.startswith(";"):
                    continue
                if ln.startswith("s_nop"):
                    ws += int(ln.split()[1]) + 1
                    continue
                d = re.match(r"v_\S+\s+v(?:(\d+)|\[(\d+):(\d+)\])", ln)
                if d and ws < 2:
                    a, b = (int(d.group(2)), int(d.group(3))) if d.group(2) else (int(d.group(1)), int(d.group(1)))
                    if not (b < lo or a > hi):
                        haz += 1
                        break
                ws += 1
                if ws >= 2:
                    break
        kind = "inline asm" if any("ASMSTART" in body[i - 1] for i in idx) else "compiler-emitted"
        print("%-100s %2d write-through stores (%s), data registers overwritten < 2 wait states later: %d" % (name[:100], len(idx), kind, haz))


if __name__ == "__main__":
    main()
