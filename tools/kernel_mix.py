#!/usr/bin/env python3
"""Instruction mix of selected kernels from a gfx950 assembly listing (tools/kernel_resources.py, SSF_KEEP_ASM=...).
    python tools/kernel_mix.py listing.s substring [substring ...]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    for m in re.finditer(r"\n(_Z\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end", txt, re.S):
        name, body = m.group(1), m.group(2)
        if not any(k in name for k in sys.argv[2:]):
            continue
        ops = collections.Counter()
        for line in body.split("\n"):
            line = line.strip()
            if not line or line.startswith((".", ";")) or line.endswith(":"):
                continue
            ops[line.split()[0]] += 1
        groups = collections.Counter()
        for op, c in ops.items():
            if op.startswith("v_pk_"): groups["v_pk"] += c
            elif "f64" in op: groups["f64"] += c
            elif "f32" in op: groups["f32"] += c
            elif op.startswith("ds_"): groups["lds"] += c
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): groups["mem"] += c
            elif op.startswith("v_"): groups["v_other"] += c
            elif op.startswith("s_"): groups["scalar"] += c
            else: groups["other"] += c
        print(name[:90], sum(ops.values()), dict(groups))
        print("    ", ops.most_common(16))


if __name__ == "__main__":
    main()
