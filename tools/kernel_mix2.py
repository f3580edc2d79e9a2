#!/usr/bin/env python3
"""Static instruction mix of selected kernels from a gfx950 assembly listing (tools/kernel_resources.py with SSF_KEEP_ASM=prefix
leaves <prefix>.engine_fused_f64.s / _f32.s):   python tools/kernel_mix2.py <listing.s> <mangled-name substring> ..."""
import collections
import re
import sys


def main(path, pats):
    txt = open(path).read()
    for m in re.finditer(r"^(_ZN\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end\d+:", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not any(p in name for p in pats):
            continue
        c = collections.Counter()
        for line in body.splitlines():
            mm = re.match(r"\s+([a-z][a-z_0-9]+)\s", line)
            if mm:
                c[mm.group(1)] += 1
        g = lambda f: sum(v for k, v in c.items() if f(k))   # noqa: E731
        print(name[:90])
        print("   total %d | VALU %d (f64 %d, pk_f32 %d, other f32 %d, int/mov %d) | LDS %d | VMEM %d | SALU %d | waitcnt %d barrier %d" % (
            sum(c.values()), g(lambda k: k.startswith("v_")), g(lambda k: k.endswith("_f64")), g(lambda k: k.startswith("v_pk_")),
            g(lambda k: k.startswith("v_") and k.endswith("_f32") and not k.startswith("v_pk_")),
            g(lambda k: k.startswith("v_") and not k.endswith("_f64") and not k.endswith("_f32")),
            g(lambda k: k.startswith("ds_")), g(lambda k: k.startswith("global_") or k.startswith("buffer_") or k.startswith("scratch_")),
            g(lambda k: k.startswith("s_") and k not in ("s_waitcnt", "s_barrier", "s_nop")), c["s_waitcnt"], c["s_barrier"]))
        print("   top:", ", ".join("%s %d" % kv for kv in c.most_common(16)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
