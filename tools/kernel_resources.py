#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the fused engine, read from the gfx950 assembly metadata.
    python tools/kernel_resources.py [extra hipcc flags ...]     (works without a GPU; ~2 min)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opticommpy_amd", "csrc")


def main():
    keep = os.environ.get("SSF_KEEP_ASM")                      # path prefix: keep / reuse the assembly listings
    txt = ""
    with tempfile.TemporaryDirectory() as td:
        for unit in ("engine_fused_f64", "engine_fused_f32"):  # the two translation units that hold the kernels
            asm = (keep + "." + unit + ".s") if keep else os.path.join(td, unit + ".s")
            if not (keep and os.path.exists(asm)):
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                       "-I" + CSRC, "-Wno-pass-failed", "-mllvm", "-amdgpu-use-amdgpu-trackers=1",      # (the Makefile's FUSED_FLAGS)
                                       "--cuda-device-only", "-S"] + sys.argv[1:] +
                                      [os.path.join(CSRC, unit + ".hip"), "-o", asm], stderr=subprocess.DEVNULL)
            txt += open(asm).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
        blk = m.group(0)
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]   # noqa: E731
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"ssf::\(anonymous namespace\)::|ssf::fused::|void ", "", name)
        name = re.sub(r"\((RowArgs|ColArgs|AmpArgs|OlsArgs)<.*", "", name)
        rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                     g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("%-64s %5s %6s %5s %6s %8s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratch"))
    for r in sorted(rows):
        print("%-64s %5s %6s %5s %6s %8s" % r[:6])


if __name__ == "__main__":
    main()
