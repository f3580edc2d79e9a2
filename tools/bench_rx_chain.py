#!/usr/bin/env python3
"""The receiver chain of bench.py's "also".rx_chain_2^20 leg by itself (pdmCoherentReceiver -> firFilter 1024 -> decimate 16 -> 2 ->
edc 800 km on the reference fixture's field, resident in HBM): as the reference's four functions one after the other and as ONE
library call (ssf_rx_chain).  Run it under `rocprofv3 --kernel-trace --stats` for the kernel list (profiles/r6_rx_chain.txt).
    python tools/bench_rx_chain.py [--reps R] [--mode four|one|both]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import synth_field  # noqa: E402


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
    mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "both"
    z = np.load(os.path.join(ROOT, "tests", "golden", "wl_rx_chain_n20.npz"))
    c = json.loads(str(z["cfg"]))

    def bag(kw):
        p = oa.parameters()
        for k, v in kw.items():
            setattr(p, k, v)
        return p
    N = int(c["synth"][0])
    E = synth_field(int(c["synth"][0]), int(c["synth"][1]), int(c["synth"][2]), float(c["synth"][3]))
    lo = oa.basicLaserModel(bag(c["lo"]))
    pulse = oa.pulseShape(bag(c["ps"]))
    Ed, Ld = oa.to_device(E), oa.to_device(lo)

    def four():
        s = oa.pdmCoherentReceiver(Ed, Ld, bag(c["fe"]), bag(c["pd"]))
        s = oa.firFilter(pulse, s)
        s = oa.decimate(s, bag(c["dec"]))
        return oa.edc(s, bag(c["edc"]))

    def one():
        return oa.pdmCoherentReceiverChain(Ed, Ld, bag(c["fe"]), bag(c["pd"]), pulse, bag(c["dec"]), bag(c["edc"]))
    d = int(c["d"])
    alg = N * (80 + 64 + 36 + 8)
    for name, fn in (("four calls", four), ("one call", one)):
        if mode not in ("both", name.split()[0]):
            continue
        out = fn().get()
        err = float(np.linalg.norm(out[::d] - z["out_dec"]) / np.linalg.norm(z["out_dec"]))
        t0 = time.perf_counter()
        for _ in range(reps):
            o = fn()
        o.get()[:1]
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:10s}: {dt * 1e3:7.3f} ms per chain at N = {N} ({alg / dt / 1e9:6.0f} GB/s algorithmic = {alg / dt / 8e12:.3f} of the HBM peak), "
              f"rel-L2 against the reference fixture {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
