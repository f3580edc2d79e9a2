#!/usr/bin/env python3
"""Differential fuzzing of the receiver-side calls on the GPU (host arrays and DeviceArrays) against their oracles:
pdmCoherentReceiver with supplied unit normals, firFilter, delaySignal, decimate, edc, blockwiseFFTConv.
Usage: python tests/tools/fuzz_rx_gpu.py [cases] [seed] [--emu]     (--emu: the CPU emulator's kernels, host arrays, no edc: a check of this script)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from oracle import rx_oracle as orx  # noqa: E402
from oracle import ssf_oracle as orc  # noqa: E402
from oracle.ssf_oracle import parameters as op  # noqa: E402


def bag(cls, kw):
    o = cls()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    emu = "--emu" in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = int(argv[0]) if len(argv) > 0 else 60
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 0)
    if emu:
        import emu_binding as eb
        from opticommpy_amd import rx as rxmod
        rxmod._backend = eb.EmuRxBackend()
    bad = 0
    for case in range(cases):
        N = int(rng.choice([64, 257, 1000, 2048, 5000, 20000, 65536, 100003][:5 if emu else 8]))
        Fs = float(rng.choice([64e9, 96e9, 128e9]))
        dev = bool(rng.integers(0, 2)) and not emu
        Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * float(rng.choice([1e-3, 0.02, 0.1]))
        Elo = np.sqrt(float(rng.choice([1e-3, 1e-2]))) * np.exp(1j * 2 * np.pi * float(rng.choice([0, 1e8, -3e8])) * np.arange(N) / Fs)
        fe = dict(Fs=Fs, polRotation=float(rng.uniform(-1, 1)), pdl=float(rng.choice([0, 0.5, -1.5])),
                  polDelay=float(rng.choice([0, 2e-12, -7e-12])), ampImbX=float(rng.uniform(-1, 1)), phaseImbX=float(rng.uniform(-0.2, 0.2)),
                  timeSkewX=float(rng.choice([0, 1e-12, -3e-12])), ampImbY=float(rng.uniform(-1, 1)), phaseImbY=float(rng.uniform(-0.2, 0.2)),
                  timeSkewY=float(rng.choice([0, 5e-12])))
        pd = dict(Fs=Fs, B=float(rng.choice([10e9, 20e9, 30e9])), N=int(rng.choice([31, 64, 255, 1001, 2001])), fType=str(rng.choice(["rect", "gauss"])),
                  R=float(rng.choice([0.5, 1.0])), currentSaturation=bool(rng.integers(0, 2)), IpdSat=float(rng.choice([1e-3, 5e-3])),
                  ideal=bool(rng.integers(0, 4) == 0), bandwidthLimitation=bool(rng.integers(0, 4) != 0))
        un = rng.normal(size=(8, 2, N))

        def pdn(s):
            return un[s][0], un[s][1]

        def pol(b):
            return (pdn(b), pdn(b + 1)), (pdn(b + 2), pdn(b + 3))
        errs = {}
        a = oa.pdmCoherentReceiver(oa.to_device(Es) if dev else Es, oa.to_device(Elo) if dev else Elo, bag(oa.parameters, fe),
                                   bag(oa.parameters, pd), _unit_normals=un)
        b = orx.pdmCoherentReceiver(Es, Elo, bag(op, fe), bag(op, pd), noise=(pol(0), pol(4)))
        errs["pdm"] = rel(a.get() if dev else a, b)
        # filters on 1 ... 4 columns, every block size of the overlap-save kernel and the segmented path
        nc = int(rng.integers(1, 5))
        x = rng.normal(size=(N, nc)) + 1j * rng.normal(size=(N, nc))
        K = int(rng.choice([1, 2, 17, 255, 256, 683, 1024, 2049, 4096, 5001]))
        h = rng.normal(size=K) / np.sqrt(K)
        y = oa.firFilter(h, oa.to_device(x) if dev else x)
        errs["fir%d" % K] = rel(y.get() if dev else y, orx.firFilter(h, x))
        if K <= 1024 or N <= 5000:
            hb = rng.normal(size=K) + 1j * rng.normal(size=K)
            fd = bool(rng.integers(0, 2))
            yb = oa.blockwiseFFTConv(oa.to_device(x[:, 0].copy()) if dev else x[:, 0], hb, freqDomainFilter=fd)
            errs["bfc"] = rel(yb.get() if dev else yb, orc.blockwiseFFTConv(x[:, 0], hb, freqDomainFilter=fd))
        dl = float(rng.choice([0.3e-11, -1.7e-11, 4.1e-11]))
        nf = rng.choice([1024, 256, 4096]) if N <= 20000 else 1024
        yd = oa.delaySignal(oa.to_device(x[:, 0].copy()) if dev else x[:, 0], dl, Fs, NFFT=int(nf))
        errs["delay%d" % nf] = rel(yd.get() if dev else yd, orx.delaySignal(x[:, 0], dl, Fs, NFFT=int(nf)))
        sps = int(rng.choice([2, 4, 8, 16]))
        Nd = N - N % sps
        if Nd >= sps:
            dp = dict(SpSin=sps, SpSout=int(rng.choice([1, 2])))
            if sps % dp["SpSout"] == 0:
                yq = oa.decimate(oa.to_device(x[:Nd].copy()) if dev else x[:Nd], bag(oa.parameters, dp))
                errs["decimate"] = 0.0 if np.array_equal(yq.get() if dev else yq, orx.decimate(x[:Nd], bag(op, dp))) else 1.0
        if emu:
            if not all(e <= 2e-11 for e in errs.values()):
                bad += 1
                print("MISMATCH case", case, "N", N, errs, flush=True)
            continue
        ep = dict(Fs=Fs, L=float(rng.choice([1, 40, 400, 1200])), D=float(rng.choice([16, 17, -4])), Fc=193.1e12, Rs=32e9)
        ye = oa.edc(oa.to_device(x) if dev else x, bag(oa.parameters, ep))
        errs["edc"] = rel(ye.get() if dev else ye, orc.edc(x, bag(op, ep)))
        if not all(e <= 2e-11 for e in errs.values()):
            bad += 1
            print("MISMATCH case", case, "N", N, "dev", dev, "cols", nc, errs, fe, pd, ep, flush=True)
    print("done:", cases, "cases,", bad, "mismatches")


if __name__ == "__main__":
    main()
