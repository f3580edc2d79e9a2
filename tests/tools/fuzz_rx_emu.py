#!/usr/bin/env python3
"""Differential fuzzing of the receiver / transmitter pipelines on the CPU emulator against their oracles.
Usage: python tests/tools/fuzz_rx_emu.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_binding as eb  # noqa: E402
import opticommpy_amd as oa  # noqa: E402
from opticommpy_amd import rx as rxmod, wdm_tx  # noqa: E402
from oracle import rx_oracle as orx, tx_oracle as otx  # noqa: E402
from oracle.ssf_oracle import parameters as op  # noqa: E402


def bag(cls, kw):
    o = cls()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    rxmod._backend = eb.EmuRxBackend()
    wdm_tx._backend = eb.EmuTxBackend()
    bad = 0
    for case in range(cases):
        N = int(rng.choice([64, 257, 1000, 2048, 5000]))
        Fs = float(rng.choice([64e9, 96e9, 128e9]))
        Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * float(rng.choice([1e-3, 0.02, 0.1]))
        Elo = np.sqrt(float(rng.choice([1e-3, 1e-2]))) * np.exp(1j * 2 * np.pi * float(rng.choice([0, 1e8, -3e8])) * np.arange(N) / Fs)
        fe = dict(Fs=Fs, polRotation=float(rng.uniform(-1, 1)), pdl=float(rng.choice([0, 0.5, -1.5])),
                  polDelay=float(rng.choice([0, 2e-12, -7e-12])), ampImbX=float(rng.uniform(-1, 1)), phaseImbX=float(rng.uniform(-0.2, 0.2)),
                  timeSkewX=float(rng.choice([0, 1e-12, -3e-12])), ampImbY=float(rng.uniform(-1, 1)), phaseImbY=float(rng.uniform(-0.2, 0.2)),
                  timeSkewY=float(rng.choice([0, 5e-12])))
        pd = dict(Fs=Fs, B=float(rng.choice([10e9, 20e9, 30e9])), N=int(rng.choice([31, 64, 255, 1001])), fType=str(rng.choice(["rect", "gauss"])),
                  R=float(rng.choice([0.5, 1.0])), currentSaturation=bool(rng.integers(0, 2)), IpdSat=float(rng.choice([1e-3, 5e-3])),
                  ideal=bool(rng.integers(0, 4) == 0), bandwidthLimitation=bool(rng.integers(0, 4) != 0))
        un = rng.normal(size=(8, 2, N))

        def pdn(s):
            return un[s][0], un[s][1]

        def pol(b):
            return (pdn(b), pdn(b + 1)), (pdn(b + 2), pdn(b + 3))
        a = oa.pdmCoherentReceiver(Es, Elo, bag(oa.parameters, fe), bag(oa.parameters, pd), _unit_normals=un)
        b = orx.pdmCoherentReceiver(Es, Elo, bag(op, fe), bag(op, pd), noise=(pol(0), pol(4)))
        err = np.max(np.abs(a - b)) / np.max(np.abs(b))
        # transmitter
        txkw = dict(M=int(rng.choice([4, 16, 64])), nBits=int(rng.choice([240, 1200, 3000])), SpS=int(rng.choice([2, 4, 8, 16])),
                    nChannels=int(rng.integers(1, 5)), nPolModes=int(rng.integers(1, 3)), seed=int(rng.integers(0, 1000)),
                    laserLinewidth=float(rng.choice([0, 1e5])), pulseType=str(rng.choice(["rrc", "nrz"])), nFilterTaps=int(rng.choice([32, 129, 1024])),
                    pulseRollOff=float(rng.choice([0.01, 0.2])), mzmScale=float(rng.choice([0.25, 0.5])), prgsBar=False)
        txkw["nBits"] -= txkw["nBits"] % int(np.log2(txkw["M"]))
        ta, sa, _ = oa.simpleWDMTx(bag(oa.parameters, txkw))
        tb, sb, _ = otx.simpleWDMTx(bag(op, txkw))
        terr = np.max(np.abs(ta - tb)) / np.max(np.abs(tb))
        if not (err <= 1e-11 and terr <= 1e-11 and np.array_equal(sa, sb)):
            bad += 1
            print("MISMATCH case", case, "rx err", err, fe, pd, "tx err", terr, txkw, flush=True)
    print("done:", cases, "cases,", bad, "mismatches")


if __name__ == "__main__":
    main()
