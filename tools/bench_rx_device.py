#!/usr/bin/env python3
"""Receiver / transmitter side with everything resident in HBM (DeviceArray in, DeviceArray out): time per call and the
algorithmic bytes it has to move at least (inputs once in, outputs once out), for the kernel summaries under profiles/
(run under `rocprofv3 --kernel-trace --stats`).    python tools/bench_rx_device.py [log2n ...] [--reps R] [--json] [--cases a,b]
(cases: pdm_notebook pdm_impaired pdm_defaults photodiode firFilter255 decimate16to2 edc800km simpleWDMTx11ch; default all)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opticommpy_amd as oa  # noqa: E402
from opticommpy_amd import _lib  # noqa: E402


def bag(**kw):
    p = oa.parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def sync():
    lib = _lib.load()
    d = oa.to_device(np.zeros(1, dtype=np.complex128))
    d.get()


def timeit(f, reps):
    f()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    sync()
    return (time.perf_counter() - t0) / reps, r


CASES = {
    # examples/test_WDM_transmission.ipynb cell 18: polarisation rotation + delay, ideal photodiodes, no IQ impairments
    "pdm_notebook": (dict(polRotation=np.pi / 3, pdl=0, polDelay=3 / 32e9), dict(B=32e9, ideal=True)),
    # every stage on: PDL, polarisation delay, band-limited noisy photodiodes, IQ imbalance and skew
    "pdm_impaired": (dict(polRotation=0.2, pdl=1.0, polDelay=2e-12, ampImbX=0.5, phaseImbY=0.05, timeSkewX=1e-12, timeSkewY=-2e-12),
                     dict(B=25e9, seed=1)),
    # band-limited noisy photodiodes (the defaults of photodiode()), nothing else
    "pdm_defaults": (dict(), dict(B=30e9, seed=2)),
}


def main():
    argv = sys.argv[1:]
    opt = {}
    for k in ("--reps", "--cases", "--edc-block"):       # options with a value
        if k in argv:
            i = argv.index(k)
            opt[k] = argv[i + 1]
            del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    reps = int(opt.get("--reps", 10))
    if "--edc-block" in opt:                             # experiment: edc's overlap-save block size (models._ols_block)
        from opticommpy_amd import models
        models._ols_block = lambda K, _n=int(opt["--edc-block"]): _n
    out = {}
    only = opt["--cases"].split(",") if "--cases" in opt else None
    want = lambda name: only is None or name in only  # noqa: E731
    rng = np.random.default_rng(1)
    for lg in [int(a) for a in args] or [20, 22]:
        N = 1 << lg
        Fs = 512e9
        Es = oa.to_device((rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02)
        Elo = oa.to_device(np.full(N, np.sqrt(8e-3), dtype=complex))
        for name, (fe, pd) in CASES.items():
            if not want(name):
                continue
            t, r = timeit(lambda: oa.pdmCoherentReceiver(Es, Elo, bag(Fs=Fs, **fe), bag(Fs=Fs, **pd)), reps)
            alg = (2 + 1 + 2) * 16 * N                       # signal and LO in, detected signal out
            out["%s_2^%d" % (name, lg)] = dict(ms=t * 1e3, alg_MiB=alg / 2**20, GBs=alg / t / 1e9, frac=alg / t / 8e12)
        if want("photodiode"):                            # (N, 2) field -> one real photocurrent: defaults (noisy, band-limited) and ideal
            for tag, kw in (("defaults", dict(B=30e9, seed=2)), ("ideal", dict(B=30e9, ideal=True))):
                t, _ = timeit(lambda: oa.photodiode(Es, bag(Fs=Fs, **kw)), reps)
                out["photodiode_%s_2^%d" % (tag, lg)] = dict(ms=t * 1e3, alg_MiB=40 * N / 2**20, GBs=40 * N / t / 1e9, frac=40 * N / t / 8e12)
        h = oa.lowPassFIR(25e9, Fs, 255)
        if want("firFilter255"):
            t, _ = timeit(lambda: oa.firFilter(h, Es), reps)
            out["firFilter255_2^%d" % lg] = dict(ms=t * 1e3, alg_MiB=64 * N / 2**20, GBs=64 * N / t / 1e9, frac=64 * N / t / 8e12)
        if want("decimate16to2"):
            t, _ = timeit(lambda: oa.decimate(Es, bag(SpSin=16, SpSout=2)), reps)
            alg = (32 + 4) * N
            out["decimate16to2_2^%d" % lg] = dict(ms=t * 1e3, alg_MiB=alg / 2**20, GBs=alg / t / 1e9, frac=alg / t / 8e12)
        if want("edc800km"):
            p = bag(L=800, D=16, Fc=193.1e12, Rs=32e9, Fs=64e9)
            t, _ = timeit(lambda: oa.edc(Es, p), reps)
            out["edc800km_2^%d" % lg] = dict(ms=t * 1e3, alg_MiB=64 * N / 2**20, GBs=64 * N / t / 1e9, frac=64 * N / t / 8e12)
        del Es, Elo
        if not want("simpleWDMTx11ch"):
            continue
        # transmitter: N = nSymbols * 16 samples, 11 channels x 2 polarisations accumulate into one (N, 2) field
        nb = 4 * (N // 16)
        tx = dict(M=16, Rs=32e9, SpS=16, nBits=nb, nChannels=11, nPolModes=2, seed=123, laserLinewidth=100e3, wdmGridSpacing=37.5e9, prgsBar=False)
        t0 = time.perf_counter()
        sig, _, _ = oa.simpleWDMTx(bag(**tx), device_output=True)
        sync()
        t = time.perf_counter() - t0
        alg = 32 * N * 11 * 2                                 # every channel / polarisation: one modulated field written, read and added once
        out["simpleWDMTx11ch_2^%d" % lg] = dict(ms=t * 1e3, alg_MiB=alg / 2**20, GBs=alg / t / 1e9, frac=alg / t / 8e12, note="wall time incl. the host draws")
        # unseeded: the laser phase-noise walks are generated on the device (Philox); the host still draws the bits
        t0 = time.perf_counter()
        sig, _, _ = oa.simpleWDMTx(bag(**dict(tx, seed=None)), device_output=True)
        sync()
        t = time.perf_counter() - t0
        out["simpleWDMTx11ch_unseeded_2^%d" % lg] = dict(ms=t * 1e3, alg_MiB=alg / 2**20, GBs=alg / t / 1e9, frac=alg / t / 8e12, note="wall time; phase noise on the device")
    if "--json" in sys.argv:
        print(json.dumps(out))
    else:
        for k, v in out.items():
            print("%-28s %9.3f ms  %8.1f MiB algorithmic  %7.1f GB/s  %.3f of 8 TB/s %s" % (k, v["ms"], v["alg_MiB"], v["GBs"], v["frac"], v.get("note", "")))


if __name__ == "__main__":
    main()
