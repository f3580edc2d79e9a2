#!/usr/bin/env python3
"""Soak test of the receiver-side calls on the GPU: thousands of calls with new filters every time (the device filter cache holds 128
entries: it fills up and later filters are per-call uploads), host arrays and DeviceArrays, every overlap-save instantiation, the
transmitter -- watching the free device memory.  Usage (GPU box): python tests/tools/soak_rx_gpu.py [rounds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opticommpy_amd as oa  # noqa: E402
from opticommpy_amd import device  # noqa: E402


def free_mem():
    hip = C.CDLL("libamdhip64.so")
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value


def bag(**kw):
    p = oa.parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(5)
    N, Fs = 1 << 16, 128e9
    Es = (rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))) * 0.02
    Elo = np.full(N, np.sqrt(8e-3), dtype=complex)
    Ed, Ld = oa.to_device(Es), oa.to_device(Elo)
    base = None
    for r in range(rounds):
        K = int(rng.choice([3, 31, 255, 700, 1500, 3000, 4500]))
        h = rng.normal(size=K)                                   # a new filter every round: the cache churns
        y = oa.firFilter(h, Ed if r & 1 else Es)
        fe = dict(Fs=Fs, polRotation=float(rng.uniform(-1, 1)), polDelay=float(rng.uniform(-5e-12, 5e-12)),
                  timeSkewX=float(rng.uniform(-2e-12, 2e-12)), ampImbX=0.3)
        s = oa.pdmCoherentReceiver(Ed if r & 1 else Es, Ld if r & 1 else Elo, bag(**fe), bag(Fs=Fs, B=float(rng.uniform(20e9, 40e9)), seed=r))
        i = oa.photodiode(Ed if r & 1 else Es, bag(Fs=Fs, B=30e9, N=int(rng.choice([33, 255, 1001])), seed=r))
        d = oa.decimate(s, bag(SpSin=4, SpSout=2))
        e = oa.edc(d, bag(Fs=Fs / 2, L=float(rng.uniform(10, 3000)), D=16, Fc=193.1e12, Rs=32e9))
        for a in (y, s, i, d, e):
            v = a.get() if device.is_device(a) else a
            assert np.all(np.isfinite(v)), r
        if r % 20 == 0:
            t, _, _ = oa.simpleWDMTx(bag(M=16, Rs=32e9, SpS=8, nBits=4 * 2048, nChannels=3, nPolModes=2, laserLinewidth=1e5, prgsBar=False), device_output=bool(r & 1))
        del y, s, i, d, e
        if r == 20:
            device.release_pool()
            base = free_mem()
    device.release_pool()
    end = free_mem()
    print("free device memory after 20 rounds / at the end: %.1f / %.1f MiB  (difference %.1f MiB)" % (base / 2**20, end / 2**20, (base - end) / 2**20))
    assert base - end < 64 * 2**20, "device memory keeps growing"
    print("soak: OK,", rounds, "rounds")


if __name__ == "__main__":
    main()
