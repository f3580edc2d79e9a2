import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opticommpy_amd as oa
def bag(cls, **kw):
    q = cls()
    for k, v in kw.items(): setattr(q, k, v)
    return q
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
tx = bag(oa.parameters, M=16, Rs=32e9, SpS=4, nBits=int(N), pulseType="rrc", nFilterTaps=4096, pulseRollOff=0.01, powerPerChannel=-2, nChannels=1, Fc=193.1e12, laserLinewidth=100e3, wdmGridSpacing=37.5e9, nPolModes=2, seed=int(N) % 9973, prgsBar=False)
sig = oa.simpleWDMTx(tx)[0]
def ch(**kw):
    return bag(oa.parameters, **dict(dict(Ltotal=500, Lspan=50, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, hz=0.5, maxIter=5, tol=1e-5, nlprMethod=True, maxNlinPhaseRot=2e-2, prgsBar=False, Fs=32e9 * 4, seed=11), **kw))
oa.manakovSSF(sig, ch(Ltotal=50))
for i in range(3):
    t0 = time.perf_counter(); out = oa.manakovSSF(sig, ch()); print("numpy", time.perf_counter() - t0, oa.last_run.get("device_ms"), oa.last_run.get("steps"))
sig_d = oa.to_device(sig)
for i in range(3):
    t0 = time.perf_counter(); out_d = oa.manakovSSF(sig_d, ch()); print("device", time.perf_counter() - t0, oa.last_run.get("device_ms"), oa.last_run.get("steps"), oa.last_run.get("pipeline"))
