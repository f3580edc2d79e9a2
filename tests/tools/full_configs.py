#!/usr/bin/env python3
"""Run BASELINE.json's configurations at full size through the public Python API (on a GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa
from opticommpy_amd import models, mgpu, _lib
from helpers import synth_field
from oracle import ssf_oracle as orc


def par(**kw):
    p = oa.parameters()
    base = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False)
    base.update(kw)
    for k, v in base.items():
        setattr(p, k, v)
    return p


def report(tag, t, E, out):
    r = models.last_run
    print(f"{tag}: {t:7.3f} s wall, {r['steps']} steps, {r['iterations']/max(r['steps'],1):.2f} it/step, device {r['device_ms']:.1f} ms, "
          f"{r['steps']/(r['device_ms']*1e-3):.0f} steps/s, nonconverged {r['nonconverged_steps']}, "
          f"P_out/P_in {orc.signalPower(out)/orc.signalPower(E):.6f}", flush=True)


E = synth_field(1 << 16, 1, 1, 0.0).reshape(-1) * np.sqrt(2)
t0 = time.time(); out = oa.ssfm(E, par(Ltotal=50, Lspan=50, hz=0.5, amp=None, saveSpanN=[])); report("C1 ssfm 2^16 100 steps", time.time() - t0, E, out)
E = synth_field(1 << 20, 2, 2, 8.4)
t0 = time.time(); out = oa.manakovSSF(E, par(Ltotal=80, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])); report("C2 manakovSSF 2^20 c128 80 km", time.time() - t0, E, out)
t0 = time.time(); out = oa.manakovSSF(E, par(Ltotal=80, Lspan=80, hz=0.08, nlprMethod=True, amp="edfa", seed=1)); report("   same, adaptive + EDFA + default saveSpanN", time.time() - t0, E, out)
back = oa.manakovDBP(out, par(Ltotal=80, Lspan=80, hz=10, nlprMethod=False, amp="edfa", saveSpanN=[]))
print("   C5-style DBP (hz = 10 km) residual vs launch field:", float(np.linalg.norm(back - E) / np.linalg.norm(E)))
E3 = synth_field(1 << 22, 2, 3, 8.4, np.complex64)
t0 = time.time(); out = oa.manakovSSF(E3, par(Ltotal=800, Lspan=80, hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[], prec=np.complex64)); report("C3 manakovSSF 2^22 c64 10 x 80 km", time.time() - t0, E3, out)
# C4: 16 independent fields, launch-power sweep, one GPU (threads entry point)
fields = np.stack([synth_field(1 << 20, 2, 100 + u, 8.4 - 8 + 0.5 * u).T for u in range(16)])
cp = models._fill_params(_lib.MODEL_MANAKOV, +1, par(Ltotal=8, Lspan=8, hz=0.08, nlprMethod=False, amp="ideal", NF=4.5), 512e9, 1, np.zeros(0, np.int32))
t0 = time.time(); outs, stats = mgpu.run_threads(fields, cp, devices=[0]); t = time.time() - t0
print(f"C4 16 fields x {stats[0]['steps']} steps on one GPU: {t:.2f} s wall, {sum(s['steps'] for s in stats)/t:.0f} field-steps/s; iterations/step {[round(s['iterations']/s['steps'],2) for s in stats]}")
