"""Why does the device-resident notebook call at 200 000 samples sometimes take 84 ms instead of 10.5 (bench.py notebook_leg)?
Repeats the leg's sequence and profiles the slow calls.   python tools/exp/notebook_dev_outlier.py [reps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import opticommpy_amd as oa


def bag(cls, **kw):
    q = cls()
    for k, v in kw.items():
        setattr(q, k, v)
    return q


N = 200_000
tx = bag(oa.parameters, M=16, Rs=32e9, SpS=4, nBits=int(N), pulseType="rrc", nFilterTaps=4096, pulseRollOff=0.01, powerPerChannel=-2,
         nChannels=1, Fc=193.1e12, laserLinewidth=100e3, wdmGridSpacing=37.5e9, nPolModes=2, seed=int(N) % 9973, prgsBar=False)


def ch(**kw):
    return bag(oa.parameters, **dict(dict(Ltotal=500, Lspan=50, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, hz=0.5, maxIter=5, tol=1e-5,
                                          nlprMethod=True, maxNlinPhaseRot=2e-2, prgsBar=False, Fs=32e9 * 4, seed=11), **kw))


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    sig = oa.simpleWDMTx(tx)[0]
    oa.manakovSSF(sig, ch(Ltotal=50))
    t0 = time.perf_counter(); oa.manakovSSF(sig, ch()); t_np = time.perf_counter() - t0
    sig_d = oa.to_device(sig)
    oa.manakovSSF(sig_d, ch(Ltotal=50))
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); out_d = oa.manakovSSF(sig_d, ch()); pr.disable(); t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); out_d2 = oa.manakovSSF(sig_d, ch()); t_dev2 = time.perf_counter() - t0
    print("rep %d: numpy %.1f ms, device %.1f ms, device again %.1f ms, device_ms %.1f" % (rep, t_np * 1e3, t_dev * 1e3, t_dev2 * 1e3,
                                                                                       float(oa.last_run.get("device_ms", 0))), flush=True)
    if t_dev > 0.03:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:2500])
    del sig_d, out_d, out_d2
    oa.models.release_plans()
