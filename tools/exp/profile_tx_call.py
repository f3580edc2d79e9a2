import cProfile, pstats, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import opticommpy_amd as oa
def bag(**kw):
    p = oa.parameters()
    for k, v in kw.items(): setattr(p, k, v)
    return p
N = 1 << 20
tx = dict(M=16, Rs=32e9, SpS=16, nBits=4 * (N // 16), nChannels=11, nPolModes=2, seed=None, laserLinewidth=100e3, wdmGridSpacing=37.5e9, prgsBar=False)
def work():
    for _ in range(5):
        s, _, _ = oa.simpleWDMTx(bag(**tx), device_output=True)
    s.get()[:1]
work()
t = time.perf_counter(); work(); print("per call ms", (time.perf_counter() - t) / 5 * 1e3)
cProfile.run('work()', '/tmp/tx.prof')
pstats.Stats('/tmp/tx.prof').sort_stats('tottime').print_stats(14)
