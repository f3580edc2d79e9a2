cat > /tmp/r.py <<'PY'
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import opticommpy_amd as oa
from helpers import synth_field, make_param, rel_l2
from opticommpy_amd import models
from oracle import ssf_oracle as orc
for N in (240000, 960000):
    E = synth_field(N, 2, 3, 8.4)
    cfg2 = dict(Fs=512e9, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Ltotal=16.0, Lspan=16.0, hz=0.08, maxIter=10, tol=1e-5, nlprMethod=False, amp="ideal", prgsBar=False, saveSpanN=[])
    oa.set_engine("fused")
    oa.manakovSSF(E, make_param(oa.parameters, cfg2)); oa.manakovSSF(E, make_param(oa.parameters, cfg2))
    r = models.last_run
    print(N, r['steps'], 'steps', round(r['device_ms'],1), 'ms', round(r['steps']/r['device_ms']*1e3), 'steps/s', flush=True)
N=240000
E = synth_field(N, 2, 3, 8.4)
cfg = dict(Fs=512e9, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, Ltotal=0.8, Lspan=0.8, hz=0.08, maxIter=10, tol=1e-5, nlprMethod=False, amp="ideal", prgsBar=False, saveSpanN=[])
tr={}; ref = orc.manakovSSF(E, make_param(orc.parameters, cfg), trace=tr)
out = oa.manakovSSF(E, make_param(oa.parameters, cfg), _trace=True)
print('parity', rel_l2(out, ref), list(models.last_run['iters'])==tr['iters'])
PY
for t in 128 256; do echo "tpr $t"; SSF_MIX_TPR=$t python /tmp/r.py; done
