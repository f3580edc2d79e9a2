#!/usr/bin/env python3
"""Unit-steps/s of U small independent fields: one call per unit against ONE batch of independent units per launch
(ssf_plan_set_units through mgpu.run_sharded).  Usage (GPU box):  python tools/bench_units.py [log2N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, synth_field  # noqa: E402
from opticommpy_amd import mgpu, models  # noqa: E402


def main():
    U = int(os.environ.get("UNITS", "16"))
    steps = int(os.environ.get("STEPS", "200"))
    for lg in [int(s) for s in sys.argv[1:]] or [12, 14, 16, 18]:
        N = 1 << lg
        fields = [synth_field(N, 2, 80 + u, 2.0 + 0.3 * u) for u in range(U)]
        cfg = dict(Fs=64e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=steps * 0.5,
                   Lspan=steps * 0.5, hz=0.5, nlprMethod=False, amp="ideal", saveSpanN=[])
        res = {}
        for mode in ("0", "1", "0", "1"):
            os.environ["SSF_MGPU_BATCH"], os.environ["SSF_MGPU_LANES"] = mode, "1"
            t0 = time.perf_counter()
            outs = mgpu.run_sharded(fields, make_param(oa.parameters, cfg))
            res[mode] = (time.perf_counter() - t0, models.last_run["device_ms"], models.last_run["steps"], outs)
        same = all(np.array_equal(a, b) for a, b in zip(res["0"][3], res["1"][3]))
        one_dev = res["0"][1] * U                               # (last_run of the one-at-a-time path: the last unit's device time)
        print(f"N=2^{lg} U={U} steps={steps}: one at a time {U * steps / res['0'][0]:9.0f} unit-steps/s wall "
              f"({U * steps / (one_dev * 1e-3):9.0f} device), batched {U * steps / res['1'][0]:9.0f} wall "
              f"({res['1'][2] / (res['1'][1] * 1e-3):9.0f} device): x{res['0'][0] / res['1'][0]:.1f} wall, "
              f"x{one_dev / res['1'][1]:.1f} device, bit-equal {same}", flush=True)
    # config 1 as a batch: 16 scalar fields of 2^16 through ssfm's plan with 16 rows (rows are independent in the NLSE model)
    for U1 in (1, 16):
        N = 1 << 16
        E = np.stack([synth_field(N, 1, 1 + u, 0.0)[:, 0] for u in range(U1)], axis=0)
        from opticommpy_amd import _lib
        import ctypes as C
        lib = _lib.load()
        h = C.c_void_p()
        _lib.raise_for(lib, None, lib.ssf_plan_create(0, N, U1, _lib.SSF_C128, 0, C.byref(h)))
        cp = _lib.Params()
        cp.model, cp.direction = _lib.MODEL_NLSE, 1
        cp.Fs, cp.Fc, cp.alpha, cp.D, cp.gamma = 512e9, 193.1e12, 0.2, 16.0, 1.3
        cp.Lspan, cp.Nspans, cp.hz, cp.maxIter, cp.tol, cp.amp = 1000 * 0.5, 1, 0.5, 1, 0.0, _lib.AMP_NONE
        st = _lib.Stats()
        soa = np.ascontiguousarray(E)
        for _ in range(2):
            lib.ssf_upload(h, soa.ctypes.data_as(C.c_void_p))
            st = _lib.Stats()
            _lib.raise_for(lib, h, lib.ssf_execute(h, C.byref(cp), 1, 1, None, C.byref(st), None))
        s = 16.0
        gbs = U1 * st.steps * 2 * 2 * s * N / (st.device_ms * 1e-3) / 1e9     # FFT + IFFT per step and row, read + written
        print(f"config 1 (ssfm 2^16, hz 0.5) with {U1:2d} field(s) in the plan: {U1 * st.steps / (st.device_ms * 1e-3):9.0f} field-steps/s, "
              f"{gbs:7.0f} GB/s algorithmic = {gbs / 8000:.3f} of 8 TB/s")
        lib.ssf_plan_destroy(h)


if __name__ == "__main__":
    main()
