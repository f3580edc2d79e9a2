"""Measured-and-rejected kernels live in the experiment library only (make -C opticommpy_amd/csrc exp -> libssf_hip_exp.so:
-DSSF_EXPERIMENTS=1 adds the persistent span kernels of fused_experiments.h and lets the tuning knobs of fused_engine.h be read
from the environment).  These tests keep them alive: each runs tests/tools/persistent_check.py in a process of its own against
that library, and is skipped where it has not been built.  The product library must NOT react to the switches."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "opticommpy_amd", "libssf_hip_exp.so")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["nlse", "mk0", "mk1"])
def test_persistent_span_kernels_reproduce_the_launch_sequence(which):
    if not os.path.exists(EXP):
        pytest.skip("experiment library not built (make -C opticommpy_amd/csrc exp)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "persistent_check.py"), which],
                       env=dict(os.environ, SSF_LIB=EXP), capture_output=True, timeout=900)
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-1000:], r.stderr.decode(errors="replace")[-3000:])


@pytest.mark.gpu
def test_stage_specialised_column_kernels_against_the_general_kernel_at_chip_filling_sizes():
    """tests/tools/split_check.py: the H | ADV | FIN kernels along the predicted sequence against the one general kernel on the GPU,
    2^20 ... 2^22 samples, both precisions, ten regimes (iteration-count change, adaptive step, rebuilds at every step, maxIter = 1,
    snapshots, back-propagation): same step and iteration counts, fields equal to rounding."""
    if not os.path.exists(EXP):
        pytest.skip("experiment library not built (make -C opticommpy_amd/csrc exp)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "split_check.py")],
                       env=dict(os.environ, SSF_LIB=EXP), capture_output=True, timeout=1500)
    assert r.returncode == 0, (r.stdout.decode(errors="replace")[-3000:], r.stderr.decode(errors="replace")[-2000:])


@pytest.mark.gpu
def test_product_library_ignores_the_experiment_switches(monkeypatch):
    """SSF_ROW_V / SSF_COL_V / SSF_SPLIT_L1 / SSF_PERSIST ... are experiment-build knobs: one stray environment variable must
    not change what the shipped library runs (bit-equal result, same launch count)."""
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    from opticommpy_amd import models
    E = synth_field(1 << 14, 2, 5, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=1.6, Lspan=1.6,
               hz=0.08, nlprMethod=False, amp="ideal", saveSpanN=[])
    models.release_plans()
    a = oa.manakovSSF(E, make_param(oa.parameters, cfg))
    for k, v in dict(SSF_ROW_V="16", SSF_COL_V="16", SSF_SPLIT_L1="6", SSF_PERSIST_MK="32", SSF_COL_HALF="64", SSF_ROW_FPW="1",
                     SSF_LIM0_BOUND="0").items():
        monkeypatch.setenv(k, v)
    models.release_plans()
    b = oa.manakovSSF(E, make_param(oa.parameters, cfg))
    models.release_plans()
    assert np.array_equal(a, b)
