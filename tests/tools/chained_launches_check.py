"""Chained launches (experiment builds of the library only): not part of the -m gpu suite.  Run on a GPU box with
    make -C opticommpy_amd/csrc variant TAG=chain VFLAGS=-DSSF_CHAIN=1
    SSF_LIB=$PWD/opticommpy_amd/libssf_hip_chain.so python -m pytest tests/tools/chained_launches_check.py -q
(the file name keeps it out of pytest's default collection).  Test infrastructure."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.gpu
@pytest.mark.skipif("_chain" not in os.path.basename(os.environ.get("SSF_LIB", "")),
                    reason="needs an experiment build of the library: make -C opticommpy_amd/csrc variant TAG=chain "
                           "VFLAGS=-DSSF_CHAIN=1, then SSF_LIB=.../libssf_hip_chain.so (the product build compiles the chained-launch "
                           "wrappers out: their mere presence cost 2 %)")
@pytest.mark.parametrize("prec", ["complex128", "complex64"])
def test_chained_launches_reproduce_the_launch_sequence(monkeypatch, prec):
    """SSF_CHAIN=1 in an experiment build (measured 1.6 - 4 x slower, profiles/r3_chained_launches_and_stagger.txt): the launches of a span
    alternate between two streams and every workgroup waits inside the kernel for the previous launch's workgroups (agent-scope
    counters, release / acquire) instead of at a kernel boundary.  Same kernels, same arithmetic: bit-equal fields, same counts."""
    import opticommpy_amd as oa
    from helpers import make_param, synth_field
    from opticommpy_amd import models
    E = synth_field(1 << 16, 2, 33, 8.4)
    cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=16.0, Lspan=8.0,
               hz=0.08, nlprMethod=True, maxNlinPhaseRot=2e-2, amp="edfa", NF=4.5, saveSpanN=[], seed=5, prec=prec)
    for adaptive in (True, False):
        runs = {}
        for chain in ("0", "1"):
            monkeypatch.setenv("SSF_CHAIN", chain)
            models.release_plans()
            out = oa.manakovSSF(E, make_param(oa.parameters, dict(cfg, nlprMethod=adaptive)))
            runs[chain] = (out, models.last_run["steps"], models.last_run["iterations"])
        assert runs["1"][1:] == runs["0"][1:] and runs["0"][1] >= (10 if adaptive else 200), runs["0"][1:]
        assert np.array_equal(runs["1"][0], runs["0"][0])
    monkeypatch.delenv("SSF_CHAIN")
    models.release_plans()
