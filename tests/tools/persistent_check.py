#!/usr/bin/env python3
"""The persistent span kernels (opticommpy_amd/csrc/fused_experiments.h) against the launch sequence and the oracle.
Run by tests/test_experiments.py in a process of its own with SSF_LIB = the experiment library (make -C opticommpy_amd/csrc exp):
the product library neither contains these kernels nor reads their switches.

    SSF_LIB=opticommpy_amd/libssf_hip_exp.so python tests/tools/persistent_check.py nlse | mk0 | mk1"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import opticommpy_amd as oa  # noqa: E402
from helpers import make_param, rel_l2, synth_field  # noqa: E402
from opticommpy_amd import models  # noqa: E402
from oracle import ssf_oracle as orc  # noqa: E402


def nlse():
    """ssfm with every stage of a span in one persistent launch (grid barrier between stages): same kernel bodies (other tile
    widths, so rounding-level differences only)."""
    E = synth_field(1 << 16, 1, 1, 0.0).reshape(-1) * np.sqrt(2)
    cfg = dict(Fs=512e9, Ltotal=100, Lspan=50, hz=0.5, alpha=0.2, D=16, gamma=1.3, Fc=193.1e12, amp="ideal", prgsBar=False, saveSpanN=[])
    out = {}
    for mode in ("0", "256"):
        os.environ["SSF_PERSIST"] = mode
        models.release_plans()
        out[mode] = oa.ssfm(E, make_param(oa.parameters, cfg))
    assert rel_l2(out["256"], out["0"]) <= 1e-12
    assert rel_l2(out["256"], orc.ssfm(E, make_param(orc.parameters, cfg))) <= 1e-10


def mk(xcd):
    """A whole Manakov span as ONE persistent launch (k_mk_span): the stage bodies and the device-resident control flow are the
    launch sequence's, so the iteration counts are identical and the field agrees to rounding -- with the agent-scope barrier
    and with the one that only admits the workgroups of one XCD.  Adaptive step and an amplifier between the spans included."""
    E = synth_field(1 << 14, 2, 31, 8.4)
    os.environ["SSF_COL_HALF"] = "128"                       # 256-thread column workgroups, as the merged kernel needs
    os.environ["SSF_ROW_V"] = os.environ["SSF_COL_V"] = "16"
    for adaptive in (False, True):
        cfg = dict(Fs=512e9, Fc=193.1e12, alpha=0.2, D=16, gamma=1.3, maxIter=10, tol=1e-5, prgsBar=False, Ltotal=8.0, Lspan=4.0,
                   hz=0.08, nlprMethod=adaptive, maxNlinPhaseRot=2e-2, amp="edfa", NF=4.5, saveSpanN=[])
        runs = {}
        for workers in ("0", "32"):
            os.environ["SSF_PERSIST_MK"] = workers
            os.environ["SSF_PERSIST_XCD"] = xcd
            models.release_plans()
            out = oa.manakovSSF(E, make_param(oa.parameters, dict(cfg, seed=3)))
            runs[workers] = (out, models.last_run["steps"], models.last_run["iterations"])
        assert runs["32"][1:] == runs["0"][1:]
        assert rel_l2(runs["32"][0], runs["0"][0]) <= 1e-12
        tr = {}
        ref = orc.manakovSSF(E, make_param(orc.parameters, dict(cfg, amp="ideal")), trace=tr)
        os.environ["SSF_PERSIST_MK"] = "32"
        out = oa.manakovSSF(E, make_param(oa.parameters, dict(cfg, amp="ideal")))
        assert models.last_run["iterations"] == tr["iterations"] and rel_l2(out, ref) <= 1e-10


if __name__ == "__main__":
    assert "exp" in os.path.basename(os.environ.get("SSF_LIB", "")), "run with SSF_LIB=.../libssf_hip_exp.so"
    {"nlse": nlse, "mk0": lambda: mk("0"), "mk1": lambda: mk("1")}[sys.argv[1]]()
    print("ok")
