"""INTEGRATION.md section 2 shows the ctypes stub a reference maintainer would add (optic/models/modelsHIP.py).  These tests take the
stub out of the document VERBATIM and hold it to the ABI: its structs against include/ssf.h (through opticommpy_amd._lib, which
tests/test_abi.py compares with the gcc-compiled header), and -- on the GPU -- its manakovSSF, which goes through ssf_run, against
a reference-generated golden vector.  A document that drifts from the header fails here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import load_golden, make_param, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding inside OptiCommPy"):text.index("## 3. Ownership")]
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def _load_stub():
    """exec the first code block of section 2 as it stands; `CDLL("libssf_hip.so")` resolves because the in-tree library is loaded
    first and carries that SONAME (opticommpy_amd/csrc/Makefile)."""
    from opticommpy_amd import _lib
    _lib.load()
    C.CDLL(os.path.join(ROOT, "opticommpy_amd", "libssf_hip.so"), mode=C.RTLD_GLOBAL)
    ns = {}
    exec(compile(_blocks()[0], "INTEGRATION.md#2", "exec"), ns)
    return ns


def _same_layout(doc_struct, lib_struct):
    a = [(n, C.sizeof(t), getattr(doc_struct, n).offset) for n, t in doc_struct._fields_]
    b = [(n, C.sizeof(t), getattr(lib_struct, n).offset) for n, t in lib_struct._fields_]
    assert a == b, (a, b)
    assert C.sizeof(doc_struct) == C.sizeof(lib_struct)


def test_the_documented_structs_are_the_headers():
    from opticommpy_amd import _lib
    ns = _load_stub()
    _same_layout(ns["_Params"], _lib.Params)
    rx = {"C": C, "np": np}
    src = _blocks()[1]
    exec(compile(src[:src.index("def pdmCoherentReceiver")], "INTEGRATION.md#2-rx", "exec"), rx)
    _same_layout(rx["_RxParams"], _lib.RxParams)
    # every entry point the document's table names is declared in the header, and the other way round
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    header = open(os.path.join(ROOT, "include", "ssf.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\**(ssf_[a-z0-9_]+)\(", header, flags=re.M))
    named = set()
    for m in re.finditer(r"`(ssf_[a-z0-9_/]+)", text):
        parts = m.group(1).split("/")
        named.add(parts[0])
        stem = parts[0].rsplit("_", 1)[0]
        for extra in parts[1:]:                                # `ssf_plan_create/destroy`, `ssf_comm_send/recv`
            named.add(extra if extra.startswith("ssf_") else stem + "_" + extra)
    missing = {d for d in declared if d not in named and not any(d.startswith(n) for n in named)}
    assert not missing, "declared in include/ssf.h, absent from INTEGRATION.md: %s" % sorted(missing)


@pytest.mark.gpu
def test_the_documented_binding_runs_and_matches_the_reference():
    ns = _load_stub()
    import opticommpy_amd as oa
    # (the stub returns the field after the last span: the last two columns of a golden vector with saveSpanN = [1, 2], the whole
    #  output of one with saveSpanN = [] -- two coupled pairs there)
    for name in ("mk_fix_p8_ideal_2span", "mk_fix_p13_ideal_k2"):
        d, cfg = load_golden(name)
        p = make_param(oa.parameters, cfg)
        for k, v in (("NF", 4.5), ("seed", None), ("maxIter", 10), ("tol", 1e-5), ("nlprMethod", True), ("maxNlinPhaseRot", 2e-2)):
            setattr(p, k, getattr(p, k, v))                     # ("... defaults exactly as optic/models/channels.py:305-322 ...")
        out = ns["manakovSSF"](d["Ei"].copy(), p)
        ref = d["out"][:, -d["Ei"].shape[1]:]
        assert out.shape == ref.shape and rel_l2(out, ref) <= 1e-10, name
