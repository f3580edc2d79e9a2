"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/ssf.h declares, struct layouts agree with the header, and the host
wrapper's argument handling mirrors the reference -- no GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import opticommpy_amd as oa
from opticommpy_amd import _lib, models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ssf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"libssf_hip.so does not export {n}"
        assert n in _lib.SYMBOLS, f"python binding misses {n}"
    assert sorted(_lib.SYMBOLS) == names


def test_struct_layouts_match_header(tmp_path):
    """Compile include/ssf.h with gcc and compare sizeof/offsetof with the ctypes mirror."""
    import subprocess
    structs = {"ssf_params": _lib.Params, "ssf_stats": _lib.Stats, "ssf_trace": _lib.Trace,
               "ssf_device_info_t": _lib.DeviceInfo, "ssf_kernel_times": _lib.KernelTimes}
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"ssf.h\"\nint main(void){" + "".join(lines) + "return 0;}"
    c = tmp_path / "layout.c"
    c.write_text(src)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls)
        for f, _ in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)


def test_version_and_error_strings():
    lib = _lib.load()
    assert b"gfx950" in lib.ssf_version()
    assert lib.ssf_last_error(None) is not None


def test_bad_arguments_are_rejected_without_a_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ssf_plan_create(0, 1, 2, _lib.SSF_C128, 0, C.byref(h)) == -1      # N < 2
    assert lib.ssf_plan_create(0, 1024, 2, 7, 0, C.byref(h)) == -1               # bad precision
    assert lib.ssf_upload(None, None) == -1
    assert lib.ssf_plan_destroy(None) == 0


@pytest.mark.skipif(oa.checkGPU(), reason="needs a box WITHOUT a GPU")
def test_no_cpu_fallback_when_no_gpu():
    p = oa.parameters()
    p.Fs = 64e9
    p.prgsBar = False
    E = np.ones((64, 2), dtype=complex)
    with pytest.raises(RuntimeError):
        oa.manakovSSF(E, p)
    with pytest.raises(RuntimeError):
        oa.ssfm(E[:, 0], p)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libssf_hip.so")
    with pytest.raises(RuntimeError, match="HIP extension not built"):
        _lib.load()


def test_missing_fs_raises_attribute_error():
    with pytest.raises(AttributeError):
        oa.manakovSSF(np.ones((8, 2), complex), oa.parameters())
    with pytest.raises(AttributeError):
        oa.ssfm(np.ones(8, complex), oa.parameters())
    with pytest.raises(AttributeError):
        oa.manakovDBP(np.ones((8, 2), complex), oa.parameters())


def test_shape_rules_match_reference():
    p = oa.parameters()
    p.Fs = 64e9
    with pytest.raises(IndexError):
        oa.manakovSSF(np.ones(8, complex), p)                       # 1-D input (channels.py:364)
    p = oa.parameters()
    p.Fs = 64e9
    with pytest.raises(ValueError, match="could not broadcast"):
        oa.manakovSSF(np.ones((8, 4), complex), p)                  # K = 2 with default saveSpanN
    # defaults were written back before the error, as in the reference
    assert p.saveSpanN == [400 // 80] and p.maxIter == 10 and p.amp == "edfa"


def test_captured_spans_follow_reference_membership_rule():
    assert models._captured_spans([2, 1], 3) == [1, 2]
    assert models._captured_spans([5.0], 5) == [5]
    assert models._captured_spans([7], 5) == []
    assert models._captured_spans([1, 1, 2], 2) == [1, 2]


def test_span_noise_reproduces_reference_draw_order():
    from oracle import ssf_oracle as orc
    n = models._span_noise(2, 16, 0.5, 11, True, np.complex128)
    ref = orc.gaussianComplexNoise((1, 16), 0.5, 11)
    assert np.array_equal(n[0], ref[0]) and np.array_equal(n[1], ref[0])   # x and y share the seed


def test_parameters_object():
    p = oa.parameters()
    p.Fs, p.name, p.arr = 64e9, "x", [1, 2]
    q = p.copy()
    q.arr.append(3)
    assert p.arr == [1, 2]
    assert p.to_engineering_notation(64e9) == "64.0 G"
    assert p.to_engineering_notation(5) == 5


def test_set_power_for_par_ssfm():
    rng = np.random.default_rng(0)
    sig = rng.normal(size=(256, 4)) + 1j * rng.normal(size=(256, 4))
    out = oa.setPowerforParSSFM(sig, np.array([0.0, 3.0]))
    pw = np.mean(np.abs(out) ** 2, axis=0)
    np.testing.assert_allclose(pw, [0.5e-3, 0.5e-3, 0.5e-3 * 10 ** 0.3, 0.5e-3 * 10 ** 0.3], rtol=1e-12)


def test_amplifier_and_passive_optics_take_device_arrays_of_their_own_precision_only():
    """edfa / linearFiberChannel / pbs / opticalHybrid2x4 have device forms (round 6): a complex128 DeviceArray stays in HBM; one of
    another precision is refused before anything runs (no hidden conversion round trip), and nothing is 'host glue' any more."""
    import numpy as np
    import opticommpy_amd as oa
    from opticommpy_amd import device
    assert not hasattr(device, "host_only")
    d = object.__new__(oa.DeviceArray)                      # (no GPU needed: the check is on the type)
    d.shape, d.dtype, d.device, d._ptr, d._owner = (8,), np.dtype(np.complex64), 0, None, d
    d2 = d.reshape(4, 2)
    p = oa.parameters()
    p.Fs = 1e9
    for f in (lambda: oa.pbs(d2), lambda: oa.opticalHybrid2x4(d, d), lambda: oa.edfa(d2, p)):
        with pytest.raises(TypeError, match="complex128"):
            f()
