import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the C-ABI library the
    # same way __graft_entry__.build() does, so that the suite does not depend on call order
    lib = os.path.join(ROOT, "opticommpy_amd", "libssf_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "opticommpy_amd", "csrc"), "-j8", "-s"])
