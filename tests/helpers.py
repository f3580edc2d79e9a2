"""Shared test helpers: golden-vector loading, parameter construction, field synthesis."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    cfg = json.loads(str(d.pop("cfg")))
    return d, cfg


def make_param(cls, cfg):
    """Build a parameters object of class ``cls`` from a golden cfg dict."""
    p = cls()
    for k, v in cfg.items():
        if k == "func":
            continue
        if k == "prec":
            v = np.dtype(v).type
        setattr(p, k, v)
    return p


def synth_field(N, ncols, seed, p_dbm, dtype=np.complex128):
    """SURVEY.md 8d input recipe: band-limited complex Gaussian, each column P/2."""
    rng = np.random.default_rng(seed)
    E = (rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols))) / np.sqrt(2)
    F = np.fft.fft(E, axis=0)
    F[N // 4: 3 * N // 4, :] = 0
    E = np.fft.ifft(F, axis=0)
    p_lin = 10 ** (p_dbm / 10) * 1e-3
    E = E * np.sqrt((p_lin / 2) / np.mean(np.abs(E) ** 2, axis=0))
    return E.astype(dtype)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def split_iters(lims, tol, maxIter):
    iters, n = [], 0
    for v in lims:
        n += 1
        if v < tol or n == maxIter:
            iters.append(n)
            n = 0
    return iters
