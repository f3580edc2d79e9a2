"""Multi-GPU sharding of INDEPENDENT fields.

A single (N, 2) field is one global FFT per transform and is never split.  What shards
is a batch of independent units -- WDM channels simulated separately, launch-power sweep
points (examples/test_NLC_withDBP_WDM_transmission.ipynb:660-675), Monte-Carlo fibre
realisations: unit u of U goes to rank u*G//U (contiguous blocks), there is no per-step
communication, and the only collective is the final gather of results.

Two ways to use the GPUs of one node:

* one process per GPU (``torchrun`` / ``python -m torch.distributed.run``), each rank calling
  :func:`run_sharded`: torch.distributed ("nccl" = RCCL over xGMI on ROCm, "gloo" on CPU-only
  test boxes) is used ONLY for the final all-gather of the outputs;
* one process, one host thread per device, through the C entry point ``ssf_mgpu_run``:
  :func:`run_threads` (what a one-node notebook user needs; no torch involved).
"""
import copy
import ctypes as C
import os

import numpy as np

from . import _lib


def shard_range(n_units, world, rank):
    """Contiguous block of unit indexes owned by `rank` (same rule as ssf_mgpu_run)."""
    return range(n_units * rank // world, n_units * (rank + 1) // world)


def owner_of(unit, n_units, world):
    for r in range(world):
        if unit in shard_range(n_units, world, r):
            return r
    raise IndexError(unit)


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


_LANES = {}


def _lane_pool(lanes):
    """Long-lived worker threads (their cached plans are reused from call to call)."""
    pool = _LANES.get(lanes)
    if pool is None:
        from concurrent.futures import ThreadPoolExecutor
        pool = _LANES[lanes] = ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="ssf-lane")
    return pool


def run_sharded(fields, param, compute=None, gather=True):
    """Propagate a list of independent fields, sharded over the ranks of the initialised
    torch.distributed group (or all of them locally when there is no group).

    fields : sequence of (N, 2K) arrays, identical shape/dtype on every rank
    param  : parameters object (deep-copied per unit, so defaults written back by one unit
             never leak into another)
    compute: callable(Ei, param) -> Eout; default opticommpy_amd.manakovSSF (GPU)
    gather : all-gather the outputs so every rank returns the full list (else: own units, None elsewhere)
    """
    if compute is None:
        from .models import manakovSSF as compute
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    n = len(fields)
    mine = shard_range(n, world, rank)
    outs = [None] * n
    # Two lanes per GPU: the rank's units are taken by two host threads (each with its own plan and stream; ctypes
    # releases the GIL inside the library), so that one unit's transfers overlap the other's kernels and the kernels of
    # two independent fields fill each other's load / store phases (ssf_mgpu_run does the same).
    lanes = max(1, min(int(os.environ.get("SSF_MGPU_LANES", "2")), len(mine)))

    def one(u):
        return u, np.asarray(compute(fields[u], copy.deepcopy(param)))

    if lanes > 1:
        for u, o in _lane_pool(lanes).map(one, mine):
            outs[u] = o
    else:
        for u in mine:
            outs[u] = one(u)[1]
    if not (dist and gather and world > 1):
        return outs
    import torch
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    # equal-sized payload per rank: pad to the largest block
    per = max(len(shard_range(n, world, r)) for r in range(world))
    probe = outs[mine[0]] if len(mine) else None
    meta = [None] * world
    dist.all_gather_object(meta, None if probe is None else (probe.shape, probe.dtype.str))
    shape, dt = next(m for m in meta if m is not None)
    dt = np.dtype(dt)
    buf = np.zeros((per,) + tuple(shape), dtype=dt)
    for i, u in enumerate(mine):
        buf[i] = outs[u]
    t = torch.view_as_real(torch.from_numpy(buf)).to(dev) if np.iscomplexobj(buf) else torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)                       # the one collective of the data path
    for r in range(world):
        arr = parts[r].cpu()
        arr = torch.view_as_complex(arr).numpy() if np.iscomplexobj(buf) else arr.numpy()
        for i, u in enumerate(shard_range(n, world, r)):
            outs[u] = arr[i].astype(dt, copy=False)
    return outs


def run_threads(fields, cparams, devices, precision=np.complex128, engine="auto"):
    """Single-process multi-GPU: one host thread per device inside libssf_hip.so
    (ssf_mgpu_run).  `fields`: (U, rows, N) SoA array; `cparams`: a filled _lib.Params.
    Returns (outputs (U, rows, N), [stats dict per unit])."""
    lib = _lib.load()
    dt = np.complex128 if np.dtype(precision) == np.dtype(np.complex128) else np.complex64
    fin = np.ascontiguousarray(fields, dtype=dt)
    U, rows, N = fin.shape
    fout = np.empty_like(fin)
    devs = (C.c_int32 * len(devices))(*devices)
    stats = (_lib.Stats * U)()
    rc = lib.ssf_mgpu_run(len(devices), devs, U, N, rows, _lib.SSF_C128 if dt == np.complex128 else _lib.SSF_C64,
                          {"auto": 0, "rocfft": 1, "fused": 2}[engine], C.byref(cparams),
                          fin.ctypes.data_as(C.c_void_p), fout.ctypes.data_as(C.c_void_p), stats)
    _lib.raise_for(lib, None, rc)
    return fout, [s.as_dict() for s in stats]
