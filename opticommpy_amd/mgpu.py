"""Multi-GPU sharding of INDEPENDENT fields.

A single (N, 2) field is one global FFT per transform and is never split.  What shards
is a batch of independent units -- WDM channels simulated separately, launch-power sweep
points (examples/test_NLC_withDBP_WDM_transmission.ipynb:660-675), Monte-Carlo fibre
realisations: unit u of U goes to rank u*G//U (contiguous blocks), there is no per-step
communication; the parameter block is broadcast, the inputs are scattered from the root and
the results gathered, all through RCCL (xGMI inside a node) bound inside libssf_hip.so
(include/ssf.h: ssf_comm_*).  No torch, no mpi4py.

Two ways to use the GPUs of one node:

* one process per GPU (``torchrun`` / ``mpirun`` / ``srun`` only as the process launcher): every rank
  builds ``comm = RcclComm.from_env()`` and calls :func:`run_sharded`;
* one process, host threads per device, through the C entry point ``ssf_mgpu_run``:
  :func:`run_threads` (what a one-node notebook user needs).

:func:`run_sharded` only needs an object with the small interface of :class:`RcclComm` (rank, world,
barrier, bcast, send, recv, allgather, allreduce): the CPU tests drive it with a gloo-backed stand-in.
"""
import copy
import ctypes as C
import hashlib
import os
import pickle
import stat
import tempfile
import time

import numpy as np

from . import _lib


def _process_start_time():
    """Wall-clock start of this process (seconds since the epoch); falls back to 'now'."""
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(float(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return time.time()


def shard_range(n_units, world, rank):
    """Contiguous block of unit indexes owned by `rank` (same rule as ssf_mgpu_run)."""
    return range(n_units * rank // world, n_units * (rank + 1) // world)


def owner_of(unit, n_units, world):
    for r in range(world):
        if unit in shard_range(n_units, world, r):
            return r
    raise IndexError(unit)


def _ptr(a):
    """(address, nbytes, keepalive) of a contiguous numpy array or a DeviceArray"""
    from . import device as _dev
    if _dev.is_device(a):
        return C.c_void_p(a.ptr.value if hasattr(a.ptr, "value") else a.ptr), a.nbytes, a
    if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
        raise ValueError("communication buffers must be C-contiguous numpy arrays or DeviceArrays")
    return a.ctypes.data_as(C.c_void_p), a.nbytes, a


class RcclComm:
    """One RCCL communicator over the processes of a launcher (one process per GPU), through libssf_hip.so."""

    def __init__(self, device, world, rank, comm_id):
        self.lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        h = C.c_void_p()
        rc = self.lib.ssf_comm_create(self.device, self.world, self.rank, comm_id, C.byref(h))
        if rc:
            msg = self.lib.ssf_comm_last_error(None)
            raise RuntimeError("RCCL communicator: %s" % (msg.decode(errors="replace") if msg else _lib.STATUS.get(rc, rc)))
        self.h = h

    # -- rendezvous -------------------------------------------------------------------------------------------
    # Rank 0 hands the 128-byte RCCL id to the other ranks through a file.  $SSF_RCCL_ID_FILE names it explicitly
    # (bench.py --gpus N and any launcher that can export one variable to all ranks); otherwise it lives in a per-user
    # 0700 directory and is keyed by what every rank of ONE job shares whatever started it: the rendezvous address and
    # port plus the launcher's run / job id (torchrun, srun, mpirun) -- not the parent pid, which differs per rank under
    # srun or mpirun with a daemon per rank.  The file is created with O_EXCL, mode 0600, and carries a magic word and
    # its creation time: readers ignore anything written before their own job could have started (a file a crashed run
    # left behind) and rank 0 unlinks it on every exit path.  Scope: ONE NODE.  The file lives in a local temporary directory, and
    # without $SSF_RCCL_ID_FILE or a launcher job id the key falls back to the launcher's pid, which differs between the launchers
    # of different nodes: a multi-node static rendezvous needs $SSF_RCCL_ID_FILE on a shared file system (otherwise the other
    # nodes' ranks wait for a file that never appears and fail with the rendezvous timeout).
    _MAGIC = b"SSFRCCL2"
    _STALE_S = 300.0

    @staticmethod
    def _nonce():
        """16 bytes that identify THIS job, from $SSF_RCCL_NONCE (any string the launcher exports to every rank: bench.py --gpus N
        and the tests do); zeros when there is none.  With a nonce an id file is this job's or it is not -- whenever it was
        written (a rank that starts minutes after rank 0 still takes it; a file a killed run left under the same key seconds ago
        is not taken).  Without one only freshness can be checked: the file must not be older than this rank's own start minus
        _STALE_S (advisor, round 3)."""
        n = os.environ.get("SSF_RCCL_NONCE")
        return hashlib.md5(n.encode()).digest() if n else b"\0" * 16

    @classmethod
    def _pack(cls, raw, stamp=None, nonce=None):
        return cls._MAGIC + np.array([time.time() if stamp is None else stamp], dtype=np.float64).tobytes() + \
            (cls._nonce() if nonce is None else nonce) + raw

    @staticmethod
    def id_path():
        env = os.environ
        p = env.get("SSF_RCCL_ID_FILE")
        if p:
            return p
        d = os.path.join(tempfile.gettempdir(), "ssf_rccl_%d" % os.getuid())
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.lstat(d)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError("RCCL rendezvous: %s is not a private directory of this user" % d)
        rid = env.get("TORCHELASTIC_RUN_ID")
        if rid in ("", "none"):                                # torchrun's static rendezvous: every job on the node is "none"
            rid = None
        job = rid or env.get("SLURM_JOB_ID") or env.get("OMPI_MCA_ess_base_jobid") or env.get("PMIX_NAMESPACE")
        if not job:
            # no job id (or torchrun's "none"): the ranks of one single-node launch are children of one launcher process, whose
            # pid tells this job from the one that ran a minute ago under the same MASTER_PORT (N = 1, 2, 4, 8 runs back to
            # back) and may have left a file behind
            job = "ppid%d" % os.getppid()
        key = "%s_%s_%s" % (env.get("MASTER_ADDR", "local"), env.get("MASTER_PORT", "0"), job)
        return os.path.join(d, "".join(c if c.isalnum() or c in "._-" else "_" for c in key) + ".id")

    @classmethod
    def _publish_id(cls, path, raw):
        try:
            os.unlink(path)                                    # whatever is there is not from this run
        except OSError:
            pass
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(cls._pack(raw))

    @classmethod
    def _read_id(cls, path, not_before):
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        except OSError:
            return None
        with os.fdopen(fd, "rb") as f:
            if os.fstat(f.fileno()).st_uid != os.getuid():
                raise RuntimeError("RCCL rendezvous: %s belongs to another user" % path)
            raw = f.read()
        if len(raw) != 32 + _lib.COMM_ID_BYTES or raw[:8] != cls._MAGIC:
            return None                                        # not complete yet (or not ours)
        mine = cls._nonce()
        if any(mine):                                          # a job nonce: identity, not age, decides
            return raw[32:] if raw[16:32] == mine else None
        if any(raw[16:32]) or float(np.frombuffer(raw[8:16], dtype=np.float64)[0]) < not_before:
            return None                                        # another job's, or left behind by an earlier run
        return raw[32:]

    @classmethod
    def from_env(cls, device=None, timeout=180.0):
        """RANK / WORLD_SIZE / LOCAL_RANK as torchrun, mpirun (OMPI_COMM_WORLD_*) or srun (SLURM_*) export them."""
        env = os.environ
        rank = int(env.get("RANK", env.get("OMPI_COMM_WORLD_RANK", env.get("SLURM_PROCID", "0"))))
        world = int(env.get("WORLD_SIZE", env.get("OMPI_COMM_WORLD_SIZE", env.get("SLURM_NTASKS", "1"))))
        local = int(env.get("LOCAL_RANK", env.get("OMPI_COMM_WORLD_LOCAL_RANK", env.get("SLURM_LOCALID", str(rank)))))
        lib = _lib.load()
        path = cls.id_path()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank == 0:
            rc = lib.ssf_comm_get_id(buf)
            if rc:
                msg = lib.ssf_comm_last_error(None)
                raise RuntimeError("RCCL: %s" % (msg.decode(errors="replace") if msg else rc))
            if world > 1:
                cls._publish_id(path, buf.raw)
        else:
            t0 = time.time()
            not_before = _process_start_time() - cls._STALE_S
            while True:
                raw = cls._read_id(path, not_before)
                if raw is not None:
                    break
                if time.time() - t0 > timeout:
                    raise TimeoutError("RCCL rendezvous: no fresh id in %s within %.0f s" % (path, timeout))
                time.sleep(0.02)
            buf = C.create_string_buffer(raw, _lib.COMM_ID_BYTES)
        try:
            comm = cls(local if device is None else device, world, rank, buf)  # (collective: returns on every rank together)
        finally:
            if rank == 0 and world > 1:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        return comm

    # -- collectives ----------------------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            msg = self.lib.ssf_comm_last_error(self.h)
            raise RuntimeError("RCCL: %s" % (msg.decode(errors="replace") if msg else _lib.STATUS.get(rc, rc)))

    def barrier(self):
        self._chk(self.lib.ssf_comm_barrier(self.h))

    def allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.lib.ssf_comm_allreduce(self.h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, {"sum": 0, "max": 1}[op]))
        return v

    def bcast(self, arr, root=0):
        p, n, _k = _ptr(arr)
        self._chk(self.lib.ssf_comm_bcast(self.h, p, n, root))
        return arr

    def send(self, arr, peer):
        p, n, _k = _ptr(arr)
        self._chk(self.lib.ssf_comm_send(self.h, p, n, peer))

    def recv(self, arr, peer):
        p, n, _k = _ptr(arr)
        self._chk(self.lib.ssf_comm_recv(self.h, p, n, peer))
        return arr

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        self._chk(self.lib.ssf_comm_allgather(self.h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.nbytes))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.ssf_comm_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def bcast_object(comm, obj, root=0):
    """Broadcast a small picklable object (the parameter block, launch powers): 8-byte length, then the bytes."""
    if comm is None or comm.world == 1:
        return obj
    raw = pickle.dumps(obj) if comm.rank == root else b""
    n = np.array([len(raw)], dtype=np.int64)
    comm.bcast(n, root)
    buf = np.frombuffer(raw, dtype=np.uint8).copy() if comm.rank == root else np.empty(int(n[0]), dtype=np.uint8)
    comm.bcast(buf, root)
    return obj if comm.rank == root else pickle.loads(buf.tobytes())


_LANES = {}


def _lane_pool(lanes):
    """Long-lived worker threads (their cached plans are reused from call to call)."""
    pool = _LANES.get(lanes)
    if pool is None:
        from concurrent.futures import ThreadPoolExecutor
        pool = _LANES[lanes] = ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="ssf-lane")
    return pool


def run_sharded(fields, param, compute=None, gather=True, comm=None, root=None):
    """Propagate a list of independent fields, sharded over the ranks of `comm` (all of them locally when comm is None).

    fields : sequence of U arrays (N, 2K).  root=None: identical on every rank (each rank takes its block);
             root=r: only rank r needs to hold them -- shape, dtype, the parameter block and the launch data are
             broadcast and every other rank receives the inputs of its block from r (`fields` may be None there)
    param  : parameters object (deep-copied per unit, so defaults written back by one unit never leak into another)
    compute: callable(Ei, param) -> Eout; default opticommpy_amd.manakovSSF (GPU)
    gather : True: every rank returns the full list (one all-gather); "root": only `root` (or rank 0) does, the
             others get None in the slots they do not own; False: own units only
    """
    default_compute = compute is None
    if compute is None:
        from .models import manakovSSF as compute
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    if root is not None and world > 1:
        meta = None
        if rank == root:
            f0 = np.asarray(fields[0])
            meta = (len(fields), f0.shape, f0.dtype.str, param)
        n, shape, dt, param = bcast_object(comm, meta, root)
        dt = np.dtype(dt)
        mine = shard_range(n, world, rank)
        local = {}
        for r in range(world):                                  # inputs: root -> owner, unit by unit
            for u in shard_range(n, world, r):
                if r == root:
                    if rank == root:
                        local[u] = np.ascontiguousarray(fields[u], dtype=dt)
                elif rank == root:
                    comm.send(np.ascontiguousarray(fields[u], dtype=dt), r)
                elif rank == r:
                    local[u] = comm.recv(np.empty(shape, dtype=dt), root)
    else:
        n = len(fields)
        mine = shard_range(n, world, rank)
        local = {u: fields[u] for u in mine}
    outs = [None] * n
    # Small units are latency-bound one at a time (a launch is one ~10 us chain of load -> transform -> store whatever its
    # size): a rank's units of one shape go to the device as ONE batch of independent units -- every launch carries all of
    # them, each with its own control block, step sizes and convergence decisions (ssf_plan_set_units) -- with results
    # bit-equal to one call per unit.  SSF_MGPU_BATCH=0 turns it off; fields of 2^19 samples and more fill the GPU alone.
    mine_left = list(mine)
    if default_compute and _batchable(local, mine, param):
        from .models import UnitsUnsupported
        try:
            _run_batched(local, list(mine), param, outs)
            mine_left = []
        except UnitsUnsupported:           # rocFFT / Bluestein / one-launch-row plans (set_engine('rocfft'), N = 1500, 3000, 30030 ...)
            pass                           # carry no independent units: one call per unit on the lanes, as include/ssf.h says
    # Two lanes per GPU: the rank's units are taken by two host threads (each with its own plan and stream; ctypes
    # releases the GIL inside the library), so that one unit's transfers overlap the other's kernels and the kernels of
    # two independent fields fill each other's load / store phases (ssf_mgpu_run does the same).
    lanes = max(1, min(int(os.environ.get("SSF_MGPU_LANES", "2")), len(mine)))

    def one(u):
        from . import models as _models
        _models._set_lane_hint(lanes if len(mine_left) > 1 else 1)
        p = copy.deepcopy(param)
        # a fixed param.seed keys ONE noise stream: unit u draws its own rows of it (ssf_params::rng_row_offset), so the
        # Monte-Carlo units of a seeded run get independent ASE noise, the same whatever the number of ranks
        ncols = int(np.shape(local[u])[1]) if np.ndim(local[u]) > 1 else 1
        try:
            p._rng_row_offset = int(getattr(param, "_rng_row_offset", 0)) + u * ncols
        except AttributeError:                                  # (a parameter object without settable attributes)
            pass
        out = np.asarray(compute(local[u], p))
        return u, out

    if lanes > 1 and len(mine_left) > 1:
        for u, o in _lane_pool(lanes).map(one, mine_left):
            outs[u] = o
    else:
        for u in mine_left:
            outs[u] = one(u)[1]
    if not (comm is not None and gather and world > 1):
        return outs
    probe = outs[mine[0]] if len(mine) else None
    metas = [bcast_object(comm, None if probe is None else (probe.shape, probe.dtype.str), r) for r in range(world)]
    oshape, odt = next(m for m in metas if m is not None)
    odt = np.dtype(odt)
    if gather == "root":
        dst = 0 if root is None else root
        for r in range(world):
            for u in shard_range(n, world, r):
                if r == dst:
                    continue
                if rank == r:
                    comm.send(np.ascontiguousarray(outs[u], dtype=odt), dst)
                elif rank == dst:
                    outs[u] = comm.recv(np.empty(oshape, dtype=odt), r)
        return outs
    per = max(len(shard_range(n, world, r)) for r in range(world))      # equal-sized payload per rank: pad to the largest block
    buf = np.zeros((per,) + tuple(oshape), dtype=odt)
    for i, u in enumerate(mine):
        buf[i] = outs[u]
    parts = comm.allgather(buf)                                      # the one collective of the data path
    for r in range(world):
        for i, u in enumerate(shard_range(n, world, r)):
            outs[u] = parts[r][i].astype(odt, copy=False)
    return outs


_BATCH_MAX_SAMPLES = 1 << 18        # per polarisation row: larger fields fill the GPU by themselves
_BATCH_MAX_UNITS = 64


def _batchable(local, mine, param):
    """True when the rank's units can go to the device as one batch of independent units: default compute, host
    arrays of one shape and dtype, more than one unit, small fields, final field only (or one pair per unit)."""
    from . import device as _dev
    if os.environ.get("SSF_MGPU_BATCH", "1") == "0" or len(mine) < 2:
        return False
    f0 = local[mine[0]]
    if _dev.is_device(f0) or np.ndim(f0) != 2 or np.shape(f0)[1] % 2 or np.shape(f0)[0] > _BATCH_MAX_SAMPLES:
        return False
    if getattr(param, "returnParameters", False):
        return False
    save = getattr(param, "saveSpanN", None)
    if np.shape(f0)[1] > 2 and (save is None or len(save) > 0):   # (K > 1 per unit needs saveSpanN = [], like the reference)
        return False
    return all(not _dev.is_device(local[u]) and np.shape(local[u]) == np.shape(f0) and np.asarray(local[u]).dtype == np.asarray(f0).dtype
               for u in mine)


def _run_batched(local, units, param, outs):
    from . import models as _models
    from .models import manakovSSF
    N, ncols = np.shape(local[units[0]])
    # From 2^16 samples on a batch alone no longer hides every load / store phase: two lanes, each with half of the units in
    # one plan, are faster still (16 units of 2^18, unit-steps/s: one at a time 7 905, two lanes 15 382, one batch of 8
    # 19 036, two lanes x batches of 4 23 808; tools/bench_units_large.py, profiles/r3_units_large.txt)
    lanes = max(1, min(int(os.environ.get("SSF_MGPU_LANES", "2")), len(units) // 2)) if N >= (1 << 16) else 1
    per = min(_BATCH_MAX_UNITS, (len(units) + lanes - 1) // lanes)
    chunks = [units[i:i + per] for i in range(0, len(units), per)]   # (contiguous unit numbers: the rank's block)

    def one(chunk):
        _models._set_lane_hint(lanes)
        p = copy.deepcopy(param)
        try:
            p._rng_row_offset = int(getattr(param, "_rng_row_offset", 0)) + chunk[0] * ncols
        except AttributeError:
            pass
        E = np.concatenate([np.asarray(local[u]) for u in chunk], axis=1)
        out = np.asarray(manakovSSF(E, p, _units=len(chunk)))
        nblk = out.shape[1] // (ncols * len(chunk))              # saved spans per unit (1: the final field)
        for j, u in enumerate(chunk):                            # snapshot b of the batch = columns [b * ncols_all, ...)
            cols = [b * ncols * len(chunk) + j * ncols + c for b in range(nblk) for c in range(ncols)]
            outs[u] = np.ascontiguousarray(out[:, cols])

    if lanes > 1 and len(chunks) > 1:
        list(_lane_pool(lanes).map(one, chunks))
    else:
        for chunk in chunks:
            one(chunk)


def run_coupled(Ei_block, param, comm):
    """One reference call on a (N, 2K) batch whose polarisation pairs are spread over the ranks of `comm`.

    A K > 1 batch in ONE `manakovSSF` call is coupled through max(phi) over all rows (adaptive step, reference
    optic/models/channels.py:394) and the all-row norms of convergenceCondition (channels.py:517-519).  Every rank
    passes ITS columns `Ei_block` (N, 2 K_local) and the same `param`; the engine's partial sums and maxima are
    all-reduced over `comm` before they are used (ssf_set_coupling), so every rank takes the step sizes and
    iteration counts of the single coupled call and returns its block of that call's result.  Host-driven control
    flow (general-length engine): an 8-byte all-reduce per step and a 16-byte one per iteration.  Independent units do
    not need this -- use run_sharded."""
    from .models import manakovSSF
    if comm is None or comm.world == 1:
        return manakovSSF(Ei_block, param)
    # rows of the coupled batch held by the ranks before this one: with amp='edfa' and a fixed seed every rank keys the
    # same Philox stream, and its pairs must draw THEIR rows of it (independent noise per column, like the single call)
    counts = comm.allgather(np.array([float(np.shape(Ei_block)[1])]))
    # (on top of an offset the caller may carry already -- run_sharded gives unit u the rows u * ncols -- and on a copy: the
    #  caller's object keeps what it had)
    p = copy.copy(param)
    base = int(getattr(param, "_rng_row_offset", 0))
    try:
        p._rng_row_offset = base + int(round(float(np.sum(counts[:comm.rank]))))
    except AttributeError:                 # (a parameters object with __slots__: every rank would then draw the SAME Philox rows)
        raise TypeError("run_coupled: the parameters object must accept new attributes (the rank's noise row offset is stored "
                        "on a copy of it)") from None
    out = manakovSSF(Ei_block, p, _coupling=comm)
    if p is not param:                     # the reference writes defaults back onto the caller's object (channels.py:305-322)
        for k, v in vars(p).items():
            if k != "_rng_row_offset" and not hasattr(param, k):
                try:
                    setattr(param, k, v)
                except AttributeError:
                    pass
    return out


def run_threads(fields, cparams, devices, precision=np.complex128, engine="auto"):
    """Single-process multi-GPU: host threads per device inside libssf_hip.so
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
