"""ctypes binding of libssf_hip.so (C ABI: include/ssf.h).  No fallback: if the
shared library is missing, importing a propagation function still works but the
first call raises RuntimeError telling the user to build it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSF_LIB", os.path.join(_HERE, "libssf_hip.so"))

SSF_C64, SSF_C128 = 0, 1
MODEL_NLSE, MODEL_MANAKOV = 0, 1
AMP_NONE, AMP_IDEAL, AMP_EDFA = 0, 1, 2
ENGINE_AUTO, ENGINE_ROCFFT, ENGINE_FUSED = 0, 1, 2
ENGINE_NAMES = {ENGINE_AUTO: "auto", ENGINE_ROCFFT: "rocfft", ENGINE_FUSED: "fused"}
PIPELINE_NAMES = {0: "fused-device", 1: "fused-rows", 2: "fused-bluestein", 3: "rocfft"}      # ssf_plan_pipeline

STATUS = {0: "OK", -1: "bad argument", -2: "HIP error", -3: "out of device memory", -4: "FFT error",
          -5: "no device", -6: "unsupported", -7: "bad call order", -8: "RCCL error"}


class Params(C.Structure):
    _fields_ = [("model", C.c_int32), ("direction", C.c_int32), ("Fs", C.c_double), ("Fc", C.c_double),
                ("alpha", C.c_double), ("D", C.c_double), ("gamma", C.c_double), ("Lspan", C.c_double),
                ("Nspans", C.c_int32), ("maxIter", C.c_int32), ("hz", C.c_double), ("tol", C.c_double),
                ("nlprMethod", C.c_int32), ("amp", C.c_int32), ("maxNlinPhaseRot", C.c_double),
                ("NF", C.c_double), ("n_save", C.c_int32), ("rng_row_offset", C.c_int32),
                ("save_spans", C.POINTER(C.c_int32)), ("rng_seed", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("iterations", C.c_int64), ("transforms", C.c_int64),
                ("nonconverged_steps", C.c_int64), ("device_ms", C.c_double),
                ("bytes_algorithmic", C.c_double), ("engine", C.c_int32), ("n_snapshots", C.c_int32),
                ("decided_ahead", C.c_int64), ("rebuilt_iterates", C.c_int64),
                ("recovered_fields", C.c_int64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["engine"] = ENGINE_NAMES.get(d["engine"], str(d["engine"]))
        return d


class Trace(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("count", C.c_int64), ("hz", C.POINTER(C.c_double)),
                ("iters", C.POINTER(C.c_int32)), ("lims", C.POINTER(C.c_double))]


class KernelTimes(C.Structure):
    _fields_ = [("row_ms", C.c_double), ("col_ms", C.c_double), ("other_ms", C.c_double),
                ("row_n", C.c_int64), ("col_n", C.c_int64), ("other_n", C.c_int64),
                ("col_h_ms", C.c_double), ("col_adv_ms", C.c_double), ("col_fin_ms", C.c_double),
                ("col_h_n", C.c_int64), ("col_adv_n", C.c_int64), ("col_fin_n", C.c_int64), ("outliers", C.c_int64)]


class RxParams(C.Structure):
    """ssf_rx_params (include/ssf.h)."""
    _fields_ = [("Fs", C.c_double), ("polRotation", C.c_double), ("pdl", C.c_double), ("polDelay", C.c_double),
                ("ampImb", C.c_double * 2), ("phaseImb", C.c_double * 2), ("timeSkew", C.c_double * 2),
                ("R", C.c_double), ("Tc", C.c_double), ("Id", C.c_double), ("RL", C.c_double), ("B", C.c_double),
                ("IpdSat", C.c_double), ("N", C.c_int32), ("fType", C.c_int32), ("ideal", C.c_int32),
                ("shotNoise", C.c_int32), ("thermalNoise", C.c_int32), ("currentSaturation", C.c_int32),
                ("bandwidthLimitation", C.c_int32), ("pad_", C.c_int32), ("rng_seed", C.c_int64), ("Fs_pd", C.c_double)]


class TxParams(C.Structure):
    """ssf_tx_params (include/ssf.h)."""
    _fields_ = [("Fs", C.c_double), ("mzmScale", C.c_double), ("nSymbols", C.c_int64), ("SpS", C.c_int32),
                ("nChannels", C.c_int32), ("nPolModes", C.c_int32), ("ntaps", C.c_int32),
                ("pn_sigma", C.c_double), ("pn_seed", C.c_uint64), ("phi_rows", C.c_int32), ("reserved", C.c_int32)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("reserved", C.c_int32), ("total_mem_bytes", C.c_int64), ("lds_per_block_bytes", C.c_int64)]


# every symbol include/ssf.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "ssf_device_count": (C.c_int, []),
    "ssf_device_info": (C.c_int, [C.c_int, C.POINTER(DeviceInfo)]),
    "ssf_version": (C.c_char_p, []),
    "ssf_plan_create": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ssf_plan_destroy": (C.c_int, [C.c_void_p]),
    "ssf_plan_set_units": (C.c_int, [C.c_void_p, C.c_int32]),
    "ssf_plan_pipeline": (C.c_int, [C.c_void_p]),
    "ssf_plan_set_lanes": (C.c_int, [C.c_void_p, C.c_int32]),
    "ssf_get_unit_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    "ssf_last_error": (C.c_char_p, [C.c_void_p]),
    "ssf_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssf_execute": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_int32, C.c_int32, C.c_void_p,
                              C.POINTER(Stats), C.POINTER(Trace)]),
    "ssf_download": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssf_download_snapshots": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssf_set_snapshot_sink": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ssf_sync_snapshots": (C.c_int, [C.c_void_p]),
    "ssf_upload_aos": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssf_download_aos": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ssf_run": (C.c_int, [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.POINTER(Stats), C.POINTER(Trace)]),
    "ssf_mgpu_run": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "ssf_set_coupling": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_set_coupling_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssf_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "ssf_get_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(KernelTimes)]),
    "ssf_overlap_save": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "ssf_device_malloc": (C.c_int, [C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "ssf_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "ssf_device_memcpy": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    "ssf_fir_filter": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_couple_reduce_selftest": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ssf_device_axpy": (C.c_int, [C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]),
    "ssf_fir_long": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ssf_delay_signal": (C.c_int, [C.c_int, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "ssf_edfa": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_pbs": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_optical_hybrid_2x4": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_nlin_phase_rot": (C.c_int, [C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ssf_convergence_condition": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_double)]),
    "ssf_decimate": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_int32)]),
    "ssf_rx_run": (C.c_int, [C.c_int, C.c_int32, C.c_int64, C.c_int32, C.POINTER(RxParams), C.c_void_p, C.c_void_p,
                             C.POINTER(C.c_double), C.c_void_p]),
    "ssf_rx_chain": (C.c_int, [C.c_int, C.c_int64, C.POINTER(RxParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]),
    "ssf_wdm_tx": (C.c_int, [C.c_int, C.POINTER(TxParams), C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                             C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double)]),
    "ssf_device_copy_bandwidth": (C.c_int, [C.c_int, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    "ssf_linear_channel": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p]),
    "ssf_comm_get_id": (C.c_int, [C.c_void_p]),
    "ssf_comm_create": (C.c_int, [C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ssf_comm_destroy": (C.c_int, [C.c_void_p]),
    "ssf_comm_rank": (C.c_int, [C.c_void_p]),
    "ssf_comm_size": (C.c_int, [C.c_void_p]),
    "ssf_comm_barrier": (C.c_int, [C.c_void_p]),
    "ssf_comm_allreduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "ssf_comm_bcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ssf_comm_send": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ssf_comm_recv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    "ssf_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "ssf_comm_last_error": (C.c_char_p, [C.c_void_p]),
}
COMM_ID_BYTES = 128
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)     # ssf_reduce_fn

_lib = None


def load():
    """Load libssf_hip.so and bind every ABI symbol.  Raises RuntimeError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C opticommpy_amd/csrc`. "
            "opticommpy_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)        # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def error_message(lib, handle, rc):
    msg = lib.ssf_last_error(handle)
    msg = msg.decode(errors="replace") if msg else ""
    return f"{STATUS.get(rc, rc)}: {msg}" if msg else str(STATUS.get(rc, rc))


def raise_for(lib, handle, rc):
    if rc == 0:
        return
    text = error_message(lib, handle, rc)
    if rc == -1:
        raise ValueError(text)
    if rc == -3:
        raise MemoryError(text)
    raise RuntimeError(text)
