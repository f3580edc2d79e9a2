#!/usr/bin/env python3
"""bench.py -- SSFM steps/s on MI355X for BASELINE.json's headline configuration.

Workload (config 2, SURVEY.md 8d "C2"): dual-pol manakovSSF, N = 2^20 complex128
samples, Fs 512 GS/s, 8.4 dBm band-limited Gaussian field (seed 2), alpha 0.2,
D 16, gamma 1.3, hz 0.08 km fixed step, maxIter 10, tol 1e-5, amp 'ideal',
saveSpanN = [].  One "step" = one pass of `while z_current < Lspan`
(reference optic/models/channels.py:387).  The timed region runs EXACTLY K
steps with the field already resident in HBM (upload before, download after).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU;
every rank propagates its own independent field (weak scaling, no data-path
collective -- SURVEY.md 8e); torch.distributed (RCCL) is used only for the
barriers and the max-over-ranks reduction of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_field(N, ncols, seed, p_dbm, dtype=np.complex128):
    rng = np.random.default_rng(seed)
    E = (rng.normal(size=(N, ncols)) + 1j * rng.normal(size=(N, ncols))) / np.sqrt(2)
    F = np.fft.fft(E, axis=0)
    F[N // 4: 3 * N // 4, :] = 0
    E = np.fft.ifft(F, axis=0)
    p_lin = 10 ** (p_dbm / 10) * 1e-3
    E = E * np.sqrt((p_lin / 2) / np.mean(np.abs(E) ** 2, axis=0))
    return E.astype(dtype)


def make_params(lib_mod, steps, hz, prec_fs=512e9):
    cp = lib_mod.Params()
    cp.model, cp.direction = lib_mod.MODEL_MANAKOV, 1
    cp.Fs, cp.Fc, cp.alpha, cp.D, cp.gamma = prec_fs, 193.1e12, 0.2, 16.0, 1.3
    cp.Lspan = (steps - 0.5) * hz          # exactly `steps` passes of the while loop (last one is half a step)
    cp.Nspans, cp.hz, cp.maxIter, cp.tol = 1, hz, 10, 1e-5
    cp.nlprMethod, cp.maxNlinPhaseRot, cp.amp, cp.NF = 0, 2e-2, lib_mod.AMP_IDEAL, 4.5
    cp.n_save, cp.save_spans = 0, None
    return cp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--prec", default="c128", choices=["c128", "c64"])
    ap.add_argument("--engine", default=os.environ.get("SSF_ENGINE", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=16)
    ap.add_argument("--no-kernel-times", action="store_true", help="skip the per-kernel HIP-event pass")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("SSF_BENCH_FORCE_DIST"):      # (the env var exercises the RCCL path on one GPU)
        import torch
        import torch.distributed as dist
        # (test knobs: SSF_BENCH_BACKEND=gloo + SSF_BENCH_DEVICE=0 run several ranks against one GPU, which RCCL
        # refuses; the driver's runs use neither)
        backend = os.environ.get("SSF_BENCH_BACKEND", "nccl")
        if "SSF_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["SSF_BENCH_DEVICE"])
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        tdev = "cuda" if backend == "nccl" else "cpu"

    from opticommpy_amd import _lib
    lib = _lib.load()
    if lib.ssf_device_count() <= 0:
        raise SystemExit("bench.py needs a GPU: no HIP device visible (there is no CPU fallback)")

    N = 1 << args.log2n
    dtype = np.complex128 if args.prec == "c128" else np.complex64
    prec = _lib.SSF_C128 if args.prec == "c128" else _lib.SSF_C64
    engine = {"auto": 0, "rocfft": 1, "fused": 2}[args.engine]
    E = synth_field(N, 2, 2 + rank, 8.4, dtype)
    soa = np.ascontiguousarray(E.T)

    h = C.c_void_p()
    _lib.raise_for(lib, None, lib.ssf_plan_create(local_rank, N, 2, prec, engine, C.byref(h)))

    def run(steps, field, sync=True):
        # sync: every rank makes this call (the timed run and its warm-up); the rank-0-only passes further down
        # must not enter a barrier the other ranks never reach
        st = _lib.Stats()
        cp = make_params(_lib, steps, 0.08)
        _lib.raise_for(lib, h, lib.ssf_upload(h, field.ctypes.data_as(C.c_void_p)))      # field resident in HBM
        if dist is not None and sync:
            dist.barrier()
        t0 = time.perf_counter()
        rc = lib.ssf_execute(h, C.byref(cp), 1, 1, None, C.byref(st), None)              # synchronous at return
        t1 = time.perf_counter()
        _lib.raise_for(lib, h, rc)
        return t1 - t0, st

    if args.warmup > 0:
        run(args.warmup, soa)
    dt, st = run(args.steps, soa)
    assert st.steps == args.steps, (st.steps, args.steps)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()

    out_soa = np.empty_like(soa)
    _lib.raise_for(lib, h, lib.ssf_download(h, out_soa.ctypes.data_as(C.c_void_p)))
    if dist is not None:
        # "gather of results only" (SURVEY.md 8e): one RCCL all-gather of a per-rank checksum
        import torch
        cs = torch.tensor([float(np.sum(np.abs(out_soa) ** 2))], dtype=torch.float64, device=tdev)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        checksums = [float(x.item()) for x in allcs]

    # per-kernel timing pass (HIP events around every launch on the plan stream; separate from the
    # headline run because the events themselves cost a few microseconds per launch)
    kernels = None
    if rank == 0 and not args.no_kernel_times and lib.ssf_set_profiling(h, 1) == 0:
        nprof = min(args.steps, 200)
        _, stp = run(nprof, soa, sync=False)
        kt = _lib.KernelTimes()
        lib.ssf_get_kernel_times(h, C.byref(kt))
        lib.ssf_set_profiling(h, 0)
        bytes_per_launch = 2 * (16 if args.prec == "c128" else 8) * N * 2       # one transform-equivalent per row
        kernels = {}
        for name, ms, n in (("row (decision + FFT.H.IFFT of rows)", kt.row_ms, kt.row_n),
                            ("col (Manakov column stage: S | H | I)", kt.col_ms, kt.col_n)):
            if n:
                avg_us = ms / n * 1e3
                kernels[name] = {"launches": int(n), "avg_us": avg_us}
                if name.startswith("row"):
                    kernels[name].update(algorithmic_bytes_per_launch=bytes_per_launch,
                                         achieved_GBs=bytes_per_launch / (avg_us * 1e-6) / 1e9,
                                         frac=bytes_per_launch / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        kernels["profiled_steps"] = int(stp.steps)

    if rank == 0:
        s = 16 if args.prec == "c128" else 8
        steps_total = args.steps * world
        value = steps_total / dt
        dev_s = st.device_ms * 1e-3
        achieved = st.bytes_algorithmic / dev_s / 1e9
        rec = {
            "metric": "SSFM steps/sec (2-pol, 2^%d samples)" % args.log2n, "value": value, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.prec == "c128" else "f32", "data": "synthetic",
            "config": {"workload": "manakovSSF config-2: 2-pol N=2^%d %s, hz=0.08 km fixed, 8.4 dBm, amp=ideal, "
                                   "1 independent field per GPU" % (args.log2n, "complex128" if s == 16 else "complex64"),
                       "engine": _lib.ENGINE_NAMES[st.engine], "fields_per_gpu": 1,
                       "iterations_per_step": st.iterations / st.steps,
                       "transforms_per_step": st.transforms / st.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "whole step pipeline (HIP events around the K timed steps on the plan stream)",
                         "algorithmic_bytes_per_step": st.bytes_algorithmic / st.steps,
                         "device_ms_per_step": st.device_ms / st.steps,
                         "kernels": kernels},
        }
        # measured memory ceiling on this box: a kernel with the row stage's memory shape and no arithmetic
        # (32 MiB in + 32 MiB out per launch at N = 2^20 c128), SURVEY.md 8d
        probe = C.c_double(0.0)
        probe_bytes = max(65536, (2 * s * N) // 65536 * 65536)
        if lib.ssf_device_copy_bandwidth(local_rank, probe_bytes, 50, C.byref(probe)) == 0:
            rec["roofline"]["measured_copy_GBs"] = probe.value
            rec["roofline"]["frac_of_measured_copy"] = achieved / probe.value
            rec["roofline"]["measured_copy_note"] = ("burst copy kernel, %d MiB read + %d MiB written per launch, "
                                                     "no arithmetic" % (probe_bytes >> 20, probe_bytes >> 20))
        if dist is not None:
            rec["rank_checksums"] = checksums
        traffic_file = os.path.join(ROOT, "profiles", "traffic_bytes_per_step.json")
        if os.path.exists(traffic_file):
            try:
                t = json.load(open(traffic_file)).get(_lib.ENGINE_NAMES[st.engine])
                if t and args.log2n == 20 and args.prec == "c128":
                    # PMC-measured HBM-side bytes per step (separate rocprofv3 passes, see profiles/), scaled from
                    # the profiled iteration count to this run's: traffic is linear in (1 + iterations/step)
                    scale = (1.0 + st.iterations / st.steps) / (1.0 + t["iterations_per_step"])
                    rec["roofline"]["traffic"] = t["bytes_per_step"] * scale
                    rec["roofline"]["traffic_source"] = "profiles/traffic_bytes_per_step.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %d steps at %.2f it/step)" % (t["steps"], t["iterations_per_step"])
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:
            from oracle import ssf_oracle as orc
            n = max(2, args.cpu_steps)
            p = orc.parameters()
            p.Fs, p.Fc, p.alpha, p.D, p.gamma = 512e9, 193.1e12, 0.2, 16, 1.3
            p.Ltotal = p.Lspan = (n - 0.5) * 0.08
            p.hz, p.maxIter, p.tol, p.nlprMethod, p.amp, p.saveSpanN, p.prgsBar = 0.08, 10, 1e-5, False, "ideal", [], False
            p.prec = dtype
            tr = {}
            t0 = time.perf_counter()
            ref = orc.manakovSSF(E, p, trace=tr)
            tc = time.perf_counter() - t0
            # same n steps on the GPU for the in-run parity gate
            _, stp = run(n, soa, sync=False)
            got = np.empty_like(soa)
            _lib.raise_for(lib, h, lib.ssf_download(h, got.ctypes.data_as(C.c_void_p)))
            err = float(np.linalg.norm(got.T.astype(np.complex128) - ref) / np.linalg.norm(ref))
            cpu_model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            rec["cpu_baseline"] = {"value": n / tc, "unit": "steps/s", "cores": 1, "kind": "port",
                                   "sample": "%d steps of the same config-2 field (numpy oracle, single thread; "
                                             "%d cores available; %s; numpy %s)" % (n, os.cpu_count(), cpu_model, np.__version__),
                                   "iterations_per_step": tr["iterations"] / tr["steps"]}
            rec["parity"] = {"rel_l2_vs_oracle": err, "steps": n,
                             "iterations_gpu": int(stp.iterations), "iterations_oracle": int(tr["iterations"]),
                             "gate": 1e-10 if s == 16 else 5e-4, "ok": bool(err <= (1e-10 if s == 16 else 5e-4))}
            rec["speedup_vs_cpu"] = value / (n / tc)
        print(json.dumps(rec))
    lib.ssf_plan_destroy(h)
    if dist is not None:
        dist.barrier()                      # rank 0 arrives after its extra passes; nobody tears the group down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
