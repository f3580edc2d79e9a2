/* ssf.h -- C ABI of the MI355X-native split-step Fourier fiber-propagation library
 * (libssf_hip.so).  POD types only, no C++ exceptions cross this boundary, every
 * entry point returns an ssf_status (0 = OK, negative = error) unless stated.
 *
 * What each entry point replaces in the reference (paths relative to the
 * OptiCommPy checkout; the reference has no FFI, its "plugin API" is the set of
 * Python functions a notebook imports from optic.models.modelsGPU):
 *
 *   ssf_device_count / ssf_device_info   optic/dsp/coreGPU.py:11-24   (checkGPU)
 *   ssf_plan_create / ssf_plan_destroy   implicit cupy allocations + cuFFT plan cache,
 *                                        optic/models/modelsGPU.py:404,450-451 (manakovSSF)
 *   ssf_upload                           cp.asarray(Ei).astype(prec)      modelsGPU.py:216,404
 *   ssf_execute  (model NLSE)            ssfm hot loop                    modelsGPU.py:241-269
 *                                        == optic/models/channels.py:215-238
 *   ssf_execute  (model MANAKOV, +1)     manakovSSF span/step/iteration loops
 *                                        modelsGPU.py:420-498 == channels.py:380-456
 *   ssf_execute  (model MANAKOV, -1)     manakovDBP                       modelsGPU.py:564-772
 *                                        == optic/dsp/equalization.py:1087-1160
 *   ssf_download                         cp.asnumpy(...)                  modelsGPU.py:271,501-509
 *   ssf_set_snapshot_sink / ssf_sync_snapshots   Ech_spans[:, 2*indRecSpan:...] = ...   channels.py:453-456
 *   ssf_run                              one whole reference call (upload+execute+download)
 *   ssf_mgpu_run                         (no reference equivalent) independent fields
 *                                        sharded over the GPUs of one node, SURVEY.md 8e
 *   ssf_plan_set_units                   (no reference equivalent) several independent fields per launch
 *   ssf_plan_set_lanes                   (no reference equivalent) this plan shares the GPU with others
 *   ssf_set_coupling[_comm]              np.max(phiRot) / scipy.linalg.norm over ALL rows of a K > 1 batch
 *                                        (channels.py:394, 517-519) when the rows live in several plans
 *   ssf_couple_reduce_selftest           (test aid) the rank-order reduction behind ssf_set_coupling_comm
 *   ssf_comm_*                           (no reference equivalent) one process per GPU: RCCL
 *                                        broadcast / scatter / gather of parameters, inputs and
 *                                        results of independent units, SURVEY.md 8e
 *   ssf_linear_channel                   linearFiberChannel               channels.py:30-109
 *   ssf_overlap_save                     blockwiseFFTConv as used by edc  optic/dsp/core.py:973-1046,
 *                                                                          optic/dsp/equalization.py:113-117
 *   ssf_nlin_phase_rot                   nlinPhaseRot                     channels.py:471-493
 *   ssf_convergence_condition            convergenceCondition             channels.py:496-519
 *   ssf_fir_filter / ssf_fir_long / ssf_delay_signal / ssf_decimate / ssf_rx_run   receiver side, see below
 *   ssf_device_malloc / ssf_device_free / ssf_device_memcpy / ssf_device_axpy   device-resident arrays, see below
 *   ssf_wdm_tx                           simpleWDMTx signal path          optic/models/tx.py:178-217
 *   ssf_device_copy_bandwidth            (no reference equivalent) measured memory ceiling
 *   ssf_set_profiling / ssf_get_kernel_times   time.time() pairs around calls in
 *                                        examples/benchmarck_GPU_processing.ipynb:389-395
 *
 * Data layout at the boundary: struct-of-arrays.  A "row" is one column of the
 * reference's (N, ncols) field, stored contiguously: rows = [x0, y0, x1, y1, ...]
 * for the Manakov model (nrows = 2K), or independent scalar fields for NLSE.
 * Complex samples are interleaved (re, im) float (SSF_C64) or double (SSF_C128).
 *
 * Ownership: the caller owns every host buffer it passes, for the duration of
 * the call only.  The library owns all device memory, FFT plans and its HIP
 * stream inside the opaque plan handle.  A plan is NOT thread-safe: use one
 * plan per (thread, device).  All calls are synchronous at return.
 */
#ifndef SSF_H
#define SSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssf_plan ssf_plan;

typedef enum {
    SSF_OK = 0,
    SSF_ERR_BAD_ARG = -1,
    SSF_ERR_HIP = -2,
    SSF_ERR_OOM = -3,
    SSF_ERR_FFT = -4,
    SSF_ERR_NO_DEVICE = -5,
    SSF_ERR_UNSUPPORTED = -6,
    SSF_ERR_STATE = -7,
    SSF_ERR_COMM = -8          /* RCCL: library missing or a collective failed                */
} ssf_status;

enum { SSF_C64 = 0, SSF_C128 = 1 };                       /* field precision            */
enum { SSF_MODEL_NLSE = 0, SSF_MODEL_MANAKOV = 1 };       /* ssfm | manakovSSF/DBP      */
enum { SSF_AMP_NONE = 0, SSF_AMP_IDEAL = 1, SSF_AMP_EDFA = 2 };
enum { SSF_ENGINE_AUTO = 0, SSF_ENGINE_ROCFFT = 1, SSF_ENGINE_FUSED = 2 };

/* Raw physical parameters exactly as the reference's parameters object holds them
 * (channels.py:305-322); every derived constant (alpha [1/km], beta2, lambda,
 * G_lin, nsp, p_noise) is computed inside the library with scipy's literals
 * c = 299792458.0, h = 6.62607015e-34. */
typedef struct {
    int32_t model;            /* SSF_MODEL_*                                             */
    int32_t direction;        /* +1 forward (ssfm, manakovSSF); -1 back-propagation      */
    double  Fs;               /* sampling rate [Hz]                                      */
    double  Fc;               /* carrier [Hz]                                            */
    double  alpha;            /* [dB/km]                                                 */
    double  D;                /* [ps/nm/km]                                              */
    double  gamma;            /* [1/W/km]                                                */
    double  Lspan;            /* [km]                                                    */
    int32_t Nspans;           /* floor(Ltotal/Lspan)                                     */
    int32_t maxIter;          /* Manakov: max fixed-point iterations per step            */
    double  hz;               /* [km] fixed step (also the cap-less nominal step)        */
    double  tol;              /* Manakov: convergence tolerance                          */
    int32_t nlprMethod;       /* Manakov: 1 = adaptive step from max nonlinear phase     */
    int32_t amp;              /* SSF_AMP_*                                               */
    double  maxNlinPhaseRot;  /* [rad]                                                   */
    double  NF;               /* EDFA noise figure [dB]                                  */
    int32_t n_save;           /* number of entries of save_spans (0 = final field only)  */
    int32_t rng_row_offset;   /* device ASE noise: row r of this plan draws stream row      */
                              /* rng_row_offset + r (ranks holding parts of one coupled     */
                              /* batch, Monte-Carlo units sharing a seed: distinct offsets  */
                              /* give independent noise); 0 for a stand-alone call          */
    const int32_t *save_spans;/* 1-based span indexes to snapshot (reference saveSpanN)  */
    int64_t rng_seed;         /* amp == EDFA and no host noise given: != 0 -> ASE noise is */
                              /* generated on the device (Philox4x32-10 keyed by rng_seed, */
                              /* counter = sample / rng_row_offset + row / span); 0 -> gain */
                              /* only                                                       */
} ssf_params;

typedef struct {
    int64_t steps;               /* passes of `while z_current < Lspan` / inner `for`    */
    int64_t iterations;          /* Manakov fixed-point iterations, summed               */
    int64_t transforms;          /* length-N FFT/IFFTs of one row, summed                */
    int64_t nonconverged_steps;  /* steps that hit maxIter (reference logs a warning)    */
    double  device_ms;           /* HIP-event time of the propagation on the plan stream */
    double  bytes_algorithmic;   /* transforms * 2 * sizeof(complex) * N      */
    int32_t engine;              /* SSF_ENGINE_* actually used                           */
    int32_t n_snapshots;         /* snapshots captured so far                            */
    int64_t decided_ahead;       /* fused engine: iterations whose convergence test was  */
                                 /*   evaluated one iteration in advance (all but the    */
                                 /*   first of every step)                               */
    int64_t rebuilt_iterates;    /* fused engine: iterates rebuilt as final (lim_0<tol)  */
    int64_t recovered_fields;    /* fused engine: step-start fields recovered from E_hd  */
                                 /*   for an exact lim_0 (three launches each)           */
} ssf_stats;

/* Optional per-step trace (for parity checks of the data-dependent control flow).
 * Arrays are caller-owned; the library writes at most `capacity` steps. */
typedef struct {
    int64_t  capacity;        /* in: number of steps the arrays can hold                 */
    int64_t  count;           /* out: steps written                                      */
    double  *hz;              /* [capacity]            step size of each step            */
    int32_t *iters;           /* [capacity]            iterations of each step           */
    double  *lims;            /* [capacity * maxIter]  convergence values (NaN = unused) */
} ssf_trace;

typedef struct {
    char     name[128];
    char     arch[32];           /* e.g. "gfx950"                                        */
    int32_t  compute_units;
    int32_t  reserved;
    int64_t  total_mem_bytes;
    int64_t  lds_per_block_bytes;
} ssf_device_info_t;

/* ---- discovery --------------------------------------------------------------------- */
int  ssf_device_count(void);                                /* >= 0, or negative status */
int  ssf_device_info(int device, ssf_device_info_t *out);
const char *ssf_version(void);

/* ---- plan life cycle ---------------------------------------------------------------- */
/* N samples per row, nrows rows (2K for Manakov), precision SSF_C64/SSF_C128,
 * engine SSF_ENGINE_AUTO picks the fused radix-2^n pipeline when N is a supported
 * power of two and the rocFFT pipeline otherwise. */
int  ssf_plan_create(int device, int64_t N, int32_t nrows, int32_t precision,
                     int32_t engine, ssf_plan **out);
int  ssf_plan_destroy(ssf_plan *plan);
const char *ssf_last_error(const ssf_plan *plan);           /* never NULL; plan may be NULL */

/* ---- staged execution (field stays resident in HBM between calls) ------------------- */
int  ssf_upload(ssf_plan *plan, const void *field_soa);     /* (nrows, N) complex        */
/* Propagate spans [span_first, span_last] (1-based, inclusive) of `params`.
 * noise: NULL, or host array (span_last-span_first+1, nrows, N) complex added after
 *        the span gain when amp == SSF_AMP_EDFA (row r of span s at
 *        ((s-span_first)*nrows + r)*N); NULL with EDFA = device-generated noise when
 *        params->rng_seed != 0, gain only otherwise.
 * stats/trace may be NULL; stats accumulate until ssf_upload resets them. */
int  ssf_execute(ssf_plan *plan, const ssf_params *params, int32_t span_first,
                 int32_t span_last, const void *noise, ssf_stats *stats, ssf_trace *trace);
int  ssf_download(ssf_plan *plan, void *field_soa);         /* (nrows, N) complex        */
int  ssf_download_snapshots(ssf_plan *plan, void *snap_soa);/* (n_snapshots, nrows, N)   */
/* Streamed snapshots.  With a sink set, the snapshots of the following ssf_execute calls are
 * not kept in device memory: capture number i (i = first_index, first_index + 1, ... in capture order) of the
 * (nrows, N) field is written to columns [i * nrows, (i + 1) * nrows) of the row-major (N, ld) complex array at
 * `dst` -- the reference's Ech_spans[:, 2 * indRecSpan : 2 * indRecSpan + 2] (optic/models/channels.py:453-456,
 * modelsGPU.py:497-498).  dst on the device: written in place by the conversion kernel.  dst on the host: moved by
 * a copy thread on its own stream while the next span propagates; ssf_sync_snapshots returns when every captured
 * snapshot has arrived (ssf_execute does not wait for them).  dst = NULL detaches the sink (and waits). */
int  ssf_set_snapshot_sink(ssf_plan *plan, void *dst, int64_t ld, int32_t first_index);
int  ssf_sync_snapshots(ssf_plan *plan);
/* Same transfers in the reference's own array layout: (N, nrows) row-major, i.e. the C-contiguous
 * numpy field with columns [x0, y0, x1, y1, ...]; the AoS <-> SoA conversion runs on the device
 * (Ei_[:, 0::2].T / Ech[:, 0::2] = Ech_x.T in the reference, modelsGPU.py:406-407, 506-509).
 * which = -1: the current field; which >= 0: snapshot number `which`. */
int  ssf_upload_aos(ssf_plan *plan, const void *field_aos);
int  ssf_download_aos(ssf_plan *plan, int32_t which, void *field_aos);

/* ---- one-shot: upload + execute all spans + download -------------------------------- */
int  ssf_run(ssf_plan *plan, const ssf_params *params, const void *field_in_soa,
             void *field_out_soa, void *snapshots_soa, const void *noise,
             ssf_stats *stats, ssf_trace *trace);

/* ---- multi-GPU: independent units sharded over devices (one host thread per device) -- */
/* n_units fields of (rows_per_unit, N) each, contiguous in in/out; unit u runs on
 * device dev_ids[u * n_dev / n_units ...] (contiguous blocks).  stats: [n_units]. */
int  ssf_mgpu_run(int32_t n_dev, const int32_t *dev_ids, int32_t n_units, int64_t N,
                  int32_t rows_per_unit, int32_t precision, int32_t engine,
                  const ssf_params *params, const void *fields_in, void *fields_out,
                  ssf_stats *stats);

/* ---- multi-GPU, one process per GPU: RCCL over xGMI ------------------------------------------
 * The same partitioning as ssf_mgpu_run (unit u of U belongs to rank u*G/U ... contiguous blocks), for launchers that
 * start one process per GPU (torchrun, mpirun, srun).  Only inputs and results of INDEPENDENT units and the
 * parameter block cross the links; there is no per-step communication.  librccl.so is opened with
 * dlopen by the first ssf_comm_get_id / ssf_comm_create call; without it these calls return SSF_ERR_COMM and
 * everything else keeps working.  Every buffer may be a host pointer (staged through device memory inside the
 * library) or a device pointer of ssf_device_malloc.  All calls are collective where RCCL's are, blocking at return.
 *   ssf_comm_get_id     rank 0 draws the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other
 *                       ranks by any out-of-band means (opticommpy_amd.mgpu: a file next to the launcher's port)
 *   ssf_comm_create     ncclCommInitRank on `device`
 *   ssf_comm_bcast      parameter block / launch powers                      (ncclBroadcast, in place)
 *   ssf_comm_send/recv  inputs from the root to the owner of a unit, results back   (ncclSend / ncclRecv)
 *   ssf_comm_allgather  results of every rank's block to every rank          (ncclAllGather)
 *   ssf_comm_allreduce  op 0 = sum, 1 = max, in place: timings and checksums (ncclAllReduce)
 *   ssf_comm_barrier    an 8-byte all-reduce */
typedef struct ssf_comm ssf_comm;
#define SSF_COMM_ID_BYTES 128
int  ssf_comm_get_id(void *id);
int  ssf_comm_create(int device, int32_t nranks, int32_t rank, const void *id, ssf_comm **out);
int  ssf_comm_destroy(ssf_comm *comm);
int  ssf_comm_rank(const ssf_comm *comm);
int  ssf_comm_size(const ssf_comm *comm);
int  ssf_comm_barrier(ssf_comm *comm);
int  ssf_comm_allreduce(ssf_comm *comm, double *values, int32_t n, int32_t op);
int  ssf_comm_bcast(ssf_comm *comm, void *buf, int64_t bytes, int32_t root);
int  ssf_comm_send(ssf_comm *comm, const void *buf, int64_t bytes, int32_t peer);
int  ssf_comm_recv(ssf_comm *comm, void *buf, int64_t bytes, int32_t peer);
int  ssf_comm_allgather(ssf_comm *comm, const void *send, void *recv, int64_t bytes_per_rank);
const char *ssf_comm_last_error(const ssf_comm *comm);       /* never NULL; comm may be NULL */

/* ---- coupled batch across plans (the caveat row) ------------------------------------------------
 * A (N, 2K) batch passed to ONE reference call is coupled: the adaptive step uses max(phi) over ALL rows
 * (optic/models/channels.py:394) and convergenceCondition the Frobenius norms over ALL rows (channels.py:517-519).  To
 * reproduce that call with the pairs spread over several plans (GPUs, processes), give each plan a reducer: the
 * general-length engine (host-driven control flow: plans created with SSF_ENGINE_ROCFFT) hands it the partial results it
 * is about to use -- op 0: values[0..n) are sums (sum |E_fd - E_conv|^2, sum |E_conv|^2), op 1: maxima (max phi) -- and
 * continues with what the reducer leaves in `values`: with an all-reduce over the participating plans every plan takes
 * the step sizes and iteration counts of the single coupled call.  reduce = NULL detaches.  Returns SSF_ERR_UNSUPPORTED
 * on the fused engine (its control flow lives on the device; independent units need no coupling). */
typedef int (*ssf_reduce_fn)(void *ctx, double *values, int32_t n, int32_t op);
int  ssf_set_coupling(ssf_plan *plan, ssf_reduce_fn reduce, void *ctx);
/* The same coupling for the device-resident pipeline (SSF_PIPE_DEVICE: natively split lengths on the fused engine), host out
 * of the loop: between the column launch that leaves the partial sums / maxima and the row launch that uses them the plan's stream
 * carries one ncclAllGather of 40 bytes over `comm` and every rank reduces the gathered values in rank order (identical bits, hence
 * identical step sizes and iteration counts, on every rank).  Every rank of `comm` must run the same call with its own pairs.
 * comm = NULL detaches.  SSF_ERR_UNSUPPORTED on the host-driven pipelines (use ssf_set_coupling there) and for plans of
 * independent units; SSF_ERR_BAD_ARG if the communicator lives on another device. */
int  ssf_set_coupling_comm(ssf_plan *plan, ssf_comm *comm);

/* ---- which pipeline a plan runs on (ssf_stats.engine says SSF_ENGINE_FUSED for everything on the hand-written kernels) ----
 *   SSF_PIPE_DEVICE     natively split length: device-resident control flow, no host synchronisation inside a span
 *   SSF_PIPE_ROWS       short 2^a 3^b 5^c length: host-driven step loop, FFT . H . IFFT of a row in one LDS launch
 *   SSF_PIPE_BLUESTEIN  any other length: host-driven step loop, every transform a Bluestein convolution on the fused kernels
 *   SSF_PIPE_ROCFFT     host-driven step loop on rocFFT transforms (the cross-check engine)
 * The host-driven ones read 16 bytes back per iteration: a bench or roofline record should not file them under the
 * device-resident pipeline.  Returns the code (>= 0) or a negative status. */
enum { SSF_PIPE_DEVICE = 0, SSF_PIPE_ROWS = 1, SSF_PIPE_BLUESTEIN = 2, SSF_PIPE_ROCFFT = 3 };
int  ssf_plan_pipeline(const ssf_plan *plan);

/* ---- independent units in one plan (no reference equivalent: the reference runs one field per call) ----------------
 * Small fields are latency-bound one at a time: a launch is one chain of load -> transform -> store of ~10 us whatever its
 * size.  ssf_plan_set_units(plan, n) declares the plan's rows to be n independent fields ("units") of nrows / n rows each,
 * rows [u * nrows / n, (u + 1) * nrows / n) = unit u (an even number of rows per unit for the Manakov models).  Every
 * launch of ssf_execute then carries all units (grid.y = n), and every unit keeps its OWN device-resident control block,
 * partial sums, step sizes and convergence decisions: the result is bit-equal to n separate plans, unlike one coupled
 * K > 1 call (which shares max(phi) and the norms over all rows, channels.py:394, 517-519).  ssf_stats counts are sums
 * over the units; traces are not recorded (SSF_ERR_BAD_ARG with a trace).  Device noise: row r draws stream row
 * rng_row_offset + r, i.e. unit u sees what a stand-alone call with rng_row_offset + u * nrows / n would.  Call it before
 * ssf_upload (the engine is rebuilt, an uploaded field is dropped).  Natively split lengths of the fused engine only
 * (SSF_ERR_UNSUPPORTED otherwise: the caller falls back to one call per unit). */
int  ssf_plan_set_units(ssf_plan *plan, int32_t n_units);
/* steps / iterations / transforms / nonconverged_steps / decided_ahead / rebuilt_iterates / recovered_fields of ONE unit since the last upload
 * (the Manakov models; the other fields of *out are those of the plan). */
int  ssf_get_unit_stats(ssf_plan *plan, int32_t unit, ssf_stats *out);

/* ---- lanes: several plans sharing one GPU concurrently (no reference equivalent) ------------------------------------
 * ssf_mgpu_run / mgpu.run_sharded / bench.py --config 4 | 5 keep two plans (own stream, own host thread) busy per GPU so that one
 * unit's launches fill the other's load / store phases.  A hint that this plan is one of n_lanes such plans: the kernels
 * then leave the issue priorities alone (with one field on the GPU, priority by phase is worth +3 %; with two fields
 * interleaving it costs 4 %: profiles/r3_lanes_prio_wt.txt).  Default 1; may be called at any time between executes. */
int  ssf_plan_set_lanes(ssf_plan *plan, int32_t n_lanes);

/* ---- the device-side reduction of a coupled batch by itself (test aid) ------------------------
 * ssf_set_coupling_comm reduces every rank's per-workgroup partials on the device, all-gathers five doubles per rank and reduces
 * those in rank order, so that every rank takes identical decisions (channels.py:394, 517-519 evaluated over ALL rows).  This
 * entry runs exactly that arithmetic on synthetic partials, without a communicator: `parts` = [nranks][5][npart] doubles, per rank
 * the arrays (max phi, sum lim_i numerator, denominator, lim_0 numerator, denominator) of npart workgroups; out5 = (sum num0, sum
 * den0, sum num, sum den, max) as the row stage of every rank would read them. */
int  ssf_couple_reduce_selftest(int device, int32_t nranks, int32_t npart, const double *parts, double *out5);

/* ---- per-kernel timing (measurement aid; fused engine only) ------------------------------ */
/* With profiling enabled every kernel launch of ssf_execute is bracketed by HIP events on the
 * plan's stream (costs a few microseconds per launch: do not enable for the headline timing).
 * Totals accumulate until ssf_upload.  Kernel classes of the fused pipeline:
 *   row   = (convergence decision,) FFT_rows . H . IFFT_rows   (one transform-equivalent per row)
 *   col   = Manakov column stage: inverse/forward column FFTs around the time-domain work
 *   other = scalar-NLSE / linear-channel column stages
 * The Manakov column stage runs as stage-specialised kernels where the field fills the chip (H: half-dispersed field out + first
 * rotation; ADV: a non-final iterate -> the next one; FIN: the final iterate, observed and stored; launches whose stage the state
 * did not ask for do nothing and are counted where they were enqueued); col_* are included in col_ms / col_n, which also hold
 * the general kernel's launches (span start, short fields, ragged tiles).                    */
typedef struct {
    double  row_ms, col_ms, other_ms;
    int64_t row_n, col_n, other_n;
    double  col_h_ms, col_adv_ms, col_fin_ms;
    int64_t col_h_n, col_adv_n, col_fin_n;
    int64_t outliers;       /* launches whose events were more than 8 x the median of their class apart (the stream was held up: a clock
                               transition, the profiler, another process): counted here, left out of the sums above */
} ssf_kernel_times;
int  ssf_set_profiling(ssf_plan *plan, int32_t enable);
int  ssf_get_kernel_times(ssf_plan *plan, ssf_kernel_times *out);

/* ---- measured memory ceiling ("also report against an empirically measured
 * device-copy bandwidth from the same run").  Launches a kernel with the memory shape of the
 * fused row stage and no arithmetic: every 256-thread workgroup fetches 16 x 16 B per thread
 * in one burst and stores them again, `bytes` in and `bytes` out per launch, ping-ponging between
 * two buffers.  *gbs = (read + written bytes) / average launch time.  No reference equivalent. */
int  ssf_device_copy_bandwidth(int device, int64_t bytes, int32_t launches, double *gbs);

/* ---- linear channel (gamma = 0 closed form): one FFT . H . IFFT over the whole length.  field_in_soa == NULL: the field the plan
 * already holds (ssf_upload / ssf_upload_aos -- the reference's own (N, ncols) layout, host or device pointer); field_out_soa ==
 * NULL: the result stays in the plan (ssf_download / ssf_download_aos).  Reference: optic/models/channels.py:30-109. */
int  ssf_linear_channel(ssf_plan *plan, double Fs, double Fc, double alpha, double D,
                        double L, const void *field_in_soa, void *field_out_soa);

/* ---- overlap-save FFT convolution: the engine of edc -------------------------------------- */
/* Replaces blockwiseFFTConv(x, h, NFFT, freqDomainFilter=True) as edc calls it, once per mode
 * (optic/dsp/equalization.py:113-117, optic/dsp/core.py:973-1046), for all modes at once.
 * sig_in / sig_out: (sigLen, nrows) row-major complex (the reference's sigIn layout).
 * Hfft: nfft complex values = fft(zero-padded impulse response) (core.py:1020); nfft = 2^m,
 * 16 <= nfft <= 8192 (SSF_C128) / 16384 (SSF_C64), K = filter length <= nfft. */
int  ssf_overlap_save(int device, int64_t sigLen, int32_t nrows, int32_t precision, int32_t nfft,
                      int32_t K, const void *Hfft, const void *sig_in, void *sig_out);

/* ---- device-resident arrays: chaining calls without crossing PCIe -----------------------------
 * The reference's cupy twin converts to numpy at every function boundary (cp.asnumpy,
 * optic/models/modelsGPU.py:271, 501-509; optic/dsp/coreGPU.py:72), so a channel -> receiver -> DBP
 * chain crosses the bus five times.  Here every `const void *` / `void *` array argument of this ABI
 * (fields, signals, LO, noise, snapshots; NOT the small filter / parameter arrays: Hfft, taps,
 * save_spans, trace) may also be a device pointer obtained from ssf_device_malloc on the same
 * device: the library recognises it (hipPointerGetAttributes) and copies device to device.
 * Layouts and sizes are unchanged.  ssf_device_memcpy copies between any two host / device
 * buffers and is synchronous at return. */
int  ssf_device_malloc(int device, int64_t bytes, void **ptr);
int  ssf_device_free(int device, void *ptr);
int  ssf_device_memcpy(int device, void *dst, const void *src, int64_t bytes);
/* y += alpha * x on n float64 values, host or device pointers (a device y is updated in place): balancedPD's i1 - i2
 * (optic/models/devices.py:456-458) when the two photocurrents are device arrays */
int  ssf_device_axpy(int device, int64_t n, double alpha, const double *x, double *y);

/* ---- receiver side of the channel: FIR filtering, fractional delay,
 * decimation and the coherent front-end.  All arrays are host buffers, complex128 interleaved,
 * (samples, columns) row-major exactly as the reference passes them; every call uploads its
 * inputs once, keeps all intermediate stages in device memory and downloads the result.
 *
 *   ssf_fir_filter       firFilter            optic/dsp/core.py:87-125 (GPU twin optic/dsp/coreGPU.py:27-78)
 *   ssf_delay_signal     delaySignal          optic/dsp/core.py:880-922
 *   ssf_decimate         decimate             optic/dsp/core.py:435-491
 *   ssf_rx_run           photodiode / balancedPD / coherentReceiver / pdmCoherentReceiver / iqMixing
 *                                             optic/models/devices.py:289-668, optic/dsp/core.py:925-970 */

/* 'same'-mode convolution of every column with `ntaps` complex taps (1 <= ntaps <= 4096) */
int  ssf_fir_filter(int device, int64_t sigLen, int32_t ncols, int32_t ntaps, const void *taps,
                    const void *sig_in, void *sig_out);
/* FIR of ANY length, evaluated on the device: sig_out[n, m] = sum_t taps[t] * sig_in[n + shift - t, m] for n in [0, outLen),
 * sig_in (inLen samples per column) extended with zeros on both sides; `shift` of either sign.  What it replaces:
 * blockwiseFFTConv's result (optic/dsp/core.py:973-1046; shift = (ntaps - 1) / 2, outLen = inLen) for filters of more than
 * 4096 taps -- edc over long links (optic/dsp/equalization.py:85-117), delaySignal with NFFT != 1024 (core.py:880-922: its
 * zero padding, np.roll(-1) and [:N] cut are shift + 1 and outLen = N), firFilter with long responses.  The impulse response is
 * cut into segments of 4096 taps, one overlap-save launch each, added in place: ceil(ntaps / 4096) passes over the signal.
 * Host or device pointers; a device signal never leaves the device. */
int  ssf_fir_long(int device, int64_t inLen, int64_t outLen, int32_t ncols, int64_t ntaps, const void *taps, int64_t shift,
                  const void *sig_in, void *sig_out);
/* one column delayed by `delay` seconds (NFFT = 1024 as the reference's default) */
int  ssf_delay_signal(int device, int64_t N, double delay, double Fs, const void *sig_in, void *sig_out);
/* The two helpers of the Manakov step the reference exports on their own (inside ssf_execute they are fused into
 * the column kernel).  n = number of samples of each array, complex128 interleaved, host or device pointers.
 *   ssf_nlin_phase_rot         nlinPhaseRot          optic/models/channels.py:471-493 (modelsGPU.py:514-535)
 *       phi[i] = (8/9) gamma (Pch[i] + |Ex[i]|^2 + |Ey[i]|^2) / 2     (Pch: the real part, n doubles)
 *   ssf_convergence_condition  convergenceCondition  optic/models/channels.py:496-519 (modelsGPU.py:538-561)
 *       *lim = sqrt(|Ex_fd - Ex_conv|^2 + |Ey_fd - Ey_conv|^2) / sqrt(|Ex_conv|^2 + |Ey_conv|^2)  (Frobenius norms) */
int  ssf_nlin_phase_rot(int device, int64_t n, double gamma, const void *Ex, const void *Ey, const double *Pch,
                        double *phi);
int  ssf_convergence_condition(int device, int64_t n, const void *Ex_fd, const void *Ey_fd, const void *Ex_conv,
                               const void *Ey_conv, double *lim);
/* maximum-variance sampling phase per column, then every decFactor-th sample; N % SpSin == 0,
 * ncols <= 8; sig_out: (ceil(N / decFactor), ncols); sampDelay (may be NULL): ncols phases */
int  ssf_decimate(int device, int64_t N, int32_t ncols, int32_t SpSin, int32_t decFactor,
                  const void *sig_in, void *sig_out, int32_t *sampDelay);

/* ---- the amplifier and the passive optics by themselves: element-wise device passes, complex128, host or device pointers (a
 * device array never leaves the device; an in-place ssf_edfa -- field_out == field_in -- is allowed).
 *   ssf_edfa                edfa              optic/models/devices.py:671-726 (GPU twin optic/models/modelsGPU.py:56-114)
 *       field_out = field_in * sqrt(G_lin) + noise over `n` complex values laid out (samples, ncols) row-major.  noise != NULL:
 *       the caller's n complex values (the reference's seeded np.random draws, draw for draw); noise == NULL and rng_seed != 0:
 *       CN(0, p_noise) from Philox4x32-10 (counter = sample, rng_row_offset + column), statistical parity like the span epilogue
 *       of ssf_execute; both absent: gain only.  G_lin / p_noise as the reference derives them (devices.py:712-722).
 *   ssf_pbs                 pbs               optic/models/devices.py:223-260
 *       (Ex, Ey) = [ex, ey] @ [[cos t, -sin t], [sin t, cos t]] for a field of shape (N, 2), or (N,) taken as [ex, 0]
 *   ssf_optical_hybrid_2x4  opticalHybrid2x4  optic/models/devices.py:462-500
 *       Eo (4, N) = T @ [Es, 0, 0, Elo] with the 90-degree hybrid's transfer matrix T */
int  ssf_edfa(int device, int64_t n, int32_t ncols, double G_lin, double p_noise, int64_t rng_seed, int32_t rng_row_offset,
              const void *field_in, const void *noise, void *field_out);
int  ssf_pbs(int device, int64_t N, int32_t ncols, double theta, const void *E, void *Ex, void *Ey);
int  ssf_optical_hybrid_2x4(int device, int64_t N, const void *Es, const void *Elo, void *Eo);

typedef struct {
    double  Fs;
    /* pdmCoherentReceiver: paramFE.polRotation / pdl / polDelay (devices.py:649-662) */
    double  polRotation, pdl, polDelay;
    /* iqMixing per polarisation: index 0 = X or the only polarisation, 1 = Y (core.py:925-970) */
    double  ampImb[2], phaseImb[2], timeSkew[2];
    /* photodiode (devices.py:289-399); defaults are applied by the caller */
    double  R, Tc, Id, RL, B, IpdSat;
    int32_t N;                    /* filter taps (an even value is incremented, devices.py:361-365) */
    int32_t fType;                /* 0 'rect', 1 'gauss' */
    int32_t ideal, shotNoise, thermalNoise, currentSaturation, bandwidthLimitation;
    int32_t pad_;
    int64_t rng_seed;             /* device noise streams (Philox4x32-10; one counter row per pair of photodiodes, single-precision Box-Muller) */
    double  Fs_pd;                /* sampling rate of the photodiode model -- noise scale, low-pass design, the Fs >= 2 B check:
                                   * paramPD.Fs of the reference (devices.py:331-353, 562-563) -- when it differs from Fs
                                   * (paramFE.Fs: polarisation delay, IQ skew); 0 = Fs */
} ssf_rx_params;

enum ssf_rx_mode {
    SSF_RX_PHOTODIODE   = 0,      /* in0 (N, nmodes) field          -> out (N,) float64            */
    SSF_RX_BALANCED_PD  = 1,      /* in0 (N, 2) = [E1, E2]          -> out (N,) float64            */
    SSF_RX_COHERENT     = 2,      /* in0 (N,) signal, lo (N,)       -> out (N,) complex128         */
    SSF_RX_PDM_COHERENT = 3,      /* in0 (N, 2) signal, lo (N,)     -> out (N, 2) complex128       */
    SSF_RX_IQ_MIXING    = 4       /* in0 (N,) complex               -> out (N,) complex128         */
};
/* unit_normals: NULL = noise drawn on the device; otherwise [(pd * 2 + kind) * N + n] standard
 * normals, kind 0 = shot / 1 = thermal, pd = photodiode slot: photodiode 0; balanced pair 0, 1;
 * coherent receiver per polarisation p: 4p + {0: I+, 1: I-, 2: Q+, 3: Q-} (devices.py:562-563) */
int  ssf_rx_run(int device, int32_t mode, int64_t N, int32_t nmodes, const ssf_rx_params *params,
                const void *in0, const void *lo, const double *unit_normals, void *out);

/* ---- the receiver side of the coherent notebooks in ONE call: pdmCoherentReceiver -> firFilter (matched filter) -> decimate -> edc
 * (examples/test_WDM_transmission.ipynb cells 17 - 23; optic/models/devices.py:574-668, optic/dsp/core.py:87-125, 435-491,
 * optic/dsp/equalization.py:36-122), with the results of the four calls made one after the other.  What one call adds: the
 * stages' launches follow each other on the stream with one host wait at the end, decimate's variance search rides in the matched
 * filter's stores and its gather in the compensating filter's loads (no pass over the filtered signal of its own, the decimated
 * signal never materialised).
 *   Es (N, 2), Elo (N,) complex128, host or device;  taps: ntaps complex128 (<= 4096), host
 *   SpSin / decFactor as ssf_decimate (N % SpSin == 0)
 *   edc_Hfft: edc_nfft complex128 = fft(zero-padded impulse response) as ssf_overlap_save takes it, edc_K its taps, host
 *   sig_out ((N + decFactor - 1) / decFactor, 2) complex128, host or device;  sampDelay (may be NULL): the two sampling phases */
int  ssf_rx_chain(int device, int64_t N, const ssf_rx_params *params, const void *Es, const void *Elo, const void *taps,
                  int32_t ntaps, int32_t SpSin, int32_t decFactor, const void *edc_Hfft, int32_t edc_K, int32_t edc_nfft,
                  void *sig_out, int32_t *sampDelay);

/* ---- WDM transmitter: the signal path of simpleWDMTx, optic/models/tx.py:178-217.
 * For every channel and polarisation: zero-stuffing to SpS samples per symbol + pulse-shaping FIR
 * ('same' mode, one overlap-save launch), normalisation to unit peak, IQ modulator with the
 * reference's default bias and extinction (optic/models/devices.py:147-220), power normalisation to
 * amp^2, frequency shift to the channel's grid position, accumulation into the WDM field.  Symbol
 * sources, constellations, pulse taps and the LO phase-noise random walk are the caller's (they are
 * host-side numpy draws in the reference: optic/comm/sources.py:137-212, optic/dsp/core.py:792-826).
 *   symbols   (nChannels, nPolModes, nSymbols) complex128
 *   taps      ntaps float64 (<= 4096)
 *   phi       (nChannels, N) float64 LO phase per channel -- or (1, N), one walk shared by every channel, with params->phi_rows = 1
 *             (what a SEEDED reference run produces: tx.py:199 reseeds np.random with the same param.seed for every channel) --
 *             or NULL for an ideal laser; N = nSymbols * SpS
 *   amp       nChannels: sqrt(Pch / nPolModes);   deltaF: nChannels grid offsets [Hz]
 *   sig_out   (N, nPolModes) complex128 (host or device);  power_out (may be NULL): nChannels * nPolModes */
typedef struct {
    double  Fs, mzmScale;
    int64_t nSymbols;
    int32_t SpS, nChannels, nPolModes, ntaps;
    /* laser phase noise generated ON THE DEVICE (used when `phi` is NULL and pn_seed != 0): per channel a random walk
     * phi[0] = 0, phi[k] = phi[k - 1] + N(0, pn_sigma^2), pn_sigma = sqrt(2 pi linewidth / Fs) (optic/dsp/core.py:792-826), from
     * Philox4x32-10 keyed by pn_seed (statistical parity: what a call WITHOUT a seed can promise; with a seed the host passes the
     * reference's own np.random draws in `phi`).  No N-sample array crosses the bus then. */
    double   pn_sigma;
    uint64_t pn_seed;
    int32_t  phi_rows;   /* rows of `phi`: 0 or nChannels = one per channel; 1 = one row for all channels (uploaded once) */
    int32_t  reserved;
} ssf_tx_params;
int  ssf_wdm_tx(int device, const ssf_tx_params *params, const void *symbols, const double *taps, const double *phi,
                const double *amp, const double *deltaF, void *sig_out, double *power_out);

#ifdef __cplusplus
}
#endif
#endif /* SSF_H */
