// ssf_snapshots.h -- streaming the per-span snapshots out of a propagation.
//
// Reference: `Ech_spans[:, 2*indRecSpan : 2*indRecSpan+2] = ...` at the end of every span listed in saveSpanN
// (optic/models/channels.py:453-456, modelsGPU.py:497-498): the result is ONE (N, 2 * len(saveSpanN)) array.  With a
// sink set (ssf_set_snapshot_sink) a captured field goes straight into its columns of that array:
//   * device destination: the SoA -> (N, ld) conversion kernel writes the columns in place (nothing else);
//   * host destination: the field is converted into one of two device staging blocks on the plan's stream, and a copy
//     thread moves it to the host on its own stream (pinned double buffer, DMA of one chunk while the previous one is
//     scattered into the caller's rows) WHILE THE NEXT SPAN PROPAGATES.  ssf_sync_snapshots waits for the last one.
// Without a sink the engines keep the snapshots in device memory until ssf_download_snapshots (round-1 behaviour).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

#include "ssf_copy.h"

namespace ssf {

// rows of an (nrows, N) field -> columns [col0, col0 + nrows) of a row-major (N, ld) array
template <typename C> __global__ void k_soa_to_cols(const C *soa, C *out, long long N, int nrows, long long ld, long long col0) {
    const long long total = N * nrows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / nrows;
        const int r = (int)(i - n * nrows);
        out[n * ld + col0 + r] = soa[(long long)r * N + n];
    }
}

class SnapshotSink {
    static constexpr size_t kChunk = 8u << 20;
    struct Job {
        int buf;
        long long N, col0;
        int nrows;
    };
    int device = 0;
    char *dst = nullptr;
    long long ld = 0;
    size_t esz = 16;
    bool dst_dev = false;
    int index = 0;
    // host path
    void *stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t ready[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    hipStream_t copy_st = nullptr;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    int inflight = 0;
    bool stop = false, started = false;
    hipError_t first_err = hipSuccess;

    void note(hipError_t e) {
        if (e != hipSuccess && first_err == hipSuccess) first_err = e;
    }
    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                j = jobs.front();
                jobs.pop_front();
            }
            hipError_t e = hipEventSynchronize(ready[j.buf]);              // the conversion kernel has written the block
            const size_t row = (size_t)j.nrows * esz;                      // bytes of one sample of the field
            const long long rows_per_chunk = std::max<long long>(1, (long long)(kChunk / row));   // (set() rejects rows wider than a chunk)
            const long long nchunks = (j.N + rows_per_chunk - 1) / rows_per_chunk;
            for (long long i = 0; e == hipSuccess && i <= nchunks; ++i) {
                if (i < nchunks) {                                          // DMA chunk i into pinned buffer i & 1
                    const int b = (int)(i & 1);
                    const long long r0 = i * rows_per_chunk, nr = std::min(rows_per_chunk, j.N - r0);
                    e = hipMemcpyAsync(pin[b], (const char *)stage[j.buf] + (size_t)r0 * row, (size_t)nr * row, hipMemcpyDeviceToHost, copy_st);
                    if (e == hipSuccess) e = hipEventRecord(pin_ev[b], copy_st);
                }
                if (e == hipSuccess && i >= 1) {                            // scatter chunk i - 1 into the caller's rows
                    const int b = (int)((i - 1) & 1);
                    const long long r0 = (i - 1) * rows_per_chunk, nr = std::min(rows_per_chunk, j.N - r0);
                    e = hipEventSynchronize(pin_ev[b]);
                    if (e == hipSuccess) {
                        const char *src = (const char *)pin[b];
                        char *d = dst + ((size_t)r0 * (size_t)ld + (size_t)j.col0) * esz;
                        if ((long long)j.nrows == ld) std::memcpy(d, src, (size_t)nr * row);
                        else
                            for (long long r = 0; r < nr; ++r) std::memcpy(d + (size_t)r * (size_t)ld * esz, src + (size_t)r * row, row);
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                note(e);
                busy[j.buf] = false;
                --inflight;
            }
            cv.notify_all();
        }
    }
    hipError_t start() {
        if (started) return hipSuccess;
        // idempotent per resource: a start that failed half way is continued, not repeated (no second set of buffers)
        hipError_t e = copy_st ? hipSuccess : hipStreamCreateWithFlags(&copy_st, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            if (!pin[i]) e = hipHostMalloc(&pin[i], kChunk);
            if (e == hipSuccess && !pin_ev[i]) e = hipEventCreateWithFlags(&pin_ev[i], hipEventDisableTiming);
            if (e == hipSuccess && !ready[i]) e = hipEventCreateWithFlags(&ready[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) return e;
        worker = std::thread([this] { run(); });
        started = true;
        return hipSuccess;
    }

  public:
    SnapshotSink() = default;
    SnapshotSink(const SnapshotSink &) = delete;
    ~SnapshotSink() {
        (void)sync();
        if (started) {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            worker.join();
        }
        for (int i = 0; i < 2; ++i) {
            if (stage[i]) (void)hipFree(stage[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (pin_ev[i]) (void)hipEventDestroy(pin_ev[i]);
            if (ready[i]) (void)hipEventDestroy(ready[i]);
        }
        if (copy_st) (void)hipStreamDestroy(copy_st);
    }
    bool active() const { return dst != nullptr; }
    int count() const { return index; }
    long long leading() const { return ld; }
    // dst = nullptr detaches.  Waits for copies into the previous destination first.
    hipError_t set(int dev, void *d, long long ld_, int first_index, size_t elem_size) {
        hipError_t e = sync();
        device = dev;
        dst = (char *)d;
        ld = ld_;
        index = first_index;
        esz = elem_size;
        dst_dev = d && on_device(d);
        return e;
    }
    // Called by the engine at a captured span, on the thread that owns the plan; `soa` is the (nrows, N) field and
    // every earlier launch on `st` has produced it.
    template <typename C> hipError_t capture(const C *soa, long long N, int nrows, hipStream_t st) {
        // capture i goes to columns [i * nrows, (i + 1) * nrows) of the caller's (N, ld) array: never beyond it (a save_spans
        // list longer than the array, or a first_index that is too large, is the caller's error, not a stray write)
        if (((long long)index + 1) * nrows > ld || (size_t)nrows * esz > kChunk) return hipErrorInvalidValue;
        const long long col0 = (long long)index * nrows;
        ++index;
        if (dst_dev) {
            k_soa_to_cols<C><<<1024, 256, 0, st>>>(soa, (C *)dst, N, nrows, ld, col0);
            return hipGetLastError();
        }
        hipError_t e = start();
        if (e != hipSuccess) return e;
        int b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy[0] || !busy[1]; });
            b = !busy[0] ? 0 : 1;
            busy[b] = true;
            ++inflight;
        }
        const size_t bytes = (size_t)N * (size_t)nrows * sizeof(C);
        if (stage_cap[b] < bytes) {
            if (stage[b]) (void)hipFree(stage[b]);
            stage[b] = nullptr;
            stage_cap[b] = 0;
            if ((e = hipMalloc(&stage[b], bytes)) == hipSuccess) stage_cap[b] = bytes;
        }
        if (e == hipSuccess) {
            k_soa_to_cols<C><<<1024, 256, 0, st>>>(soa, (C *)stage[b], N, nrows, (long long)nrows, 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(ready[b], st);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (e == hipSuccess) jobs.push_back(Job{b, N, col0, nrows});
            else {                                                  // nothing was queued: give the block back
                busy[b] = false;
                --inflight;
            }
        }
        cv.notify_all();
        return e;
    }
    // every captured snapshot has reached its destination (host destinations; device ones are ordered on the plan stream)
    hipError_t sync() {
        if (!started) return hipSuccess;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return inflight == 0; });
        const hipError_t e = first_err;
        first_err = hipSuccess;
        return e;
    }
};

}  // namespace ssf
