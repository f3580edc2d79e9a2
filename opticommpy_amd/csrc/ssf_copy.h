// ssf_copy.h -- host <-> device field transfer shared by both engines: pinned, double-buffered
// staging (pageable numpy memory is copied chunk-wise into pinned buffers while the previous
// chunk's DMA runs) and the AoS <-> SoA conversion on the device, so the host never transposes.
// Reference: cp.asarray(Ei).astype(prec) / Ei_[:, 0::2].T / cp.asnumpy (modelsGPU.py:404-407, 501-509).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

namespace ssf {

// (N, nrows) row-major interleaved  <->  (nrows, N) rows
template <typename C> __global__ void k_aos_to_soa(const C *aos, C *soa, long long N, int nrows) {
    const long long total = N * nrows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / nrows;
        const int r = (int)(i - n * nrows);
        soa[(long long)r * N + n] = aos[i];
    }
}
template <typename C> __global__ void k_soa_to_aos(const C *soa, C *aos, long long N, int nrows) {
    const long long total = N * nrows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / nrows;
        const int r = (int)(i - n * nrows);
        aos[i] = soa[(long long)r * N + n];
    }
}

// true if p points into device memory (a buffer of ssf_device_malloc, or any other HIP allocation): the
// "host" pointers of the C ABI may be device pointers, in which case transfers become device copies
inline bool on_device(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();               // plain pageable host memory is not known to the runtime
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

class Stager {
    static constexpr size_t kChunk = 8u << 20;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok_ = false;

  public:
    hipError_t init() {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&pin[i], kChunk);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        ok_ = true;
        return hipSuccess;
    }
    ~Stager() {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
    // synchronous at return
    hipError_t h2d(void *dev, const void *host, size_t n, hipStream_t s) {
        if (on_device(host)) {                  // device-resident input: no staging, no PCIe
            hipError_t e = hipMemcpyAsync(dev, host, n, hipMemcpyDeviceToDevice, s);
            return e != hipSuccess ? e : hipStreamSynchronize(s);
        }
        if (!ok_ || n < (1u << 20)) {
            hipError_t e = hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, s);
            return e != hipSuccess ? e : hipStreamSynchronize(s);
        }
        size_t off = 0;
        for (int i = 0; off < n; ++i, off += kChunk) {
            const int b = i & 1;
            const size_t len = n - off < kChunk ? n - off : kChunk;
            if (i >= 2) {
                hipError_t e = hipEventSynchronize(ev[b]);
                if (e != hipSuccess) return e;
            }
            std::memcpy(pin[b], (const char *)host + off, len);
            hipError_t e = hipMemcpyAsync((char *)dev + off, pin[b], len, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return e;
            if ((e = hipEventRecord(ev[b], s)) != hipSuccess) return e;
        }
        return hipStreamSynchronize(s);
    }
    hipError_t d2h(void *host, const void *dev, size_t n, hipStream_t s) {
        if (on_device(host)) {
            hipError_t e = hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToDevice, s);
            return e != hipSuccess ? e : hipStreamSynchronize(s);
        }
        if (!ok_ || n < (1u << 20)) {
            hipError_t e = hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, s);
            return e != hipSuccess ? e : hipStreamSynchronize(s);
        }
        const int nchunks = (int)((n + kChunk - 1) / kChunk);
        for (int i = 0; i <= nchunks; ++i) {
            if (i < nchunks) {
                const int b = i & 1;
                const size_t off = (size_t)i * kChunk, len = n - off < kChunk ? n - off : kChunk;
                // buffer b was drained two iterations ago (the memcpy below is synchronous)
                hipError_t e = hipMemcpyAsync(pin[b], (const char *)dev + off, len, hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) return e;
                if ((e = hipEventRecord(ev[b], s)) != hipSuccess) return e;
            }
            if (i >= 1) {
                const int b = (i - 1) & 1;
                const size_t off = (size_t)(i - 1) * kChunk, len = n - off < kChunk ? n - off : kChunk;
                hipError_t e = hipEventSynchronize(ev[b]);
                if (e != hipSuccess) return e;
                std::memcpy((char *)host + off, pin[b], len);
            }
        }
        return hipSuccess;
    }
};

}  // namespace ssf
