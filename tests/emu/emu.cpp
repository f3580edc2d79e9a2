// emu.cpp -- CPU emulator for the fused-engine kernels.  TEST INFRASTRUCTURE ONLY: it is
// built into tests/emu/libssf_emu.so, loaded only by tests/test_emu_fused.py, and is not
// reachable from the opticommpy_amd package (the product has no CPU path).
//
// It compiles the SAME kernel bodies (opticommpy_amd/csrc/fused_kernels.h) and the SAME host
// control code (fused_engine.h) as libssf_hip.so, with a backend that "launches" a kernel by
// stepping every workgroup's threads as fibers: ctx.sync() yields to a round-robin scheduler,
// so one scheduler round == one __syncthreads() interval.  Workgroups run on a few host
// threads.  This checks the index maps, twiddles, LDS exchange pattern, device-side control
// state machine and host enqueue logic against the oracle without a GPU.
#include <ucontext.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "fused_engine.h"
#include "mixed_fft.h"
#include "rx_pipeline.h"

namespace {

struct EmuCtx {
    int tid, bid, nthreads, nblocks;
    char *lds;
    static constexpr bool kWaveOps = false;
    void sync();
    void mark(int) {}
    void flush(int) {}
    void issue_fence() {}
    template <int P> void setprio() {}
    double xchg(double v);                    // the value thread tid ^ 1 passes (DevCtxCore::xchg: DPP)
    double wave_sum(double v) { return v; }
    double wave_max(double v) { return v; }
};

// Context switch of the fibers.  x86-64: six callee-saved registers and the stack pointer, in user space (glibc's swapcontext makes a
// rt_sigprocmask system call per switch -- a third of the suite's CPU time was spent in the kernel); elsewhere: ucontext.
#if defined(__x86_64__)
#define EMU_ASM_SWITCH 1
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");
#else
#define EMU_ASM_SWITCH 0
#endif

struct Fiber {
#if EMU_ASM_SWITCH
    void *sp = nullptr;
#else
    ucontext_t uc;
#endif
    char *stack = nullptr;
    bool done = true;
    EmuCtx ctx;
};

constexpr size_t kStack = 96 * 1024;

struct Pool {
    std::vector<Fiber> f;
#if EMU_ASM_SWITCH
    void *main_sp = nullptr;
#else
    ucontext_t main_uc;
#endif
    Fiber *cur = nullptr;
    const std::function<void(EmuCtx &)> *body = nullptr;
    std::vector<char> lds;
    std::vector<double> xch;
    ~Pool() {
        for (auto &x : f) free(x.stack);
    }
};
thread_local Pool g_pool;

inline void to_main(Pool &p, Fiber *me) {
#if EMU_ASM_SWITCH
    emu_switch(&me->sp, p.main_sp);
#else
    swapcontext(&me->uc, &p.main_uc);
#endif
}
inline void to_fiber(Pool &p, Fiber &f) {
#if EMU_ASM_SWITCH
    emu_switch(&p.main_sp, f.sp);
#else
    swapcontext(&p.main_uc, &f.uc);
#endif
}

void trampoline() {
    Pool &p = g_pool;
    Fiber *me = p.cur;
    (*p.body)(me->ctx);
    me->done = true;
    to_main(p, me);
    abort();                                   // (a finished fiber is never resumed)
}

void EmuCtx::sync() {
    Pool &p = g_pool;
    to_main(p, p.cur);
}

double EmuCtx::xchg(double v) {
    Pool &p = g_pool;
    p.xch[(size_t)tid] = v;
    sync();
    const double r = p.xch[(size_t)(tid ^ 1)];
    sync();                                    // (nobody overwrites a value its partner has not read yet)
    return r;
}

void run_block(int bid, int nblocks, int nthreads, size_t lds_bytes, const std::function<void(EmuCtx &)> &body) {
    Pool &p = g_pool;
    if (p.xch.size() < (size_t)nthreads) p.xch.resize((size_t)nthreads);
    if ((int)p.f.size() < nthreads) {
        const size_t old = p.f.size();
        p.f.resize((size_t)nthreads);
        for (size_t i = old; i < p.f.size(); ++i) p.f[i].stack = (char *)malloc(kStack);
    }
    if (p.lds.size() < lds_bytes + 64) p.lds.resize(lds_bytes + 64);
    // poison the LDS so that reads of never-written slots show up as NaNs
    memset(p.lds.data(), 0xFF, lds_bytes + 64);
    p.body = &body;
    for (int t = 0; t < nthreads; ++t) {
        Fiber &f = p.f[(size_t)t];
#if EMU_ASM_SWITCH
        // the frame emu_switch pops: six registers, then `ret` into trampoline with the stack as a call would have left it
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        void **sp = (void **)(top - 8);
        *sp = nullptr;                                         // (trampoline's "return address": it never returns)
        *--sp = (void *)&trampoline;
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        f.sp = (void *)sp;
#else
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack;
        f.uc.uc_stack.ss_size = kStack;
        f.uc.uc_link = nullptr;
        makecontext(&f.uc, trampoline, 0);
#endif
        f.done = false;
        f.ctx = EmuCtx{t, bid, nthreads, nblocks, p.lds.data()};
    }
    int alive = nthreads;
    while (alive > 0) {
        for (int t = 0; t < nthreads; ++t) {
            Fiber &f = p.f[(size_t)t];
            if (f.done) continue;
            p.cur = &f;
            to_fiber(p, f);
            if (f.done) --alive;
        }
    }
}

void run_grid(int grid, int block, size_t lds, const std::function<void(EmuCtx &)> &body) {
    const int nw = std::max(1, std::min(8, grid));
    if (nw == 1) {
        for (int b = 0; b < grid; ++b) run_block(b, grid, block, lds, body);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int w = 0; w < nw; ++w)
        th.emplace_back([&] {
            for (;;) {
                const int b = next.fetch_add(1);
                if (b >= grid) break;
                run_block(b, grid, block, lds, body);
            }
        });
    for (auto &t : th) t.join();
}

struct EmuBackend {
    long launches = 0;
    void *alloc(size_t n) { return calloc(n ? n : 1, 1); }
    void free(void *p) { ::free(p); }
    void h2d(void *d, const void *h, size_t n) { memcpy(d, h, n); }
    void d2h(void *h, const void *d, size_t n) { memcpy(h, d, n); }
    void d2d(void *d, const void *s, size_t n) { memcpy(d, s, n); }
    void h2d_big(void *d, const void *h, size_t n) { memcpy(d, h, n); }
    void d2h_big(void *h, const void *d, size_t n) { memcpy(h, d, n); }
    template <typename C> void aos_to_soa(C *soa, const C *aos, long long N, int nrows) {
        for (long long n = 0; n < N; ++n)
            for (int r = 0; r < nrows; ++r) soa[(long long)r * N + n] = aos[n * nrows + r];
    }
    template <typename C> void soa_to_aos(C *aos, const C *soa, long long N, int nrows) {
        for (long long n = 0; n < N; ++n)
            for (int r = 0; r < nrows; ++r) aos[n * nrows + r] = soa[(long long)r * N + n];
    }
    void memset(void *d, int v, size_t n) { ::memset(d, v, n); }
    void prepare(size_t, size_t) {}
    void sync() {}
    bool ok() const { return true; }
    std::string last_error() const { return ""; }
    void time_begin() {}
    double time_end() { return 0.0; }
    template <typename T> void launch_row(const ssf::fused::RowArgs<T> &a0, int grid, int block, size_t lds, int units = 1) {
        ++launches;
        for (int u = 0; u < units; ++u) {
            const ssf::fused::RowArgs<T> a = ssf::fused::unit_view(a0, u);
            launch_row_unit(a, grid, block, lds);
        }
    }
    template <typename T> void launch_row_unit(const ssf::fused::RowArgs<T> &a, int grid, int block, size_t lds) {
        if (getenv("SSF_EMU_DEBUG") && launches < 8) fprintf(stderr, "emu row: vpt %d block %d grid %d lds %zu\n", a.vpt, block, grid, lds);
        if constexpr (!std::is_same<T, ssf::fused::pf2>::value) {
            if (a.mixed) {
                run_grid(grid, block, lds, [&](EmuCtx &c) { ssf::fused::row_mixed_body<T>(c, a, a.plan); });
                return;
            }
        }
        if (a.vpt == 8) run_grid(grid, block, lds, [&](EmuCtx &c) { ssf::fused::row_body<T, 0, 8>(c, a); });
        else run_grid(grid, block, lds, [&](EmuCtx &c) { ssf::fused::row_body<T, 0>(c, a); });
    }
    static constexpr bool kCanPersist = false;         // (no grid barrier between the emulator's sequential workgroups)
    static constexpr bool kCanCouple = false;          // (no communicator)
    bool sink_active() const { return false; }
    template <typename C> void sink_capture(const C *, long long, int) {}
    void launch_repack(const ssf::fused::RepackArgs &a, int grid, int block) {
        ++launches;
        run_grid(std::min(grid, 8), block, 64, [&](EmuCtx &c) { ssf::fused::repack_body(c, a); });
    }
    template <typename T> bool can_split_cols(const ssf::fused::ColArgs<T> &a, int) const { return !a.N2 && a.vpt != 8; }
    template <typename T> void launch_col(const ssf::fused::ColArgs<T> &a0, int grid, int block, size_t lds, int units = 1) {
        ++launches;
        for (int u = 0; u < units; ++u) {
            const ssf::fused::ColArgs<T> a = ssf::fused::unit_view(a0, u);
            launch_col_unit(a, grid, block, lds);
        }
    }
    template <typename T> void launch_col_unit(const ssf::fused::ColArgs<T> &a, int grid, int block, size_t lds) {
        if (getenv("SSF_EMU_DEBUG") && launches < 8) fprintf(stderr, "emu col: vpt %d block %d grid %d lds %zu\n", a.vpt, block, grid, lds);
        using namespace ssf::fused;
        if constexpr (std::is_same<T, pf2>::value) {          // packed pair: only the Manakov stage exists
            // (the HIP backend's choice: the columns of a workgroup interleaved in LDS where the chip-filling geometries have 4 / 8)
            const int cols = (block / a.npol) / ((1 << a.log2N1) / (a.vpt == 8 ? 8 : 16));
            const bool il = !getenv("SSF_COL_IL") || atoi(getenv("SSF_COL_IL")) != 0;
            if (a.vpt != 8 && a.sg && a.sg != SG_ALL) {      // stage-specialised kernels (FusedCore::run_span, col_split)
                switch (a.sg) {
                case SG_H | SG_RARE: run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 0, SG_H | SG_RARE>(c, a); }); break;
                case SG_ADV: run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 0, SG_ADV>(c, a); }); break;
                case SG_FIN: run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 0, SG_FIN>(c, a); }); break;
                default: run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 0, SG_ALL>(c, a); }); break;
                }
                return;
            }
            if (a.vpt == 8) run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 8>(c, a); });
            else if (il && cols == 4) run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 4>(c, a); });
            else if (il && cols == 8) run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0, 16, 8>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_pk_body<0>(c, a); });
            return;
        } else {
        if (a.N1mix) {                                      // mixed-radix columns (col_mixed_body)
            switch (a.mode) {
            case CM_NLSE_FIRST: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_NLSE_FIRST>(c, a, a.plan1); }); break;
            case CM_NLSE_STEP: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_NLSE_STEP>(c, a, a.plan1); }); break;
            case CM_NLSE_LAST: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_NLSE_LAST>(c, a, a.plan1); }); break;
            case CM_MK: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_MK>(c, a, a.plan1); }); break;
            case CM_PLAIN_FWD: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_PLAIN_FWD>(c, a, a.plan1); }); break;
            default: run_grid(grid, block, lds, [&](EmuCtx &c) { col_mixed_body<T, CM_PLAIN_INV>(c, a, a.plan1); }); break;
            }
            return;
        }
        if (a.vpt == 8 && !a.N2) {                          // eight values per thread (no ragged variant)
            switch (a.mode) {
            case CM_NLSE_FIRST: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_FIRST, false, 8>(c, a); }); break;
            case CM_NLSE_STEP: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_STEP, false, 8>(c, a); }); break;
            case CM_NLSE_LAST: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_LAST, false, 8>(c, a); }); break;
            case CM_MK: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 8>(c, a); }); break;
            case CM_PLAIN_FWD: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_FWD, false, 8>(c, a); }); break;
            default: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_INV, false, 8>(c, a); }); break;
            }
            return;
        }
        switch (a.mode) {
        case CM_NLSE_FIRST:
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_FIRST, true>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_FIRST, false>(c, a); });
            break;
        case CM_NLSE_STEP:
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_STEP, true>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_STEP, false>(c, a); });
            break;
        case CM_NLSE_LAST:
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_LAST, true>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_NLSE_LAST, false>(c, a); });
            break;
        case CM_MK: {
            const int cols = (block / a.npol) / ((1 << a.log2N1) / 16);
            const bool il = std::is_same<T, double>::value && cols == 8 && (!getenv("SSF_COL_IL") || atoi(getenv("SSF_COL_IL")) != 0);
            if (!a.N2 && a.sg && a.sg != SG_ALL) {           // stage-specialised kernels (FusedCore::run_span, col_split)
                switch (a.sg) {
                case SG_H | SG_RARE: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 16, 0, SG_H | SG_RARE>(c, a); }); break;
                case SG_ADV: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 16, 0, SG_ADV>(c, a); }); break;
                case SG_FIN: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 16, 0, SG_FIN>(c, a); }); break;
                default: run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 16, 0, SG_ALL>(c, a); }); break;
                }
                break;
            }
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, true>(c, a); });
            else if (il) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false, 16, 8>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_MK, false>(c, a); });
            break;
        }
        case CM_PLAIN_FWD:
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_FWD, true>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_FWD, false>(c, a); });
            break;
        default:
            if (a.N2) run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_INV, true>(c, a); });
            else run_grid(grid, block, lds, [&](EmuCtx &c) { col_body<T, 0, CM_PLAIN_INV, false>(c, a); });
            break;
        }
        }
    }
    template <typename T> void launch_amp(const ssf::fused::AmpArgs<T> &a, int grid, int block) {
        ++launches;
        run_grid(std::min(grid, 8), block, 64, [&](EmuCtx &c) { ssf::fused::amp_body<T>(c, a); });
    }
    // receiver pipeline (rx_pipeline.h)
    void launch_ols(const ssf::fused::OlsArgs<double> &a) {
        ++launches;
        const ssf::fused::OlsLaunch o = ssf::fused::ols_launch(a.log2nfft, a.nrows, a.njobs);
        ssf::fused::ols_dispatch(o, [&](auto lg, auto cc) {
            run_grid((int)o.grid, o.threads, o.lds_bytes, [&](EmuCtx &c) { ssf::fused::ols_body<double, decltype(lg)::value, decltype(cc)::value>(c, a); });
        });
    }
    static int ew_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>(3, (n + 63) / 64)); }
    void launch_rx_ols(const ssf::rx::RxOlsArgs &a) {
        ++launches;
        const ssf::fused::OlsLaunch o = ssf::fused::ols_launch(a.o.log2nfft, a.o.nrows, a.o.njobs);
        const bool found = ssf::rx::rx_ols_dispatch(a, o, [&](auto lg, auto cc, auto pre, auto noise) {
            run_grid((int)o.grid, o.threads, o.lds_bytes, [&](EmuCtx &c) {
                ssf::rx::rx_ols_body<decltype(lg)::value, decltype(cc)::value, decltype(pre)::value, decltype(noise)::value>(c, a);
            });
        });
        if (!found) {
            fprintf(stderr, "emu: no fused overlap-save kernel for this stage / transform size\n");
            abort();
        }
    }
    void launch_det(const ssf::rx::DetKernelArgs &a) { ++launches; run_grid(ew_grid(a.det.N * a.det.nm), 64, 64, [&](EmuCtx &c) { ssf::rx::det_body(c, a); }); }
    void launch_axpy(const ssf::rx::AxpyArgs &a) { run_grid(ew_grid(a.n), 64, 64, [&](EmuCtx &c) { ssf::rx::axpy_body(c, a); }); }
    void launch_iqf(const ssf::rx::IqfArgs &a) { ++launches; run_grid(ew_grid(a.N * a.nm), 64, 64, [&](EmuCtx &c) { ssf::rx::iqf_body(c, a); }); }
    bool is_resident(const void *) const { return true; }      // (the emulator's "device" memory is the host's)
    void *filter_lookup(const void *, size_t) { return nullptr; }                  // (no filter cache: every call builds its own)
    void *filter_store(const void *, size_t, const void *, size_t) { return nullptr; }
    void launch_front(const ssf::rx::FrontArgs &a) { ++launches; run_grid(ew_grid(a.f.N), 64, 64, [&](EmuCtx &c) { ssf::rx::front_body(c, a); }); }
    void launch_optics(const ssf::rx::OpticsArgs &a) { run_grid(ew_grid(a.n), 64, 64, [&](EmuCtx &c) { ssf::rx::optics_body(c, a); }); }
    void launch_nlin_phase(const ssf::rx::NlinPhaseArgs &a) { run_grid(ew_grid(a.n), 64, 64, [&](EmuCtx &c) { ssf::rx::nlin_phase_body(c, a); }); }
    void launch_conv_sums(const ssf::rx::ConvSumsArgs &a, int nblocks) { run_grid(nblocks, 64, 4096, [&](EmuCtx &c) { ssf::rx::conv_sums_body(c, a); }); }
    void launch_absmax(const ssf::rx::AbsMaxArgs &a, int nblocks) { run_grid(nblocks, 64, 4096, [&](EmuCtx &c) { ssf::rx::absmax_body(c, a); }); }
    void launch_pn(const ssf::rx::PnArgs &a, int nchunks) { run_grid(nchunks, 64, 64 * sizeof(double), [&](EmuCtx &c) { ssf::rx::pn_body(c, a); }); }
    void launch_iqm(const ssf::rx::IqmArgs &a, int nblocks) { run_grid(nblocks, 64, 4096, [&](EmuCtx &c) { ssf::rx::iqm_body(c, a); }); }
    void launch_shift_add(const ssf::rx::ShiftAddArgs &a) { run_grid(ew_grid(a.N), 64, 64, [&](EmuCtx &c) { ssf::rx::shift_add_body(c, a); }); }
    void launch_chain_ols(const ssf::rx::ChainOlsArgs &a, int mode) {
        ++launches;
        const ssf::fused::OlsLaunch o = ssf::fused::ols_launch(a.o.log2nfft, a.o.nrows, a.o.njobs);
        const bool found = ssf::rx::chain_ols_dispatch(o, [&](auto lg, auto cc) {
            constexpr int LG = decltype(lg)::value, C = decltype(cc)::value;
            if (mode == ssf::rx::CH_STATS) run_grid((int)o.grid, o.threads, o.lds_bytes, [&](EmuCtx &c) { ssf::rx::chain_ols_body<LG, C, ssf::rx::CH_STATS>(c, a); });
            else run_grid((int)o.grid, o.threads, o.lds_bytes, [&](EmuCtx &c) { ssf::rx::chain_ols_body<LG, C, ssf::rx::CH_GATHER>(c, a); });
        });
        if (!found) {
            fprintf(stderr, "emu: no chain kernel for this transform size\n");
            abort();
        }
    }
    void launch_chain_finish(const ssf::rx::ChainFinishArgs &a) { ++launches; run_grid(1, 1024, sizeof(double) * (3 * 1024 + 256), [&](EmuCtx &c) { ssf::rx::chain_finish_body(c, a); }); }
    void launch_dec_stats(const ssf::rx::DecStatsArgs &a, int nblocks, int nthreads) {
        run_grid(nblocks, nthreads, 3 * sizeof(double) * (size_t)nthreads, [&](EmuCtx &c) { ssf::rx::dec_stats_body(c, a); });
    }
    void launch_dec_sum(const ssf::rx::DecSumArgs &a, int nblocks, int nthreads) {
        run_grid(nblocks, nthreads, 2 * sizeof(double) * (size_t)nthreads, [&](EmuCtx &c) { ssf::rx::dec_sum_body(c, a); });
    }
    void launch_dec_finish(const ssf::rx::DecFinishArgs &a) { run_grid(1, 256, sizeof(double) * 3 * 256, [&](EmuCtx &c) { ssf::rx::dec_finish_body(c, a); }); }
    void launch_dec_gather(const ssf::rx::DecGatherArgs &a) {
        run_grid(ew_grid(a.Nout * a.ncols), 64, 64, [&](EmuCtx &c) { ssf::rx::dec_gather_body(c, a); });
    }
};

template <typename T>
int run_t(int64_t N, int nrows, int prec, const ssf_params *p, const void *in, void *out, void *snaps,
          const void *noise, ssf_stats *st, ssf_trace *tr, long *launches) {
    EmuBackend be;
    const char *eu = getenv("SSF_EMU_UNITS");                 // the rows form this many independent units
    ssf::fused::FusedCore<T, EmuBackend> core(be, N, nrows, prec, nullptr, eu ? atoi(eu) : 1);
    int rc = core.init();
    if (rc) return rc;
    if ((rc = core.upload(in, false))) return rc;
    ssf_stats s{};
    if (tr) tr->count = 0;
    if ((rc = core.execute(*p, 1, p->Nspans, noise, &s, tr))) {
        fprintf(stderr, "emu: %s\n", core.err.c_str());
        return rc;
    }
    s.bytes_algorithmic = (double)s.transforms * 2.0 * sizeof(ssf::fused::cx<T>) * (double)N;
    s.engine = SSF_ENGINE_FUSED;
    if (st) *st = s;
    if (out && (rc = core.download(out, -1, false))) return rc;
    if (snaps)
        for (size_t i = 0; i < core.snaps.size(); ++i)
            if ((rc = core.download((char *)snaps + i * core.field_bytes, (int)i, false))) return rc;
    if (launches) *launches = be.launches;
    return SSF_OK;
}

template <typename T>
int lin_t(int64_t N, int nrows, int prec, double Fs, double Fc, double alpha, double D, double L, const void *in,
          void *out) {
    EmuBackend be;
    ssf::fused::FusedCore<T, EmuBackend> core(be, N, nrows, prec);
    int rc = core.init();
    if (rc) return rc;
    if ((rc = core.upload(in, false))) return rc;
    if ((rc = core.linear_channel(Fs, Fc, alpha, D, L))) return rc;
    return core.download(out, -1, false);
}

}  // namespace

// the general-length engine's one-launch linear step at short 5-smooth lengths (engine_fused_impl.h: FusedRowsImpl):
// out = ifft(fft(in) * exp((lin_a + j lin_b w^2) hzh)) row by row, the whole row in LDS (row_mixed_body with N1 = 1)
template <typename T>
static int rows_lin_t(int64_t N, int nrows, const void *in, void *out, double hzh, double lin_a, double lin_b, double w_scale) {
    using namespace ssf::fused;
    int tpr = 128, rows_wg;
    while (16 * tpr < N) tpr *= 2;
    rows_wg = std::max(1, 256 / tpr);
    while (nrows % rows_wg) rows_wg >>= 1;
    MixPlan plan;
    if (N > 8192 || !mix_make_plan((int)N, &plan, tpr)) return SSF_ERR_UNSUPPORTED;
    std::vector<cx<double>> w((size_t)N);
    for (int64_t q = 0; q < N; ++q) {
        const double a = -2.0 * 3.14159265358979323846 * (double)q / (double)N;
        w[(size_t)q].re = std::cos(a);
        w[(size_t)q].im = std::sin(a);
    }
    const double w2 = (w_scale / (double)N) * (w_scale / (double)N);
    LinOp lo = make_linop(hzh, lin_a, lin_b, w2, 1.0 / (double)N, 4);
    RowArgs<T> a{};
    a.G = (cx<T> *)out;
    a.src = in == out ? nullptr : (const cx<T> *)in;
    a.nfft = nrows;
    a.lin = &lo;
    a.N2 = (int)N;
    a.N = N;
    a.mixed = 1;
    a.plan = plan;
    a.wtab = w.data();
    a.rows_per_wg = rows_wg;
    run_grid(nrows / rows_wg, tpr * rows_wg, 4096 + (size_t)rows_wg * (size_t)N * sizeof(cx<T>), [&](EmuCtx &c) { row_mixed_body<T>(c, a, a.plan); });
    return SSF_OK;
}

extern "C" {

int emu_supported(int64_t N, int precision) {
    if (N >= 2 && (N & (N - 1))) {
        int l1, n2, n1, c;
        return (ssf::fused::choose_mixed_split(N, precision, &l1, &n2) || ssf::fused::choose_mixed2_split(N, precision, &n1, &n2, &c)) ? 1 : 0;
    }
    if (N < 2 || (N & (N - 1))) return 0;
    int l = 0;
    while ((1ll << l) < N) ++l;
    ssf::fused::Split s;
    return ssf::fused::choose_split(l, precision, &s) ? 1 : 0;
}

// the N1 x N2 split (and columns per workgroup) of the mixed-radix column stage, 0 when the length does not go there
int emu_mixed2_split(int64_t N, int precision, int *n1, int *n2, int *c) {
    int l1, m2, f1 = 0, fc = 0;
    if (const char *e = getenv("SSF_MIX2")) {
        f1 = atoi(e);
        if (const char *q = strchr(e, ',')) fc = atoi(q + 1);
    }
    if (!ssf::fused::choose_nonpow2_split(N, precision, f1, fc, getenv("SSF_MIX_L1") != nullptr, &l1, n1, &m2, c)) return 0;
    *n2 = m2;
    return *n1 > 0 ? 1 : 0;
}
int emu_split(int64_t N, int precision, int *l1, int *l2) {
    int l = 0;
    while ((1ll << l) < N) ++l;
    ssf::fused::Split s;
    if (!ssf::fused::choose_split(l, precision, &s)) return -1;
    *l1 = s.l1;
    *l2 = s.l2;
    return 0;
}

int emu_run(int64_t N, int nrows, int precision, const ssf_params *p, const void *in, void *out, void *snaps,
            const void *noise, ssf_stats *st, ssf_trace *tr, long *launches) {
    if (!emu_supported(N, precision)) return SSF_ERR_UNSUPPORTED;
    return precision == SSF_C128 ? run_t<double>(N, nrows, precision, p, in, out, snaps, noise, st, tr, launches)
                                 : run_t<float>(N, nrows, precision, p, in, out, snaps, noise, st, tr, launches);
}

void emu_philox(unsigned long long ctr_lo, unsigned long long ctr_hi, unsigned long long key, unsigned *out) {
    const ssf::Philox4 r = ssf::philox4x32_10(ctr_lo, ctr_hi, key);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}

void emu_gauss(long long n, unsigned row, unsigned span, unsigned long long seed, double sigma, double *re, double *im) {
    for (long long i = 0; i < n; ++i) ssf::gauss_pair((unsigned long long)i, row, span, seed, sigma, re[i], im[i]);
}

int emu_overlap_save(int64_t sigLen, int nrows, int lg, int K, const void *Hfft, const void *in, void *out) {
    using namespace ssf::fused;
    using Cc = cx<double>;
    const int nfft = 1 << lg, d = nfft - K + 1;
    std::vector<Cc> Hs((size_t)nfft);
    for (int i = 0; i < nfft; ++i) {
        Hs[(size_t)i].re = ((const Cc *)Hfft)[i].re / nfft;
        Hs[(size_t)i].im = ((const Cc *)Hfft)[i].im / nfft;
    }
    ols_permute_filter(Hs.data(), lg);
    OlsArgs<double> a{};
    a.in = (const Cc *)in;
    a.out = (Cc *)out;
    a.H = Hs.data();
    a.sigLen = sigLen;
    a.njobs = ((sigLen + K - 1 + d - 1) / d) * nrows;
    a.nrows = nrows;
    a.log2nfft = lg;
    a.d = d;
    a.discard = K - 1;
    a.D = (K - 1) / 2;
    ols_defaults(a);
    const OlsLaunch o = ols_launch(lg, nrows, a.njobs);
    ols_dispatch(o, [&](auto lgc, auto cc) {
        run_grid((int)o.grid, o.threads, o.lds_bytes, [&](EmuCtx &c) { ols_body<double, decltype(lgc)::value, decltype(cc)::value>(c, a); });
    });
    return 0;
}

static long g_rx_launches = 0;
long emu_rx_launches() { return g_rx_launches; }                   // kernel launches of the last emu_rx_run
int emu_rx_run(int mode, int64_t N, int nmodes, const ssf_rx_params *p, const void *in0, const void *lo, const double *un, void *out) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    int rc = core.run(mode, N, nmodes, *p, in0, lo, un, out);
    g_rx_launches = be.launches;
    if (rc) fprintf(stderr, "emu: %s\n", core.err.c_str());
    return rc;
}
int emu_rx_chain(int64_t N, const ssf_rx_params *p, const void *Es, const void *Elo, const void *taps, int ntaps, int SpSin, int dec,
                 const void *edcH, int edcK, int edc_nfft, void *out, int32_t *sd) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    int rc = core.chain(N, *p, Es, Elo, taps, ntaps, SpSin, dec, edcH, edcK, edc_nfft, out, sd);
    g_rx_launches = be.launches;
    if (rc) fprintf(stderr, "emu: %s\n", core.err.c_str());
    return rc;
}
int emu_fir_nfft(int K) { return ssf::rx::fir_nfft(K); }                   // (the block-size rule: tests compare it with models._ols_block)
int emu_fir(int64_t sigLen, int ncols, int ntaps, const void *taps, const void *in, void *out) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.fir(sigLen, ncols, ntaps, taps, in, out);
}
int emu_fir_long(int64_t inLen, int64_t outLen, int ncols, int64_t ntaps, const void *taps, int64_t shift, const void *in, void *out) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.fir_long(inLen, outLen, ncols, ntaps, taps, shift, in, out);
}
int emu_nlin_phase_rot(int64_t n, double gamma, const void *Ex, const void *Ey, const double *Pch, double *phi) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.nlin_phase(n, gamma, Ex, Ey, Pch, phi);
}
int emu_convergence_condition(int64_t n, const void *xfd, const void *yfd, const void *xc, const void *yc, double *lim) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.convergence(n, xfd, yfd, xc, yc, lim);
}
int emu_optics(int op, int64_t n, int ncols, double p0, double p1, unsigned long long seed, unsigned row0, const void *a, const void *b,
               void *o0, void *o1) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.optics(op, n, ncols, p0, p1, seed, row0, a, b, o0, o1);
}
int emu_delay(int64_t N, double delay, double Fs, const void *in, void *out) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.delay(N, delay, Fs, in, out);
}
// standalone mixed-radix transform of `rows` rows of length L (complex128): dir < 0 forward (output in
// natural order via mix_bin), dir > 0: forward then inverse (round trip, unscaled: x * L)
// plan_threads: the thread count the radix plan is made for (0: largest radix first)
int emu_mixed_fft_t(int L, int rows, int dir, int plan_threads, const void *in, void *out) {
    using namespace ssf::fused;
    using Cc = cx<double>;
    MixPlan p;
    if (!mix_make_plan(L, &p, plan_threads)) return SSF_ERR_UNSUPPORTED;
    const Cc *src = (const Cc *)in;
    Cc *dst = (Cc *)out;
    run_grid(rows, 128, (size_t)L * sizeof(Cc), [&](EmuCtx &c) {
        Cc *x = (Cc *)c.lds;
        for (int i = c.tid; i < L; i += c.nthreads) x[i] = src[(size_t)c.bid * L + i];
        c.sync();
        mix_dif<-1>(c, p, c.tid, c.nthreads, x);
        if (dir > 0) {
            mix_dit<+1>(c, p, c.tid, c.nthreads, x);
            for (int i = c.tid; i < L; i += c.nthreads) dst[(size_t)c.bid * L + i] = x[i];
        } else {
            for (int i = c.tid; i < L; i += c.nthreads) dst[(size_t)c.bid * L + mix_bin(p, i)] = x[i];
        }
    });
    return 0;
}

int emu_mixed_fft(int L, int rows, int dir, const void *in, void *out) { return emu_mixed_fft_t(L, rows, dir, 0, in, out); }
// the radix plan mix_make_plan picks for a row of L values transformed by `threads` threads: radices[0..n-1], n returned (0: none)
int emu_mix_plan(int L, int threads, int *radices) {
    ssf::fused::MixPlan p;
    if (!ssf::fused::mix_make_plan(L, &p, threads)) return 0;
    for (int i = 0; i < p.npass; ++i) radices[i] = p.r[i];
    return p.npass;
}

int emu_wdm_tx(const ssf_tx_params *p, const void *symbols, const double *taps, const double *phi, const double *amp,
               const double *deltaF, void *out, double *power_out) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    int rc = core.wdm_tx(*p, symbols, taps, phi, amp, deltaF, out, power_out);
    if (rc) fprintf(stderr, "emu: %s\n", core.err.c_str());
    return rc;
}
int emu_decimate(int64_t N, int ncols, int SpSin, int dec, const void *in, void *out, int32_t *sd) {
    EmuBackend be;
    ssf::rx::RxCore<EmuBackend> core(be);
    return core.decimate(N, ncols, SpSin, dec, in, out, sd);
}

int emu_rows_lin(int64_t N, int nrows, int precision, const void *in, void *out, double hzh, double lin_a, double lin_b, double w_scale) {
    return precision == SSF_C128 ? rows_lin_t<double>(N, nrows, in, out, hzh, lin_a, lin_b, w_scale)
                                 : rows_lin_t<float>(N, nrows, in, out, hzh, lin_a, lin_b, w_scale);
}

int emu_linear_channel(int64_t N, int nrows, int precision, double Fs, double Fc, double alpha, double D, double L,
                       const void *in, void *out) {
    if (!emu_supported(N, precision)) return SSF_ERR_UNSUPPORTED;
    return precision == SSF_C128 ? lin_t<double>(N, nrows, precision, Fs, Fc, alpha, D, L, in, out)
                                 : lin_t<float>(N, nrows, precision, Fs, Fc, alpha, D, L, in, out);
}

}  // extern "C"
