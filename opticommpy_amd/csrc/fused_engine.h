// fused_engine.h -- host side of the fused radix-2^n engine, templated on a Backend so the
// same control code drives real HIP launches (engine_fused_impl.h) and the CPU emulator
// (tests/emu).  No host<->device synchronisation inside a step: the data-dependent control
// flow (iteration count, adaptive step, span end) lives in a device-resident Ctrl block that
// every launch reads and forwards; the host only enqueues a uniform launch sequence
// Col [Row, Col]* in chunks and looks at the Ctrl block between chunks.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "fused_kernels.h"
#include "ssf_derived.h"

namespace ssf {
namespace fused {

// Tuning knobs of the A/B experiments (forced splits, values per thread, persistent kernels, ...) are read from the environment
// only by experiment builds (-DSSF_EXPERIMENTS=1: `make exp`, the CPU emulator of tests/emu); the product library reads
// SSF_C64_PACKED, SSF_MGPU_LANES and SSF_MGPU_BATCH and nothing else.
#ifndef SSF_EXPERIMENTS
#define SSF_EXPERIMENTS 0
#endif
// (static: one copy per translation unit -- the experiment library links units built with and without SSF_EXPERIMENTS, and a merged
//  inline function would be whichever copy the linker keeps; the same for the split rules below, which call it)
static inline const char *tune_env(const char *name) {
#if SSF_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

struct Split {
    int l1, l2;   // log2 N1 (column length), log2 N2 (row length)
};

// How N = 2^m is split.  Columns want >= 256 B contiguous per row of a tile
// (C = 4096/N1 columns x sizeof(complex)); rows are bounded by the 160 KiB LDS.
static inline bool choose_split(int log2N, int precision, Split *s, bool packed = false) {
    if (log2N < 8) return false;
    if (const char *e = tune_env("SSF_SPLIT_L1")) {        // tuning knob: force log2 N1
        const int l1 = std::atoi(e);
        if (l1 >= 4 && log2N - l1 >= 4 && log2N - l1 <= 14) {
            s->l1 = l1;
            s->l2 = log2N - l1;
            return true;
        }
    }
    const bool dbl = precision == SSF_C128;
    const int l1pref = 8, l2max = dbl ? 13 : 14;     // (measured: 8 is best for both precisions at 2^20)
    int l1 = std::min(l1pref, log2N / 2);
    int l2 = log2N - l1;
    if (packed && l2 > 12) {                          // packed pairs (16-byte elements): rows of 4096 are three radix-16 passes and
        s->l1 = std::min(log2N - 12, 10);             // two workgroups per CU, rows of 8192 four passes and one: measured at 2^22
        s->l2 = log2N - s->l1;                        // row launch 59 -> 46 us, column launch (1024 long, 64-B segments) 60 -> 63 us
        return s->l2 <= 13;                           // (2^23: 1024 x 8192; beyond that the pairs do not fit the LDS rows)
    }
    const int l2soft = l2max - 1;          // prefer two workgroups per CU
    if (l2 > l2soft) {
        l1 = std::min(log2N - l2soft, l1pref + 1);
        l2 = log2N - l1;
    }
    if (l2 > l2max) {                      // the longest fields (2^23 complex128, 2^24 complex64): rows as long as the LDS takes,
        l2 = l2max;                        // columns of up to 1024 (64-B row segments: slower per sample, but the hand-written
        l1 = log2N - l2;                   // kernels then cover every Bluestein length up to N = 2^22 / 2^23 as well)
    }
    if (l1 > 10 || l1 < 4 || l2 < 4) return false;
    s->l1 = l1;
    s->l2 = l2;
    return true;
}

// Lengths with factors 3 and 5 (N = 2^a * m, m odd and 5-smooth -- notebook lengths are SpS x Nsymbols):
// the column length stays a power of two (2^8 if possible, else 2^9, 2^7, 2^10, 2^6 ... 2^4 out of the 2^a), the rest is the row length, transformed
// by the mixed-radix row kernel (mixed_fft.h).  Rows of up to 8192 values (16 per thread, 512 threads, one row per workgroup:
// 132 KiB of LDS in double precision).
constexpr int64_t kMixMaxRow = 8192;
static inline bool choose_mixed_split(int64_t N, int precision, int *l1, int *N2) {
    (void)precision;
    if (N < 1 || (N & (N - 1)) == 0) return false;
    int a = 0;
    int64_t m = N;
    while (m % 2 == 0) {
        m /= 2;
        ++a;
    }
    int64_t rest = m;
    for (int q : {3, 5})
        while (rest % q == 0) rest /= q;
    if (rest != 1 || a < 4) return false;
    if (const char *e = tune_env("SSF_MIX_L1")) {               // tuning knob: force log2 N1
        const int l = std::atoi(e);
        const int64_t n2 = N >> l;
        MixPlan mp;
        if (l >= 4 && l <= std::min(a, 10) && n2 >= 64 && n2 <= kMixMaxRow && mix_make_plan((int)n2, &mp)) {
            *l1 = l;
            *N2 = (int)n2;
            return true;
        }
    }
    // 8, 9, 7, 10: measured (tools/exp/mix_split_sweep.py): 2^8 columns beat 2^9 by 8-13 %, 2^10 loses 15-25 %; 6, 5, 4: lengths with
    // few factors of two (12 000 = 2^5 x 375, 6000 = 2^4 x 375): short columns, still far cheaper than a Bluestein transform
    for (int l : {8, 9, 7, 10, 6, 5, 4}) {
        if (l > a) continue;
        const int64_t n2 = N >> l;
        MixPlan mp;
        if (n2 >= 64 && n2 <= kMixMaxRow && mix_make_plan((int)n2, &mp)) {
            *l1 = l;
            *N2 = (int)n2;
            return true;
        }
    }
    return false;
}

// Lengths 2^a 3^b 5^c that choose_mixed_split does not take -- fewer than four factors of two, or a row above kMixMaxRow whatever
// power of two goes into the columns (2 000 000 = 2^7 x 15 625) -- split as N1 x N2 with BOTH factors mixed-radix: the column stage
// is col_mixed_body (a tile of C columns x N1 of both polarisations in LDS), the row stage the mixed-radix rows as before.
// Preference: both stages at two workgroups per CU, then wide global segments (C), then columns near 256.
inline size_t mix2_col_lds(int N1, int C, int npol, int elem_bytes) { return (size_t)kColMixScratch + (size_t)npol * C * N1 * elem_bytes; }
// (force_n1 / force_c: the experiment knob SSF_MIX2 = "N1,C", parsed by the CALLER -- this function is emitted once per library, and
//  the linker may keep the copy of a translation unit that was built without the experiment switches)
static inline bool choose_mixed2_split(int64_t N, int precision, int *N1o, int *N2o, int *Co, int force_n1 = 0, int force_c = 0) {
    if (N < 4096 || (N & (N - 1)) == 0) return false;
    int64_t rest = N;
    for (int q : {2, 3, 5})
        while (rest % q == 0) rest /= q;
    if (rest != 1) return false;
    const int s = precision == SSF_C128 ? 16 : 8;
    double best = 1e300;
    for (int n1 = 16; n1 <= kMix2MaxCol; ++n1) {
        if (N % n1 || (force_n1 && n1 != force_n1)) continue;
        const int64_t n2 = N / n1;
        if (n2 < 64 || n2 > kMixMaxRow) continue;
        MixPlan mp;
        if (!mix_make_plan(n1, &mp) || !mix_make_plan((int)n2, &mp)) continue;
        for (int C : {8, 4, 2}) {
            if (force_c && C != force_c) continue;
            const size_t cl = mix2_col_lds(n1, C, 2, s), rl = 4096 + (size_t)n2 * s;
            if (cl > 156 * 1024) continue;
            double score = (cl <= 80 * 1024 ? 0.0 : 4.0) + ((rl <= 72 * 1024 && n2 <= 4096) ? 0.0 : 3.0) + (C == 8 ? 0.0 : C == 4 ? 1.0 : 3.0) +
                           0.5 * std::fabs(std::log2((double)n1 / 256.0));
            // short fields: both launches should still put a workgroup on every CU (200 000 = 125 x 1600: C = 4, 400 column
            // workgroups, 7 127 steps/s; C = 8, 200 workgroups, 6 724; 250 x 800 with C = 8: 5 681 -- profiles/r6_mix2_short.txt)
            const double col_wgs = (double)((n2 + C - 1) / C), row_wgs = (2 * n1 >= 1024 && n2 <= 2048) ? (double)n1 : 2.0 * n1;
            score += 4.0 * std::max(0.0, std::log2(256.0 / col_wgs)) + 4.0 * std::max(0.0, std::log2(256.0 / row_wgs));
            // ... and a column length whose passes take one round of butterflies each with the threads a transform gets (125 = 5 x 5 x 5
            // with 32 threads: column launch 19.3 us; 200 = 5 x 5 x 8, two rounds in two passes: 24.2 us)
            MixPlan cp;
            const int tcol = 256 / (2 * C);
            if (mix_make_plan(n1, &cp, tcol, false))
                for (int i = 0; i < cp.npass; ++i) score += 0.5 * (double)((n1 / cp.r[i] + tcol - 1) / tcol - 1);
            if (score < best) {
                best = score;
                *N1o = n1;
                *N2o = (int)n2;
                *Co = C;
            }
        }
    }
    return best < 1e300;
}

// The split of a length that is not a power of two.  l1 > 0: power-of-two columns (2^l1) x mixed-radix rows (N2); N1 > 0: both factors
// mixed-radix (column stage col_mixed_body, C columns per workgroup).  f1 / fc: the experiment knob SSF_MIX2 (force the second kind).
// A length with fewer than seven factors of two (200 000 = 2^6 x 3125) takes the second kind although the first exists: 2 x 64 rows
// of 3125 leave most of 256 CUs idle, 125 x 1600 fills them: 5 293 -> 7 127 steps/s (profiles/r6_mix2_short.txt).
static inline bool choose_nonpow2_split(int64_t N, int precision, int f1, int fc, bool l1_forced, int *l1, int *N1, int *N2, int *C) {
    *l1 = *N1 = *N2 = *C = 0;
    if (f1 > 0 && choose_mixed2_split(N, precision, N1, N2, C, f1, fc)) return true;
    if (!choose_mixed_split(N, precision, l1, N2)) {
        *l1 = *N2 = 0;
        return choose_mixed2_split(N, precision, N1, N2, C);
    }
    if (*l1 < 7 && !l1_forced) {
        int n1 = 0, n2 = 0, c = 0;
        if (choose_mixed2_split(N, precision, &n1, &n2, &c)) {
            *l1 = 0;
            *N1 = n1;
            *N2 = n2;
            *C = c;
        }
    }
    return true;
}

// T = double / float: one complex row per polarisation (nrows rows);  T = pf2: the packed pair of the complex64
// Manakov path (fused_core.h), nrows = number of polarisation PAIRS, 16-byte elements -- created and driven by the
// float core (run_manakov_packed), only its Manakov span loop is used.
template <typename T, class Backend> class FusedCore {
  public:
    using C = cx<T>;
    using S = scalar_t<T>;
    static constexpr bool kPacked = sizeof(T) != sizeof(S);
    Backend &be;
    int64_t N;
    int nrows, log2N;
    Split sp;
    int N2mix = 0;               // > 0: row length of the mixed-radix path (then sp.l2 is unused)
    int mix_rows = 1;            // rows per workgroup there
    MixPlan mix_plan{};          // radices of the row passes, chosen for the threads a row gets
    cx<double> *wtab = nullptr;  // cis(-2 pi k / N2mix)
    int N1mix = 0, mix_cols = 0; // > 0: column length / columns per workgroup of the mixed-radix COLUMN stage (col_mixed_body; then sp.l1 is unused too)
    MixPlan mix_plan1{};         // radices of its passes
    cx<double> *wtab1 = nullptr; // cis(-2 pi k / N1mix)
    int n1() const { return N1mix ? N1mix : 1 << sp.l1; }      // column length
    int row_tw_off = 0;          // LDS offset of the row workgroups' twiddle table (single precision; 0 = none)
    size_t field_bytes;
    C *G = nullptr, *T0 = nullptr, *T1 = nullptr, *Ehd = nullptr, *noise_d = nullptr;
    S *P = nullptr, *Theta = nullptr;
    bool own_G = true;           // the packed core works in the float core's exchange buffer
    void *cpl_comm = nullptr;    // coupled batch across ranks (ssf_set_coupling_comm): the communicator ...
    double *cpl_work = nullptr;  // ... and 8 + 5 x ranks doubles: [0, 5) this rank's / the reduced values, [8, ...) the gathered ones
    std::unique_ptr<FusedCore<pf2, Backend>> pk;     // float core: the packed-pair Manakov pipeline (created on first use)
    bool use_packed = true;      // SSF_C64_PACKED=0: complex64 Manakov on the one-row-per-polarisation kernels (A/B runs)
    bool lim0_bound = true;       // SSF_LIM0_BOUND=0: always evaluate lim_0 on all samples (A/B runs)
    Ctrl *ctrl = nullptr;        // [2]
    LinOp *linops = nullptr;     // [2]
    unsigned *gbar = nullptr;    // persistent span kernel: arrivals, generation, abort
    double *part = nullptr;      // 3 * npart_max
    double *tr_hz = nullptr, *tr_lim = nullptr;
    int *tr_it = nullptr;
    long long tr_cap = 0;
    int tr_maxIter = 0;
    std::vector<C *> snaps;
    int row_v = 16;              // values per thread of the radix-2^n row kernel (SSF_ROW_V=8: 128-register kernels)
    int col_v = 16;              // values per thread of the column kernels (SSF_COL_V=8)
    bool underfilled = false;    // the field does not fill the chip: 8-value kernels, one row per workgroup (init)
    int lanes_hint = 1;          // plans that share the GPU concurrently (ssf_plan_set_lanes): > 1 turns the phase priorities off
    // Stage-specialised column kernels (H | ADV | FIN, + the general one for the rare stages) along the predicted stage sequence
    // (run_span; fused_kernels.h: stage_group): +1.5 ... 2.3 % steps/s at config 2, 3 of 3 (profiles/r5_ab_col_split.txt).2 = a
    // transforming kernel (H, ADV, rare) and an observing one (FIN).  SSF_COL_SPLIT in experiment builds.
    int col_split = 1;           // (experiment builds: SSF_COL_SPLIT=0 -- the general kernel at every launch)
    bool split_ok = true;        // ... until rare stages turn out to be the rule in this call (then the general kernel for the rest of it
    int split_penalty = 0;       //     and for the next kSplitPenalty calls of the plan)
    static constexpr int kSplitPenalty = 8;
    int hint_n_it = 0;           // iterations of the latest finished step of the previous call (0: unknown)
    int cur = 0;                 // which of T0/T1 holds the current field
    unsigned seq = 0;
    // launch geometry
    int row_block, row_grid, col_block_mk, col_grid_mk, col_block_1, col_grid_1, npart_max;
    size_t row_lds, col_lds_mk, col_lds_1;
    std::string err;

    std::vector<ssf_stats> ustats;   // per unit, accumulated since the last upload (ssf_get_unit_stats)
    int units = 1;               // the rows form `units` independent fields: own control block, partial sums, step sizes and
                                 // convergence decisions per unit, every launch carries all of them (grid.y = units)
    int npairs() const { return kPacked ? nrows : std::max(nrows / 2, 1); }
    int rows_u() const { return nrows / units; }                 // rows of one unit
    int pairs_u() const { return std::max(npairs() / units, 1); }
    FusedCore(Backend &b, int64_t N_, int nrows_, int precision, void *borrowed_G = nullptr, int units_ = 1)
        : be(b), N(N_), nrows(nrows_), units(units_ > 0 ? units_ : 1) {
        if (borrowed_G) {
            G = (C *)borrowed_G;
            own_G = false;
        }
        log2N = 0;
        while ((1ll << log2N) < N) ++log2N;
        if ((N & (N - 1)) == 0) {
            choose_split(log2N, precision, &sp, kPacked);
        } else {
            sp.l1 = sp.l2 = 0;
            // (experiment builds: SSF_MIX2="N1,C" puts a length the radix-2^n columns would take on the mixed-radix column stage)
            int f1 = 0, fc = 0;
            if (const char *e = tune_env("SSF_MIX2")) {
                f1 = std::atoi(e);
                if (const char *q = std::strchr(e, ',')) fc = std::atoi(q + 1);
            }
            if (!choose_nonpow2_split(N, precision, f1, fc, tune_env("SSF_MIX_L1") != nullptr, &sp.l1, &N1mix, &N2mix, &mix_cols))
                sp.l1 = N1mix = N2mix = mix_cols = 0;
        }
        field_bytes = sizeof(C) * (size_t)N * (size_t)nrows;
    }

    // npol = 2: a workgroup carries both rows of a polarisation pair (x threads | y threads)
    void col_geometry(int groups, int npol, int *block, int *grid, size_t *lds) const {
        if (N1mix) {                                           // mixed-radix columns: the tile lives in LDS (col_mixed_body)
            *block = 256;
            *grid = groups * ((N2mix + mix_cols - 1) / mix_cols);
            *lds = mix2_col_lds(N1mix, mix_cols, npol, (int)sizeof(C));
            return;
        }
        const int tpf = (1 << sp.l1) / col_v, N2 = N2mix ? N2mix : 1 << sp.l2;
        int half = col_v == 8 ? 512 / npol : 256;              // (eight values per thread: 512-thread workgroups, two per CU)
        if (const char *e = tune_env("SSF_COL_HALF")) {        // tuning knob: threads per polarisation row
            const int h = std::atoi(e);
            if (h >= tpf && h <= 512 && (h & (h - 1)) == 0) half = h;
        }
        // (packed pairs, columns of 1024: 512-thread workgroups with 128-B row segments measured slower than 256 threads with
        //  64-B segments: column launch 67.5 vs 63.2 us at 2^22 -- one 139 KiB workgroup per CU against two of 70 KiB)
        while (half / tpf > N2 && half > tpf) half >>= 1;    // (a power of two also when N2 is not)
        // a grid that only just covers the 256 CUs leaves every CU with one lock-stepped workgroup:
        // prefer two smaller independent ones (measured +3 % at N = 2^20) while rows stay >= 128 B wide
        // ... and below 256 workgroups (2^16 ... 2^18 samples, one field) even 64-B segments pay: the launch is a latency chain,
        // more and smaller workgroups shorten it (measured, gpurun_out/r3k: 2^16 9 222 -> 9 720 steps/s, 2^18 8 188 -> 8 531)
        // (per unit, not per launch: a batch of independent units keeps the geometry -- and so the partial sums -- of the single plan)
        auto wgs = [&](int h) { return (long long)groups * (N2 / (h / tpf)); };
        while (!tune_env("SSF_COL_HALF") && half > 64 &&
               ((wgs(half) < 512 && (half / tpf) * sizeof(C) > 128) || (wgs(half) < 256 && (half / tpf) * sizeof(C) > 64)))
            half >>= 1;
        const int Cc = half / tpf;
        *block = half * npol;
        *grid = groups * ((N2 + Cc - 1) / Cc);
        *lds = std::max((size_t)npol * Cc * lds_col_stride(1 << sp.l1, Cc, (int)sizeof(C)) * sizeof(C), (size_t)half * npol * 16 + 1024);
    }

    int init() {
        if (const char *e = tune_env("SSF_LIM0_BOUND")) lim0_bound = std::atoi(e) != 0;
        if (const char *e = tune_env("SSF_COL_SPLIT")) col_split = std::atoi(e);
        // Fields that do not fill the chip -- fewer than two 16-value waves per SIMD: rows x N <= 2^20 values per unit, i.e. up to
        // 2^19 samples for a complex128 pair, 2^20 for a packed complex64 pair -- run on the 8-value kernels (twice the waves
        // and workgroups, shorter dependent chains per thread) with one row per workgroup: measured +7 ... +75 % there
        // (profiles/r3_underfilled_sweep.txt: 2^12 7 138 -> 12 489 steps/s, 2^16 9 739 -> 11 940, 2^18 8 557 -> 10 219, 2^19 7 462 ->
        // 7 992; complex64 2^20 6 080 -> 6 949; config 1 51 422 -> 60 329), while a field that fills the chip is faster on the
        // 16-value kernels (section 3.16).  Columns of 128 keep 16 values (8 values: three passes instead of two, -13 % at 2^14 /
        // 2^15).  Decided per unit, so a batch of independent units has the geometry -- and the arithmetic -- of the single plan.
        underfilled = (double)rows_u() * (double)N <= 1048576.0;
        col_v = underfilled && sp.l1 != 7 ? 8 : 16;
        if (const char *e = tune_env("SSF_COL_V")) col_v = std::atoi(e) == 8 ? 8 : 16;
        if (N2mix || sp.l1 < 6 || sp.l1 > 10) col_v = 16;      // (ragged tiles / very short or very long columns: 16-value kernels only)
        if (N1mix && !mix_make_plan(N1mix, &mix_plan1, 256 / (2 * mix_cols), false)) {
            err = "fused engine: no pass plan for the column length";
            return SSF_ERR_UNSUPPORTED;
        }
        if (const char *e = N1mix ? tune_env("SSF_MIX_PLAN1") : nullptr) {   // experiments: the column passes, "10,5,10"
            int r[kMixMaxPass], n = 0;
            for (const char *q = e; *q && n < kMixMaxPass;) {
                r[n++] = std::atoi(q);
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            MixPlan mp;
            if (mix_plan_from_radices(N1mix, r, n, &mp)) mix_plan1 = mp;
        }
        const int64_t nfft = (int64_t)rows_u() * n1();         // row transforms of one unit
        if (N2mix) {               // rows in LDS after 4 KiB of scratch; 128 threads per row while 16 values per thread suffice
            int tpr = 128;
            if (const char *e = tune_env("SSF_MIX_TPR")) tpr = std::max(64, std::min(1024, std::atoi(e)));
            while (16 * tpr < N2mix) tpr *= 2;
            mix_rows = std::max(1, 256 / tpr);
            while (nfft % mix_rows) mix_rows >>= 1;
            while (mix_rows > 1 && 4096 + (size_t)mix_rows * N2mix * sizeof(C) > 72 * 1024) mix_rows >>= 1;   // two workgroups per CU
            while (mix_rows > 1 && nfft / mix_rows < 512) {           // few rows: a workgroup per row rather than idle CUs
                mix_rows >>= 1;                                       // (240 000 = 256 rows of 1875: +9 % measured)
                tpr *= 2;
            }
            row_block = tpr * mix_rows;
            row_grid = (int)(nfft / mix_rows);
            row_lds = 4096 + (size_t)mix_rows * N2mix * sizeof(C);
            mix_make_plan(N2mix, &mix_plan, tpr);
            if (const char *e = tune_env("SSF_MIX_PLAN")) {       // experiments: "15,5,5,5"
                int r[kMixMaxPass], n = 0;
                for (const char *q = e; *q && n < kMixMaxPass;) {
                    r[n++] = std::atoi(q);
                    while (*q && *q != ',') ++q;
                    if (*q == ',') ++q;
                }
                MixPlan mp;
                if (mix_plan_from_radices(N2mix, r, n, &mp) && mp.r[mp.npass - 1] <= kMixMaxOpRadix) mix_plan = mp;
            }
        } else {
            row_v = underfilled ? 8 : 16;
            if (const char *e = tune_env("SSF_ROW_V")) row_v = std::atoi(e) == 8 ? 8 : 16;
            if (sp.l2 < 6) row_v = 16;
            const int tpf2 = (1 << sp.l2) / row_v, wg = row_v == 8 ? 512 : 256;
            int fpw = tpf2 >= wg ? 1 : wg / tpf2;             // row transforms per workgroup
            // under-filled chip: a workgroup per row.  A batch of independent units fills the chip by itself: 256-thread workgroups
            // there (rows per workgroup do not enter a row's arithmetic, so the units' fields stay those of the single plan)
            if (underfilled && row_v == 8) fpw = units > 1 ? std::max(1, std::min(fpw, 256 / std::max(tpf2, 1))) : 1;
            if (const char *e = tune_env("SSF_ROW_FPW")) fpw = std::max(1, std::min(fpw, std::atoi(e)));   // tuning knob
            while (nfft % fpw) fpw >>= 1;                      // (nrows need not be a power of two)
            row_block = fpw * tpf2;
            row_grid = (int)(nfft / fpw);
            row_lds = std::max((size_t)fpw * lds_slots_per_fft(1 << sp.l2) * sizeof(C), (size_t)row_block * 16 + 2048);
            if ((SSF_TW_TAB & 1) && sizeof(S) == 4) {          // single precision: the workgroup's twiddle table behind the transform
                row_tw_off = (int)((row_lds + 15) / 16 * 16);  // area (fused_kernels.h: TwSrc; double precision and the column
                row_lds = (size_t)row_tw_off + kTwLdsBytes;    // stages are faster without: profiles/r4_ab_twiddle_tables.txt)
            }
        }
        if (const char *e = std::getenv("SSF_C64_PACKED")) use_packed = std::atoi(e) != 0;
        col_geometry(pairs_u(), kPacked ? 1 : 2, &col_block_mk, &col_grid_mk, &col_lds_mk);
        col_geometry(rows_u(), 1, &col_block_1, &col_grid_1, &col_lds_1);
        npart_max = std::max(col_grid_mk, col_grid_1);
        if (own_G && !(G = (C *)be.alloc(field_bytes))) return oom();
        if (!(T0 = (C *)be.alloc(field_bytes))) return oom();
        if (kPacked && mk_buffers()) return SSF_ERR_OOM;      // (the other cores allocate them when a Manakov run needs them)
        if (!(ctrl = (Ctrl *)be.alloc(2 * (size_t)units * sizeof(Ctrl)))) return oom();      // [launch parity][unit]
        if (!(linops = (LinOp *)be.alloc(2 * sizeof(LinOp)))) return oom();
        if (!(gbar = (unsigned *)be.alloc(8 * sizeof(unsigned)))) return oom();
        if (!(part = (double *)be.alloc(sizeof(double) * 5 * (size_t)units * (size_t)npart_max))) return oom();   // [array][unit][npart_max]
        if (N2mix) {
            if (!(wtab = (cx<double> *)be.alloc(sizeof(cx<double>) * (size_t)N2mix))) return oom();
            std::vector<cx<double>> w((size_t)N2mix);
            for (int q = 0; q < N2mix; ++q) {                  // octant-reduced angles: exact symmetries, < 1 ulp
                const double a = -2.0 * 3.14159265358979323846 * (double)q / (double)N2mix;
                w[(size_t)q].re = std::cos(a);
                w[(size_t)q].im = std::sin(a);
            }
            be.h2d(wtab, w.data(), sizeof(cx<double>) * (size_t)N2mix);
        }
        if (N1mix) {
            if (!(wtab1 = (cx<double> *)be.alloc(sizeof(cx<double>) * (size_t)N1mix))) return oom();
            std::vector<cx<double>> w((size_t)N1mix);
            for (int q = 0; q < N1mix; ++q) {
                const double a = -2.0 * 3.14159265358979323846 * (double)q / (double)N1mix;
                w[(size_t)q].re = std::cos(a);
                w[(size_t)q].im = std::sin(a);
            }
            be.h2d(wtab1, w.data(), sizeof(cx<double>) * (size_t)N1mix);
        }
        be.prepare(row_lds, std::max(col_lds_mk, col_lds_1));
        return SSF_OK;
    }
    // second time-domain field, E_hd, Pch (two buffers) and the phase array: only the Manakov pipeline uses them
    int mk_buffers() {
        if (T1) return SSF_OK;
        if (!(T1 = (C *)be.alloc(field_bytes)) || !(Ehd = (C *)be.alloc(field_bytes))) return oom();
        if (!(P = (S *)be.alloc(2 * sizeof(S) * (size_t)N * (size_t)npairs()))) return oom();
        if (!(Theta = (S *)be.alloc(sizeof(S) * (size_t)N * (size_t)npairs()))) return oom();
        return SSF_OK;
    }
    int oom() {
        err = "out of memory allocating fused-engine buffers: " + be.last_error();
        return SSF_ERR_OOM;
    }
    ~FusedCore() {
        if (!own_G) G = nullptr;
        if (cpl_work) be.free(cpl_work);
        for (void *p : {(void *)G, (void *)T0, (void *)T1, (void *)Ehd, (void *)P, (void *)Theta, (void *)ctrl, (void *)linops, (void *)gbar,
                        (void *)part, (void *)wtab, (void *)wtab1, (void *)tr_hz, (void *)tr_lim, (void *)tr_it, (void *)noise_d})
            if (p) be.free(p);
        for (C *s : snaps) be.free(s);
    }

    C *Tcur() { return cur ? T1 : T0; }
    int cpl_nranks = 0;
    int set_couple(void *comm, int nranks) {
        if (cpl_work) be.free(cpl_work);
        cpl_work = nullptr;
        cpl_comm = nullptr;
        cpl_nranks = 0;
        if (!comm) return SSF_OK;
        cpl_nranks = nranks;
        if (!(cpl_work = (double *)be.alloc(sizeof(double) * (size_t)(8 + 5 * nranks)))) return oom();
        cpl_comm = comm;
        return SSF_OK;
    }

    int upload(const void *field, bool aos) {
        if constexpr (kPacked) return SSF_ERR_UNSUPPORTED;
        cur = 0;
        be.h2d_big(aos ? G : T0, field, field_bytes);
        if (aos) be.aos_to_soa(T0, G, N, nrows);
        for (C *s : snaps) be.free(s);
        snaps.clear();
        n_sunk = 0;
        ustats.assign((size_t)units, ssf_stats{});
        if (pk) pk->ustats.assign((size_t)units, ssf_stats{});
        return be.ok() ? SSF_OK : hiperr();
    }
    // counters of one unit (the packed-pair core keeps them for complex64 Manakov runs)
    bool unit_stats(int u, ssf_stats *out) const {
        const std::vector<ssf_stats> &v = (pk && !pk->ustats.empty()) ? pk->ustats : ustats;
        if (u < 0 || u >= (int)v.size()) return false;
        *out = v[(size_t)u];
        return true;
    }
    int download(void *field, int which, bool aos) {
        if constexpr (kPacked) return SSF_ERR_UNSUPPORTED;
        const C *src = which < 0 ? Tcur() : snaps[(size_t)which];
        if (aos) {
            be.soa_to_aos(G, src, N, nrows);
            src = G;
        }
        be.d2h_big(field, src, field_bytes);
        return be.ok() ? SSF_OK : hiperr();
    }
    int hiperr() {
        err = be.last_error();
        return SSF_ERR_HIP;
    }

    // ---------------------------------------------------------------- launch helpers
    RowArgs<T> row_args() const {
        RowArgs<T> a{};
        a.G = G;
        a.log2N1 = sp.l1;
        a.log2N2 = sp.l2;
        a.nfft = (int)((int64_t)rows_u() * n1());
        a.N1mix = N1mix;
        a.u_elems = (long long)rows_u() * N;
        a.u_part = npart_max;
        a.N2 = N2mix ? N2mix : 1 << sp.l2;
        a.N = N;
        a.mixed = N2mix ? 1 : 0;
        if (N2mix) a.plan = mix_plan;
        a.wtab = wtab;
        a.rows_per_wg = mix_rows;
        a.vpt = row_v;
        a.prio = lanes_hint <= 1 ? 1 : 0;
        a.tw_off = N2mix ? 0 : row_tw_off;
        return a;
    }
    ColArgs<T> col_args(int npol, int mode) const {
        ColArgs<T> a{};
        a.G = G;
        a.T0 = T0;
        a.T1 = T1;
        a.Ehd = Ehd;
        a.P = P;
        a.Theta = Theta;
        a.log2N1 = sp.l1;
        a.log2N2 = sp.l2;
        a.N2 = N2mix;
        a.N = N;
        a.N1mix = N1mix;
        a.mix_cols = mix_cols;
        if (N1mix) a.plan1 = mix_plan1;
        a.wtab1 = wtab1;
        a.npol = npol;
        a.mode = mode;
        a.ngroups = pairs_u();
        a.vpt = col_v;
        a.prio = lanes_hint <= 1 ? 1 : 0;
        a.u_elems = (long long)rows_u() * N;
        a.u_part = npart_max;
        const size_t ps = (size_t)units * (size_t)npart_max;       // one array of partial sums: [unit][npart_max]
        a.pmax = part;
        a.pnum = part + ps;
        a.pden = part + 2 * ps;
        a.pnum0 = part + 3 * ps;
        a.pden0 = part + 4 * ps;
        return a;
    }
    void launch_row_lin(const LinOp *lin) {
        RowArgs<T> a = row_args();
        a.use_ctrl = 0;
        a.lin = lin;
        be.launch_row(a, row_grid, row_block, row_lds, units);
    }
    // fixed-kernel convolution (FusedConv): forward-only row stage / multiplier-array row stage
    void launch_row_conv(const C *harr, int fwd_only) {
        RowArgs<T> a = row_args();
        a.use_ctrl = 0;
        a.lin = nullptr;
        a.harr = harr;
        a.fwd_only = fwd_only;
        be.launch_row(a, row_grid, row_block, row_lds, units);
    }
    void launch_col_plain(int mode, C *timebuf, S g_hz) {
        if constexpr (kPacked) return;
        ColArgs<T> a = col_args(1, mode);
        a.T0 = timebuf;
        a.g_hz = g_hz;
        a.npart = col_grid_1;
        be.launch_col(a, col_grid_1, col_block_1, col_lds_1, units);
    }
    void launch_amp(C *E, S gain, const C *noise, double sigma = 0.0, unsigned long long seed = 0, unsigned span = 0,
                    unsigned row0 = 0) {
        if constexpr (!kPacked) {
            AmpArgs<S> a{};
            a.E = E;
            a.noise = noise;
            a.total = (long long)N * nrows;
            a.N = N;
            a.gain = gain;
            a.sigma = sigma;
            a.seed = seed;
            a.span = span;
            a.row0 = row0;
            be.launch_amp(a, 1024, 256);
        }
    }
    int n_sunk = 0;              // snapshots handed to the plan's sink since the last upload
    int snapshot() {
        if (be.sink_active()) {                      // streamed out (ssf_snapshots.h), not kept
            be.sink_capture(Tcur(), (long long)N, nrows);
            ++n_sunk;
            return be.ok() ? SSF_OK : hiperr();
        }
        C *s = (C *)be.alloc(field_bytes);
        if (!s) return oom();
        snaps.push_back(s);
        be.d2d(s, Tcur(), field_bytes);
        return SSF_OK;
    }
    static bool wants_snapshot(const ssf_params &p, int span) {
        for (int i = 0; i < p.n_save; ++i)
            if (p.save_spans[i] == span) return true;
        return false;
    }
    int amp_fwd(const ssf_params &p, const Derived &d, int span, int span_rel, const void *noise, double ideal_gain) {
        if (p.amp == SSF_AMP_EDFA) {
            const C *nz = nullptr;
            if (noise) {
                if (!noise_d && !(noise_d = (C *)be.alloc(field_bytes))) return oom();
                be.h2d_big(noise_d, (const char *)noise + (size_t)span_rel * field_bytes, field_bytes);
                nz = noise_d;
            }
            const bool dev_noise = !noise && p.rng_seed != 0;                     // devices.py:723-726
            launch_amp(Tcur(), (S)std::sqrt(d.G_lin), nz, dev_noise ? std::sqrt(d.p_noise / 2) : 0.0,
                       (unsigned long long)p.rng_seed, (unsigned)span, (unsigned)p.rng_row_offset);
        } else if (p.amp == SSF_AMP_IDEAL) {
            launch_amp(Tcur(), (S)ideal_gain, nullptr);
        }
        return SSF_OK;
    }

    // ---------------------------------------------------------------- scalar NLSE
    int run_nlse(const ssf_params &p, const Derived &d, int s0, int s1, const void *noise, ssf_stats *st) {
        if constexpr (kPacked) return SSF_ERR_UNSUPPORTED;
        const int nsteps = (int)std::floor(p.Lspan / p.hz);
        const double w2 = (d.w_scale / (double)N) * (d.w_scale / (double)N);
        LinOp lo[2];
        lo[0] = make_linop(p.hz / 2, d.lin_a, d.lin_b, w2, 1.0 / (double)N, log2N);
        lo[1] = make_linop(p.hz, d.lin_a, d.lin_b, w2, 1.0 / (double)N, log2N);     // lin * lin
        be.h2d(linops, lo, sizeof(lo));
        // small N: the whole span in one persistent launch (engine_fused_impl.h: k_nlse_span) when both stage grids fit the
        // CUs with 256-thread workgroups
        int pgrid = 0, prow = 0, pcol = 0, ptw = 0;
        size_t plds = 0;
        if constexpr (Backend::kCanPersist && !kPacked) {
            const int tpf1 = (1 << sp.l1) / 16, tpf2 = (1 << sp.l2) / 16, lim = be.persist_limit();
            if (!N2mix && tpf1 <= 256 && tpf2 <= 256 && nsteps >= 1 && lim > 0 && units == 1 && row_v == 16 && col_v == 16) {
                const int fpw = 256 / tpf2, Cc = 256 / tpf1;
                const int64_t nfft = (int64_t)nrows << sp.l1;
                if (nfft % fpw == 0 && (1 << sp.l2) % Cc == 0) {
                    prow = (int)(nfft / fpw);
                    pcol = nrows * ((1 << sp.l2) / Cc);
                    if (std::max(prow, pcol) <= lim) {
                        pgrid = std::max(prow, pcol);
                        plds = std::max((size_t)fpw * lds_slots_per_fft(1 << sp.l2), (size_t)Cc * lds_col_stride(1 << sp.l1, Cc, (int)sizeof(C))) * sizeof(C);
                        plds = std::max(plds, (size_t)256 * 16 + 2048);
                        if ((SSF_TW_TAB & 1) && sizeof(S) == 4) {       // the row stage's twiddle table (single precision)
                            ptw = (int)((plds + 15) / 16 * 16);
                            plds = (size_t)ptw + kTwLdsBytes;
                        }
                    }
                }
            }
        }
        for (int span = s0; span <= s1; ++span) {
            C *E = Tcur();
            if (pgrid > 0) {
                if constexpr (Backend::kCanPersist && !kPacked) {
                    SpanNlseArgs<T> a{};
                    a.row = row_args();
                    a.row.use_ctrl = 0;
                    a.row.vpt = 16;
                    a.row.tw_off = ptw;
                    a.col = col_args(1, CM_NLSE_STEP);
                    a.col.vpt = 16;
                    a.col.T0 = E;
                    a.col.g_hz = (S)(p.gamma * p.hz);
                    a.col.npart = pcol;
                    a.lin_half = linops + 0;
                    a.lin_full = linops + 1;
                    a.nsteps = nsteps;
                    a.row_grid = prow;
                    a.col_grid = pcol;
                    a.bar = gbar;
                    int rc = be.launch_nlse_span(a, pgrid, plds);
                    if (rc) return hiperr();
                }
            } else if (nsteps >= 1) {
                launch_col_plain(CM_NLSE_FIRST, E, (S)0);                              // channels.py:216
                launch_row_lin(linops + 0);
                for (int s = 1; s < nsteps; ++s) {
                    launch_col_plain(CM_NLSE_STEP, E, (S)(p.gamma * p.hz));
                    launch_row_lin(linops + 1);
                }
                launch_col_plain(CM_NLSE_STEP, E, (S)(p.gamma * p.hz));
                launch_row_lin(linops + 0);
                launch_col_plain(CM_NLSE_LAST, E, (S)0);                               // channels.py:232
            }
            int rc = amp_fwd(p, d, span, span - s0, noise, std::exp(d.alpha_lin / 2 * nsteps * p.hz));
            if (rc) return rc;
            if (wants_snapshot(p, span) && (rc = snapshot())) return rc;
            st->steps += nsteps;
            st->transforms += (int64_t)nrows * (2 * (int64_t)nsteps + 2);
        }
        be.sync();
        if (pgrid > 0) {                                               // a barrier that could not complete set the abort word
            unsigned flags[3] = {0, 0, 0};
            be.d2h(flags, gbar, sizeof(flags));
            if (flags[2]) {
                err = "persistent span kernel: grid barrier timed out (workgroups not co-resident?)";
                return SSF_ERR_STATE;
            }
        }
        return be.ok() ? SSF_OK : hiperr();
    }

    // ---------------------------------------------------------------- Manakov / DBP
    MkConst mk_const(const ssf_params &p, const Derived &d) const {
        MkConst k{};
        k.Lspan = p.Lspan;
        k.hz_fixed = p.hz;
        k.tol = p.tol;
        k.maxRot = p.maxNlinPhaseRot;
        k.c8g = d.c8g;
        k.sgn = p.direction >= 0 ? 1.0 : -1.0;
        k.lin_a = d.lin_a;
        k.lin_b = d.lin_b;
        k.w2 = (d.w_scale / (double)N) * (d.w_scale / (double)N);
        k.invN = 1.0 / (double)N;
        k.maxIter = p.maxIter;
        k.adaptive = p.nlprMethod ? 1 : 0;
        k.exact_lim0 = (p.maxIter == 1 || !lim0_bound) ? 1 : 0;      // (+ whenever a trace is recorded, run_manakov)
        k.log2N = log2N;
        k.trace_cap = tr_cap;
        k.tr_hz = tr_hz;
        k.tr_it = tr_it;
        k.tr_lim = tr_lim;
        return k;
    }
    void launch_mk_row(const MkConst &k) {
        RowArgs<T> a = row_args();
        a.use_ctrl = 1;
        a.cin = ctrl + (size_t)(seq & 1) * units;
        a.cout = ctrl + (size_t)((seq + 1) & 1) * units;
        a.k = k;
        const size_t ps = (size_t)units * (size_t)npart_max;
        a.pmax = part;
        a.pnum = part + ps;
        a.pden = part + 2 * ps;
        a.pnum0 = part + 3 * ps;
        a.pden0 = part + 4 * ps;
        a.npart = col_grid_mk;
        if constexpr (Backend::kCanCouple) {
            if (cpl_comm) {                            // the pairs of a coupled batch live on several ranks: all-rank sums / maxima
                if (be.couple(cpl_comm, part, ps, col_grid_mk, cpl_work)) return;
                a.pnum0 = cpl_work;
                a.pden0 = cpl_work + 1;
                a.pnum = cpl_work + 2;
                a.pden = cpl_work + 3;
                a.pmax = cpl_work + 4;
                a.npart = 1;
            }
        }
        be.launch_row(a, row_grid, row_block, row_lds, units);
        ++seq;
    }
    void launch_mk_col(const MkConst &k, int mode, int sg = SG_ALL) {
        ColArgs<T> a = col_args(kPacked ? 1 : 2, mode);
        a.sg = sg;
        a.cin = ctrl + (size_t)(seq & 1) * units;
        a.cout = ctrl + (size_t)((seq + 1) & 1) * units;
        a.k = k;
        a.npart = col_grid_mk;
        be.launch_col(a, col_grid_mk, col_block_mk, col_lds_mk, units);
        ++seq;
    }
    int prepare_trace(ssf_trace *trace, int maxIter) {
        const long long cap = trace ? trace->capacity : 0;
        if (cap > tr_cap || maxIter != tr_maxIter) {
            for (void *q : {(void *)tr_hz, (void *)tr_lim, (void *)tr_it})
                if (q) be.free(q);
            tr_hz = tr_lim = nullptr;
            tr_it = nullptr;
            tr_cap = 0;
            if (cap > 0) {
                tr_hz = (double *)be.alloc(sizeof(double) * (size_t)cap);
                tr_it = (int *)be.alloc(sizeof(int) * (size_t)cap);
                tr_lim = (double *)be.alloc(sizeof(double) * (size_t)cap * (size_t)maxIter);
                if (!tr_hz || !tr_it || !tr_lim) return oom();
                tr_cap = cap;
                tr_maxIter = maxIter;
            }
        }
        if (tr_cap > 0) be.memset(tr_lim, 0xFF, sizeof(double) * (size_t)tr_cap * (size_t)maxIter);   // NaN pattern
        return SSF_OK;
    }

    // One span on the device: the field is in T[cur] on entry and in T[cur] (updated) on exit.  The host enqueues
    // Col, [Row, Col]* in chunks and reads the control block between chunks (a synchronising 300-byte read).
    struct SpanRun {
        long long trace_n = 0;
        double avg_it = 3.0;
        int n_it = 0;            // iterations of the latest finished step (0: none yet)
    };
    int run_span(const ssf_params &p, const MkConst &k, SpanRun &sr, ssf_stats *st) {
        std::vector<Ctrl> cs((size_t)units);
        for (auto &c : cs) {
            c = Ctrl{};
            c.state = ST_NEED_S;
            c.cur = cur;
            c.trace_n = sr.trace_n;
        }
        const size_t cbytes = sizeof(Ctrl) * (size_t)units;
        be.h2d(ctrl + (size_t)(seq & 1) * units, cs.data(), cbytes);
        bool persisted = false;
        if constexpr (Backend::kCanPersist && !kPacked) {
            // experiment (off by default): the whole span as ONE persistent launch (engine_fused_impl.h: k_mk_span)
            const int workers = be.persist_mk_workers();
            if (workers > 0 && units == 1 && !N2mix && row_block == 256 && col_block_mk == 256 && row_v == 16 && col_v == 16 &&
                std::is_same<T, double>::value) {
                SpanMkArgs<T> a{};
                a.row = row_args();
                a.row.use_ctrl = 1;
                a.row.k = k;
                a.row.pmax = part;
                a.row.pnum = part + npart_max;
                a.row.pden = part + 2 * (size_t)npart_max;
                a.row.pnum0 = part + 3 * (size_t)npart_max;
                a.row.pden0 = part + 4 * (size_t)npart_max;
                a.row.npart = col_grid_mk;
                a.col = col_args(2, CM_MK);
                a.row.tw_off = 0;
                a.col.k = k;
                a.col.npart = col_grid_mk;
                a.ctrl = ctrl;
                a.seq0 = seq;
                a.row_grid = row_grid;
                a.col_grid = col_grid_mk;
                a.max_stages = 1 << 20;
                a.nworkers = std::min(workers, std::max(row_grid, col_grid_mk));
                a.xcd = be.persist_mk_xcd() ? 0 : -1;
                a.ctrl_lds = (std::max(row_lds, col_lds_mk) + 15) / 16 * 16;
                a.bar = gbar;
                const int grid = a.xcd >= 0 ? 8 * a.nworkers : a.nworkers;
                if (be.launch_mk_span(a, grid, a.ctrl_lds + sizeof(Ctrl) + 64)) return hiperr();
                unsigned flags[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                be.d2h(flags, gbar, sizeof(flags));
                if (!be.ok()) return hiperr();
                if (flags[2]) {
                    err = "persistent Manakov kernel: grid barrier timed out (workgroups not co-resident / not on one XCD?)";
                    return SSF_ERR_STATE;
                }
                seq = (seq & ~1u) | (flags[4] & 1u);                                       // parity of the block the span ended in
                be.d2h(cs.data(), ctrl + (seq & 1), sizeof(Ctrl));
                if (!be.ok()) return hiperr();
                if (cs[0].state != ST_SPAN_DONE || cs[0].pend0) {
                    err = "persistent Manakov kernel: span not finished";
                    return SSF_ERR_STATE;
                }
                persisted = true;
            }
        }
        if (!persisted) {
        launch_mk_col(k, CM_MK);                                                       // first step start
        int guard = 0;
        long long prev_steps = 0, prev_iters = 0, prev_rare = 0;
        // Stage-specialised column kernels (col_split; fused_kernels.h: stage_group): the host enqueues them along the sequence it
        // predicts -- per step H, ADV x (iterations - 1), FIN, with the iteration count of the latest finished step
        // (Ctrl::last_nit) -- and a kernel whose stage the state does not ask for does nothing, so a wrong guess (the 3 -> 2
        // crossover of a lossy span: once per span) costs idle launches until the pattern meets the state again -- two per step
        // when the guess is one too high, four when it is one too low, for the rest of one chunk -- never a wrong result.
        // The rare stages (rebuilds of iterate 0, recoveries of the step-start field) ride in the H kernel: they wait for the
        // next H slot (at most one predicted step).  A chunk without progress is followed by a short general chunk; where the
        // rare stages are the rule (weak nonlinearity: lim_0 < tol at every step) the call goes back to the general kernel for
        // good (and so do the next kSplitPenalty calls of the plan).
        // (a coupled batch across ranks: every row launch carries an all-gather, so all ranks must enqueue the SAME launches.  The
        //  pattern, the prediction and the chunk size depend on per-rank state -- column geometry by pairs per rank, plan history --
        //  so a coupled span runs the general kernel, in chunks derived from the rank-identical control block only: run_manakov)
        const bool can_split = col_split && split_penalty == 0 && units == 1 && !cpl_comm &&
                               be.can_split_cols(col_args(kPacked ? 1 : 2, CM_MK), col_block_mk);
        bool general_chunk = false;
        int stalls = 0, pos = 0;                 // pos: position in the predicted pattern (kept from chunk to chunk)
        long long idle_pairs = 0;                // pairs enqueued since a step last finished
        for (;;) {
            // [Row, Col] pairs still needed for this span: (1 + nIter) per step.  A surplus pair is a no-op launch
            // (~10 us); a chunk that ends short of the span costs a synchronising read and an idle stream (~100-150 us),
            // so the estimate is rounded up, not down: fixed step = remaining steps x (1 + recent iterations per step)
            // + a few pairs (the 20-step driver run needed three rounds with the old 0.95 x estimate: -10 % steps/s).
            // Several units: the one with the most steps left decides (the others idle through their surplus launches).
            double steps_rem = 0.0;
            for (const Ctrl &c : cs) {
                double r;
                if (c.state == ST_SPAN_DONE && !c.pend0) r = 0.0;
                else if (c.steps == 0 && c.state == ST_NEED_S) r = p.nlprMethod ? 8.0 : std::ceil(p.Lspan / p.hz);
                else r = std::max(1.0, std::ceil((p.Lspan - c.z) / (c.hz > 0 ? c.hz : p.hz)));
                steps_rem = std::max(steps_rem, r);
            }
            const bool use_split = can_split && split_ok && !general_chunk;
            const int n_pred = std::max(1, std::min(p.maxIter, sr.n_it > 0 ? sr.n_it : (int)std::lround(sr.avg_it)));
            double est = p.nlprMethod ? steps_rem * (1.0 + sr.avg_it) * 0.6
                                      : steps_rem * (1.0 + (use_split ? std::max((double)n_pred, sr.avg_it) : sr.avg_it)) + 3.0;
            int chunk = (int)std::min(512.0, std::max(2.0, std::ceil(est)));
            if (can_split && split_ok && general_chunk) chunk = std::min(chunk, 8);
            else if (use_split && sr.n_it == 0) chunk = std::min(chunk, 4 * (n_pred + 1));     // nothing known yet: a short look first
            for (int i = 0; i < chunk; ++i) {
                launch_mk_row(k);
                int sg = SG_ALL;
                if (use_split) {
                    sg = pos == 0 ? (SG_H | SG_RARE) : pos < n_pred ? SG_ADV : SG_FIN;
                    pos = pos >= n_pred ? 0 : pos + 1;
                }
                launch_mk_col(k, CM_MK, sg);
            }
            be.d2h(cs.data(), ctrl + (size_t)(seq & 1) * units, cbytes);              // synchronising read
            if (!be.ok()) return hiperr();
            long long steps = 0, iters = 0, rare = 0;
            bool done = true;
            double worst = 0.0;
            for (const Ctrl &c : cs) {
                steps += c.steps;
                iters += c.iterations;
                rare += c.n_rebuilt + c.n_recovered;
                done = done && c.state == ST_SPAN_DONE && !c.pend0;
                if (c.steps > 0) worst = std::max(worst, (double)c.iterations / (double)c.steps);
            }
            if (steps > prev_steps) {                                                 // iterations per step of the last chunk
                sr.avg_it = units == 1 ? (double)(iters - prev_iters) / (double)(steps - prev_steps) : worst;
                sr.n_it = cs[0].last_nit;
            }
            if (can_split && split_ok && !done) {
                // no step finished while the pattern went round twice: the pattern and the state do not meet (e.g. a run that
                // converges at iterate 0 as a rule needs ADV then a rebuild where one iteration per step predicts H, FIN)
                idle_pairs = steps == prev_steps ? idle_pairs + chunk : 0;
                const bool stuck = idle_pairs >= 2 * (n_pred + 1) + 2;
                if (general_chunk && !stuck) general_chunk = false;                   // the general kernel got it going: back to the pattern
                else if (stuck) {
                    general_chunk = true;
                    if (++stalls >= 2) split_ok = false;
                }
                // rebuilds / recoveries wait for the next H slot: where they are the rule the general kernel is the faster one
                if (rare - prev_rare > (steps - prev_steps) / 4 + 2) split_ok = false;
                if (!split_ok) split_penalty = kSplitPenalty;
            }
            prev_steps = steps;
            prev_iters = iters;
            prev_rare = rare;
            if (done) break;
            if (++guard > (1 << 22)) {
                err = "fused engine: span did not terminate";
                return SSF_ERR_STATE;
            }
        }
        }   // (!persisted)
        if (units == 1) {
            cur = cs[0].cur;
        } else {                                   // units may have taken different numbers of steps (adaptive step): bring
            const size_t ub = sizeof(C) * (size_t)rows_u() * (size_t)N;                // every field back into T[cur]
            for (int u = 0; u < units; ++u)
                if (cs[(size_t)u].cur != cur)
                    be.d2d((char *)(cur ? T1 : T0) + (size_t)u * ub, (char *)(cur ? T0 : T1) + (size_t)u * ub, ub);
        }
        sr.trace_n = cs[0].trace_n;
        if ((int)ustats.size() != units) ustats.assign((size_t)units, ssf_stats{});
        for (int u = 0; u < units; ++u) {
            const Ctrl &c = cs[(size_t)u];
            ssf_stats &us = ustats[(size_t)u];
            us.steps += c.steps;
            us.iterations += c.iterations;
            us.nonconverged_steps += c.nonconv;
            us.decided_ahead += c.n_ahead;
            us.rebuilt_iterates += c.n_rebuilt;
            us.recovered_fields += c.n_recovered;
            us.transforms += (int64_t)(kPacked ? 2 * rows_u() : rows_u()) * (2 * c.steps + 2 * c.iterations);
        }
        for (const Ctrl &c : cs) {
            st->steps += c.steps;
            st->iterations += c.iterations;
            st->nonconverged_steps += c.nonconv;
            st->decided_ahead += c.n_ahead;
            st->rebuilt_iterates += c.n_rebuilt;
            st->recovered_fields += c.n_recovered;
            st->transforms += (int64_t)(kPacked ? 2 * rows_u() : rows_u()) * (2 * c.steps + 2 * c.iterations);
        }
        return SSF_OK;
    }
    int fetch_trace(ssf_trace *trace, long long trace_n, int maxIter) {
        if (trace) {
            trace->count = trace_n;
            const long long n = std::min(trace_n, (long long)trace->capacity);
            if (n > 0) {
                if (trace->hz) be.d2h(trace->hz, tr_hz, sizeof(double) * (size_t)n);
                if (trace->iters) be.d2h(trace->iters, tr_it, sizeof(int) * (size_t)n);
                if (trace->lims) be.d2h(trace->lims, tr_lim, sizeof(double) * (size_t)n * (size_t)maxIter);
            }
        }
        return be.ok() ? SSF_OK : hiperr();
    }
    MkConst mk_const_for(const ssf_params &p, const Derived &d, ssf_trace *trace) const {
        MkConst k = mk_const(p, d);
        if (!trace || units > 1) k.trace_cap = 0;                     // (no per-unit traces)
        if (k.trace_cap > 0) k.exact_lim0 = 1;
        return k;
    }

    int run_manakov(const ssf_params &p, const Derived &d, int s0, int s1, const void *noise, ssf_stats *st,
                    ssf_trace *trace) {
        if constexpr (kPacked) return SSF_ERR_UNSUPPORTED;
        int rc = mk_buffers();
        if (rc) return rc;
        if ((rc = prepare_trace(trace, p.maxIter))) return rc;
        const MkConst k = mk_const_for(p, d, trace);
        SpanRun sr;
        if (!cpl_comm) {                                              // (coupled ranks: no per-plan history in the launch count)
            sr.n_it = hint_n_it;                                      // (the previous call's iteration count: a guess, see run_span)
            if (hint_n_it > 0) sr.avg_it = (double)hint_n_it;
        }
        for (int span = s0; span <= s1; ++span) {
            if (p.direction < 0 && (p.amp == SSF_AMP_EDFA || p.amp == SSF_AMP_IDEAL))      // equalization.py:1090-1092
                launch_amp(Tcur(), (S)std::exp(-d.alpha_lin / 2 * p.Lspan), nullptr);
            if ((rc = run_span(p, k, sr, st))) return rc;
            if (p.direction >= 0 && (rc = amp_fwd(p, d, span, span - s0, noise, std::exp(d.alpha_lin / 2 * p.Lspan)))) return rc;
            if (wants_snapshot(p, span) && (rc = snapshot())) return rc;
        }
        be.sync();
        if (!be.ok()) return hiperr();
        hint_n_it = sr.n_it;
        return fetch_trace(trace, sr.trace_n, p.maxIter);
    }

    // complex64 Manakov on the packed-pair pipeline: per span the rows of T[cur] are packed into the pair core's field
    // (one pass), propagated there, and unpacked again for the amplifier / snapshot stage (one pass): two extra passes
    // over the field per SPAN against (2 + 2 nIter) per STEP.
    bool packed_ok() const {
        Split t;
        return std::is_same<T, float>::value && use_packed && !N2mix && (nrows % 2) == 0 && choose_split(log2N, SSF_C128, &t, true);
    }
    int run_manakov_packed(const ssf_params &p, const Derived &d, int s0, int s1, const void *noise, ssf_stats *st,
                           ssf_trace *trace) {
        if constexpr (!std::is_same<T, float>::value) return SSF_ERR_UNSUPPORTED;
        else {
            int rc;
            if (!pk) {
                pk.reset(new FusedCore<pf2, Backend>(be, N, nrows / 2, SSF_C128, (void *)G, units));
                pk->lanes_hint = lanes_hint;
                if ((rc = pk->init())) {
                    err = pk->err;
                    pk.reset();
                    return rc;
                }
                // the packed core is created on first use: a coupling communicator attached before that (models._manakov attaches,
                // then executes) must reach it too, or the first complex64 coupled call would reduce over this rank's pairs only
                if (cpl_comm && (rc = pk->set_couple(cpl_comm, cpl_nranks))) {
                    err = pk->err;
                    pk.reset();
                    return rc;
                }
            }
            if ((rc = pk->prepare_trace(trace, p.maxIter))) return rc;
            const MkConst k = pk->mk_const_for(p, d, trace);
            typename FusedCore<pf2, Backend>::SpanRun sr;
            if (!pk->cpl_comm) {
                sr.n_it = pk->hint_n_it;
                if (pk->hint_n_it > 0) sr.avg_it = (double)pk->hint_n_it;
            }
            for (int span = s0; span <= s1; ++span) {
                if (p.direction < 0 && (p.amp == SSF_AMP_EDFA || p.amp == SSF_AMP_IDEAL))
                    launch_amp(Tcur(), (S)std::exp(-d.alpha_lin / 2 * p.Lspan), nullptr);
                pk->cur = 0;
                be.launch_repack(RepackArgs{Tcur(), pk->T0, (long long)N, nrows / 2, 1}, 1024, 256);
                if ((rc = pk->run_span(p, k, sr, st))) {
                    err = pk->err;
                    return rc;
                }
                be.launch_repack(RepackArgs{Tcur(), pk->Tcur(), (long long)N, nrows / 2, 0}, 1024, 256);
                if (p.direction >= 0 && (rc = amp_fwd(p, d, span, span - s0, noise, std::exp(d.alpha_lin / 2 * p.Lspan)))) return rc;
                if (wants_snapshot(p, span) && (rc = snapshot())) return rc;
            }
            be.sync();
            if (!be.ok()) return hiperr();
            pk->hint_n_it = sr.n_it;
            if ((rc = pk->fetch_trace(trace, sr.trace_n, p.maxIter))) err = pk->err;
            return rc;
        }
    }

    int execute(const ssf_params &p, int s0, int s1, const void *noise, ssf_stats *st, ssf_trace *trace) {
        const Derived d = derive(p);
        if (units > 1 && (trace || (p.model != SSF_MODEL_NLSE && (rows_u() % 2)))) {
            err = "independent units: an even number of rows per unit for the Manakov models, no trace";
            return SSF_ERR_BAD_ARG;
        }
        be.time_begin();
        split_ok = true;
        if (split_penalty > 0) --split_penalty;
        if (pk) {
            pk->split_ok = true;
            if (pk->split_penalty > 0) --pk->split_penalty;
        }
        int rc = p.model == SSF_MODEL_NLSE ? run_nlse(p, d, s0, s1, noise, st)
                 : packed_ok()             ? run_manakov_packed(p, d, s0, s1, noise, st, trace)
                                           : run_manakov(p, d, s0, s1, noise, st, trace);
        if (rc) return rc;
        st->device_ms += be.time_end();
        st->n_snapshots = (int32_t)snaps.size() + n_sunk;
        return SSF_OK;
    }

    // Eo = ifft(fft(Ei) * exp(-alpha/2 L + j beta2/2 w^2 L))     (channels.py:97)
    int linear_channel(double Fs, double Fc, double alpha, double D, double L) {
        if constexpr (kPacked) return SSF_ERR_UNSUPPORTED;
        ssf_params p{};
        p.Fs = Fs; p.Fc = Fc; p.alpha = alpha; p.D = D; p.direction = 1; p.Lspan = 1; p.NF = 4.5;
        const Derived d = derive(p);
        const double w2 = (d.w_scale / (double)N) * (d.w_scale / (double)N);
        LinOp lo = make_linop(L, d.lin_a, d.lin_b, w2, 1.0 / (double)N, log2N);
        be.h2d(linops, &lo, sizeof(lo));
        launch_col_plain(CM_PLAIN_FWD, Tcur(), (S)0);
        launch_row_lin(linops);
        launch_col_plain(CM_PLAIN_INV, Tcur(), (S)0);
        be.sync();
        return be.ok() ? SSF_OK : hiperr();
    }
};

}  // namespace fused
}  // namespace ssf
