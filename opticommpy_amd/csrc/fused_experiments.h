// fused_experiments.h -- kernels that were built, measured and rejected, kept out of the product build (-DSSF_EXPERIMENTS=1,
// `make exp` -> libssf_hip_exp.so): the persistent span kernels of north_star's "persistent HIP pipeline" -- a whole span of
// the scalar-NLSE / Manakov launch sequence inside ONE launch with grid barriers in between.  Same stage bodies, same results,
// slower at every size (profiles/r2_* ssfm, profiles/r3_persistent_manakov.txt; DESIGN.md appendix "measured and rejected").
// Included by engine_fused_impl.h inside namespace ssf::{anonymous}.
#pragma once

// ---- persistent span kernel (scalar NLSE): every stage of a span in ONE launch ----------------------------------------
// For small N a launch is one latency chain (dispatch -> loads -> transform -> stores -> end-of-kernel write-back),
// ~10 us whatever the size, and a span is 2 * nsteps + 1 of them.  Here the stages run inside one launch of <= 256
// co-resident workgroups (one per CU at most) with a grid barrier in between: arrival counter + generation word,
// agent-scope release before / acquire after (the L2s of the eight XCDs are not coherent with each other, so the stage's
// output is written back and the readers' lines invalidated), bounded spin (a barrier that cannot complete sets the abort
// word and the host reports it instead of hanging).  The stage bodies are the ones of the per-stage kernels, called with
// virtual workgroup numbers.  Reference loop: optic/models/channels.py:215-232.
__device__ __forceinline__ bool grid_sync(unsigned *bar, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                          // release: this workgroup's stores, device-wide
        const unsigned g = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24) || __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __threadfence();
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // every wave: no stale L1 / L2 lines of the previous stage
    return __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}
template <typename T, int LGR, int LGC> __global__ void __launch_bounds__(256) k_nlse_span(const SpanNlseArgs<T> a) {
    SSF_DEV_CTX(0);
    const int me = (int)blockIdx.x, nwg = (int)gridDim.x;
    auto col_stage = [&](int mode) {
        for (int vb = me; vb < a.col_grid; vb += nwg) {
            ctx.bid = vb;
            if (mode == CM_NLSE_FIRST) col_body<T, LGC, CM_NLSE_FIRST, false>(ctx, a.col);
            else if (mode == CM_NLSE_STEP) col_body<T, LGC, CM_NLSE_STEP, false>(ctx, a.col);
            else col_body<T, LGC, CM_NLSE_LAST, false>(ctx, a.col);
            __syncthreads();
        }
    };
    auto row_stage = [&](const LinOp *lin) {
        RowArgs<T> ra = a.row;
        ra.lin = lin;
        for (int vb = me; vb < a.row_grid; vb += nwg) {
            ctx.bid = vb;
            row_body<T, LGR>(ctx, ra);
            __syncthreads();
        }
    };
    col_stage(CM_NLSE_FIRST);                                     // channels.py:216
    if (!grid_sync(a.bar, nwg)) return;
    row_stage(a.lin_half);
    if (!grid_sync(a.bar, nwg)) return;
    for (int s = 1; s < a.nsteps; ++s) {
        col_stage(CM_NLSE_STEP);
        if (!grid_sync(a.bar, nwg)) return;
        row_stage(a.lin_full);                                    // lin * lin: second half of one step, first half of the next
        if (!grid_sync(a.bar, nwg)) return;
    }
    col_stage(CM_NLSE_STEP);
    if (!grid_sync(a.bar, nwg)) return;
    row_stage(a.lin_half);
    if (!grid_sync(a.bar, nwg)) return;
    col_stage(CM_NLSE_LAST);                                      // channels.py:232
}

// ---- persistent span kernel (Manakov): Col, [Row, Col]* of a span in ONE launch ------------------------------------------------------
// north_star's "persistent HIP pipeline" for the Manakov path.  Same stage bodies, virtual workgroup numbers, the control block of
// section 3.3 read through an LDS copy (fetched with L1-bypassing loads after every barrier: the scalar cache is not covered
// by an acquire).  Two barriers: the agent-scope one of the ssfm span kernel (grid_sync: L2 write-back + invalidate), and an
// XCD-confined one -- only the workgroups that happen to run on one XCD take part (ticket counter; the others exit at once), their
// stores meet in that XCD's L2, so a barrier is: drain the stores, arrive / spin, invalidate the L1.  OFF by default
// (SSF_PERSIST_MK=<workers>, SSF_PERSIST_XCD=1): measured against the launch sequence in profiles/r3_persistent_manakov.txt.
__device__ __forceinline__ bool grid_sync_xcd(unsigned *bar, unsigned nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores are in the XCD's L2 (the L1 writes through)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22) || __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(bar + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // no stale L1 lines of what the other CUs of this XCD wrote
    return __hip_atomic_load(bar + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}
template <typename T, int LGR, int LGC> __global__ void __launch_bounds__(256) k_mk_span(const SpanMkArgs<T> a) {
    SSF_DEV_CTX(0);
    int *s_me = (int *)(ssf_smem + a.ctrl_lds + sizeof(Ctrl));          // (no static LDS: the dynamic part may take all 160 KiB)
    int me = (int)blockIdx.x, nwg = (int)gridDim.x;
    if (a.xcd >= 0) {
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 0xf;
            *s_me = (int)xcc == a.xcd ? (int)__hip_atomic_fetch_add(a.bar + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        }
        __syncthreads();
        me = *s_me;
        nwg = a.nworkers;
        if (me < 0 || me >= nwg) return;
    }
    Ctrl *lc = (Ctrl *)(ssf_smem + a.ctrl_lds);
    unsigned seq = a.seq0;
    for (int stage = 0; stage < a.max_stages; ++stage) {
        const Ctrl *gin = a.ctrl + (seq & 1);
        Ctrl *gout = a.ctrl + ((seq + 1) & 1);
        for (int i = (int)threadIdx.x; i < (int)(sizeof(Ctrl) / 8); i += (int)blockDim.x)
            ((unsigned long long *)lc)[i] = __hip_atomic_load((const unsigned long long *)gin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (lc->state == ST_SPAN_DONE && !lc->pend0) {
            if (me == 0 && threadIdx.x == 0) a.bar[4] = seq & 1u;           // which block holds the final state
            break;
        }
        if ((stage & 1) == 0) {
            ColArgs<T> ca = a.col;
            ca.cin = lc;
            ca.cout = gout;
            ctx.nblocks = a.col_grid;                                     // (the stage's own grid: Owned slots are keyed by it)
            for (int vb = me; vb < a.col_grid; vb += nwg) {
                ctx.bid = vb;
                col_body<T, LGC, CM_MK, false>(ctx, ca);
                __syncthreads();
            }
        } else {
            RowArgs<T> ra = a.row;
            ra.cin = lc;
            ra.cout = gout;
            ctx.nblocks = a.row_grid;
            for (int vb = me; vb < a.row_grid; vb += nwg) {
                ctx.bid = vb;
                row_body<T, LGR>(ctx, ra);
                __syncthreads();
            }
        }
        ++seq;
        const bool ok = a.xcd >= 0 ? grid_sync_xcd(a.bar, (unsigned)nwg) : grid_sync(a.bar, (unsigned)nwg);
        if (!ok) return;
    }
}

