// velo_lm_ag_kernels.h -- the all-gather form of a whole Levenberg-Marquardt solve in ONE launch (reference: ceres::Solve, velo.h:897-902).
// Included by velo_lm_ag.hip only, behind velo_kernels.h (same namespace): that translation unit is compiled with
// -mllvm -disable-machine-licm, because hoisting the ~50 constant materialisations of the transition (double-precision division / square
// root / sine-cosine sequences) out of the solve loop pins them in registers across the sweep -- 238 VGPRs, one wave per SIMD -- where the
// loop fits 167 without them (no scratch).  The flag is per translation unit, and the association kernels must keep their hoisting.
#pragma once
#include "velo_kernels.h"

namespace velo {

// ---- a whole solve in ONE launch, all-gather form (round 5) ------------------------------------------------------------------------
// The launch-per-iteration kernels pay, per LM iteration, a kernel boundary, the cold loads of eval point and rows, a ticket, the
// stepping workgroup's re-read of every partial row, the state's way back through memory -- ~20 us alone and 25-30 us beside other
// groups' association kernels for ~5 us of arithmetic -- and a chained call has to PREDICT how many launches a solve will need.
// Here the workgroups of a context stay for the whole solve: workgroup bx owns virtual block bx in every iteration (grid = the plan's
// blocks: the host sizes the plan so that every lock-step group's workgroups are resident together, eval_plan), publishes its partial row
// with write-through stores and a tagged flag, waits for the flags of ALL blocks of its context and then runs the transition ITSELF --
// the one-launch iteration's redundant lm_advance (lm_iter_lean_body): identical inputs, identical arithmetic, identical state and eval
// point in every workgroup's LDS.  ONE hand-off per iteration (an all-gather of nb x 224 bytes) instead of two (fan-in to a stepping
// workgroup, broadcast of the eval point), no kernel boundary, no launch-count prediction: a solve is one launch however many
// iterations it takes.  The rows of the next sweep are requested BEFORE the wait.  Virtual blocks, per-thread rows, reductions and
// lm_advance are those of every other path, so states, costs and counts are bit-identical to them.
// Hand-off: partial rows double-buffered by the parity of the sweep (a workgroup can be at most one sweep ahead of the slowest: to
// publish sweep k + 1 it must have seen every block's sweep k); flag[parity][bx] = epoch + k + 1 once row bx of sweep k is written
// through (row stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> one flag store, all agent scope, no fence).  The epoch is read at
// the start and moved past every tag of this launch by workgroup 0 at the end (it cannot end before every workgroup has read it: it
// needs their rows) -- nothing is ever reset, comparisons are wrap-safe differences.
// Every wait is bounded (kAgTimeoutTicks of the 100 MHz clock): a workgroup that waits longer -- its peers never became resident, e.g.
// beside other processes' kernels -- raises the context's abort word, everybody leaves, the state stays "not done" and the pose record
// "not ready": the chained call fails like a call whose launch prediction fell short and is repeated host-driven (same results).
#ifndef VELO_AG_KATTR
#define VELO_AG_KATTR
#endif
#ifndef VELO_AG_PRE
#define VELO_AG_PRE VELO_LEAN_PRE
#endif
#ifndef VELO_AG_WAVES
#define VELO_AG_WAVES 3
#endif
#ifndef VELO_AG_NOINLINE
#define VELO_AG_NOINLINE __forceinline__
#endif
constexpr unsigned long long kAgTimeoutTicks = 2000000ull;                 // 20 ms
__device__ __forceinline__ bool ag_wait(AgCtl* __restrict__ ctl, const int parity, const int nb, const int tag, const int abort_tag, const int tid) {
    bool ok = true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < nb; i += kEvalThreads) {
        while ((int)(ctl_load(&ctl->flag[parity][i]) - tag) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kAgTimeoutTicks || ctl_load(&ctl->abort) == abort_tag) { ok = false; break; }
        }
        if (!ok) break;
    }
    const int all = __syncthreads_and(ok ? 1 : 0);
    if (!all && tid == 0) ctl_store(&ctl->abort, abort_tag);
    return all != 0;
}
// the transition as a real function: inlined into the loop it drives the kernel to 260 registers (one wave per SIMD: it could not sit beside
// association workgroups)
template <int CHUNK>
__device__ VELO_AG_NOINLINE void ag_advance(const LMParams* Q, const LMState* S, const double* xd, const int* n_valid, PoseRecord* pose_out, SolveLog* log,
                                                    const double* partials, int nb, int first, double* s_scratch, LMState* sL, LMEvalPoint* s_pt, bool writer, int tid) {
    lm_advance<true, CHUNK, false, true>(*Q, S, partials, nb, first ? 2 : 0, xd, n_valid, s_scratch, sL, s_pt, nullptr, 0, nullptr, pose_out, log, writer, first != 0, tid);
}
template <bool M_LDS, int PRE, int CHUNK>
__device__ __forceinline__ void lm_solve_ag_body(const LMParams& Q, const LMBatchItem& it, AgCtl* __restrict__ ctl, const int kmax, const size_t half) {
    __shared__ LMEvalPoint s_pt;
    __shared__ LMState sL;
    __shared__ double s_scratch[CHUNK * kNumAcc];
    const int t = threadIdx.x, bx = blockIdx.x;
    const int nb = it.nb_icp;
    if (bx >= nb) return;
    const int epoch = ctl->epoch;                                         // written by the previous launch on this context
    {
        if (t < 4) {                                                      // the start's eval point, built here (eval_step_batch_body, first)
            double x[6];
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = it.xd ? it.xd[k] : it.S->x[k];
            eval_point_column(x, 0, t, &s_pt);
        }
        __syncthreads();
    }
    bool finished = false;
    RowPrefetchT<PRE> f = prefetch_rows<PRE>(it.A, bx, nb);
#pragma clang loop unroll(disable)
    for (int k = 0; k < kmax; k++) {
        int tt = t;
        asm volatile("" : "+v"(tt));                                      // opaque per iteration: see block_reduce_store
        const EvalArgs& A = it.A;
        double acc[kNumAcc];
        sweep_rows<M_LDS, PRE>(A, f, s_pt, bx, nb, acc, tt);
        block_reduce_store<true>(acc, A.partials + (size_t)(k & 1) * half + (size_t)bx * kNumAcc, s_scratch, tt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this thread's row entries have been written through
        __syncthreads();
        if (tt == 0) ctl_store(&ctl->flag[k & 1][bx], epoch + k + 1);
#ifdef VELO_AG_PREFETCH
        f = prefetch_rows<PRE>(A, bx, nb, tt);                            // the rows do not depend on the pose: the next sweep's first ones are requested ahead of the wait
#endif
        if (!ag_wait(ctl, k & 1, nb, epoch + k + 1, epoch + 1, tt)) break;    // (the abort word carries the launch's own tag: nothing to reset)
        ag_advance<CHUNK>(&Q, it.S, it.xd, it.n_valid, it.pose_out, it.log, A.partials + (size_t)(k & 1) * half, nb, k == 0 ? 1 : 0, s_scratch, &sL, &s_pt, bx == 0, tt);
        if (sL.done) { finished = true; break; }
#ifndef VELO_AG_PREFETCH
        f = prefetch_rows<PRE>(A, bx, nb, tt);
#endif
    }
    if (bx == 0) {
        if (finished && t < (int)(sizeof(LMState) / 8)) reinterpret_cast<unsigned long long*>(it.S)[t] = reinterpret_cast<const unsigned long long*>(&sL)[t];
        if (t == 0) ctl->epoch = epoch + kmax + 4;
    }
}
__global__ void __launch_bounds__(kEvalThreads, VELO_AG_WAVES) VELO_AG_KATTR
lm_solve_ag_batch_kernel(LMParams Q, LMBatchPackV P, AgCtl* __restrict__ ctl, int kmax, size_t half) {
    lm_solve_ag_body<true, VELO_AG_PRE, 64>(Q, P.item[blockIdx.y], ctl + blockIdx.y, kmax, half);
}


}  // namespace velo
