// velo_hip.hip -- host side of the C-ABI declared in include/velo_hip.h (gfx950 only; no CPU fallback).
//
// One context = one HIP stream + device-resident target index, source cloud, correspondence table and LM
// state.  frame_to_frame (reference velo.h:598-919) runs as
//     for iter:  visual gate kernel;  for icp_iter:  association kernel;  LM solve = chunks of
//     [eval sweep -> (RCCL all-reduce) -> lm_step] launches that early-exit on the device-side `done` flag,
// with one small D2H status copy per chunk (the only host synchronisation inside a solve).
// velo_frame_to_frame_batch advances several contexts in lock-step groups with shared LM launches (f2f_batch_lockstep);
// problems whose sweep is a few workgroups run a whole solve in one launch (lm_solve_small_kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <list>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <chrono>

#include "../../include/velo_hip.h"
#include "velo_kernels.h"
#include "velo_depth_kernels.h"
#include "velo_tri_kernels.h"

using namespace velo;

// velo_lm_ag.hip: the all-gather solve's launcher (a translation unit of its own, see velo_lm_ag_kernels.h)
extern "C" int velo_launch_lm_solve_ag(int nb_max, int n, void* stream, const void* lm_params, size_t lm_params_bytes, const void* pack, size_t pack_bytes,
                                       void* ctl, int kmax, size_t half, void* a, void* b);

// ---- the host side, in parts (round 6: one 4,700-line file before) ---------------------------------------------------------------------
// ONE translation unit: the parts share unnamed-namespace helpers and static functions, and their order is the order of definition.
// The kernels live elsewhere: every family is defined in a velo_unit_*.hip of its own (velo_kernels.h, "translation units").
#include "velo_host_types.inl"   // error state, device buffers, the search index, TargetData and velo_ctx: what one context holds
#include "velo_host_index.inl"   // launch timing, uploads, the index build (build_grid), the query list in patch order
#include "velo_host_assoc.inl"   // pose scalars, parameter blocks, evaluation plan, seeds and askers, the association driver of one context (do_associate)
#include "velo_host_lm.inl"   // peer checks, association timing read-out, the visual gate, the Levenberg-Marquardt driver of one context (do_solve)
#include "velo_host_loaders.inl"   // direction image, target / source ingest and finalize: how a scan enters a context
#include "velo_host_pool.inl"   // resident host threads for the batch entry points (WorkerPool)
#include "velo_api_context.inl"   // C-ABI: create / destroy / params, set_target / set_source / scan cache / promotion, clouds and visual matches
#include "velo_api_solve.inl"   // C-ABI: associate (incl. target-sharded partials and merge), correspondences, evaluate, functors, residual statistics, solve
#include "velo_host_chain.inl"   // chain mode (a whole frame_to_frame call as one chain of launches), kernel times, velo_frame_to_frame
#include "velo_host_batch.inl"   // the lock-step batch driver: group association launches, the next frame loaded behind the chain, f2f_batch_lockstep, jobs, sequences, velo_register_*
#include "velo_api_pose_comm.inl"   // C-ABI: pose helpers and hand-off, the communicators (RCCL, peer slabs), shards, synchronize
#include "velo_api_next_rows.inl"   // C-ABI: SURVEY 8(f) rows 3 and 4 -- projection, keypoint depth, batched triangulation
