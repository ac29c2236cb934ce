// velo_lm_ag.hip -- the all-gather Levenberg-Marquardt solve (velo_lm_ag_kernels.h) as a translation unit of its own, compiled with
// -mllvm -disable-machine-licm (see that header).  It exports ONE host function, the launcher; the shared kernel header is included
// inside an unnamed namespace, so every kernel it defines has internal linkage here and nothing of it collides with velo_hip.hip's copies.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/velo_hip.h"

#define VELO_UNIT_LM_ONLY 1
namespace {
#include "velo_device_math.h"
#include "velo_kernels.h"
#include "velo_lm_ag_kernels.h"
}  // namespace

// The argument blocks are velo_hip.hip's LMParams / LMBatchPackV / AgCtl (the same header, hence the same layout), handed over as bytes.
// a, b: the event pair of a timed launch (or null).  -> hipError_t of the launch
extern "C" __attribute__((visibility("hidden")))
int velo_launch_lm_solve_ag(int nb_max, int n, void* stream, const void* lm_params, size_t lm_params_bytes, const void* pack, size_t pack_bytes,
                            void* ctl, int kmax, size_t half, void* a, void* b) {
    if (lm_params_bytes != sizeof(velo::LMParams) || pack_bytes != sizeof(velo::LMBatchPackV)) return (int)hipErrorInvalidValue;
    velo::LMParams Q;
    velo::LMBatchPackV P;
    __builtin_memcpy(&Q, lm_params, sizeof(Q));
    __builtin_memcpy(&P, pack, sizeof(P));
    hipExtLaunchKernelGGL(velo::lm_solve_ag_batch_kernel, dim3(nb_max, n), dim3(velo::kEvalThreads), 0, (hipStream_t)stream, (hipEvent_t)a, (hipEvent_t)b, 0,
                          Q, P, (velo::AgCtl*)ctl, kmax, half);
    return (int)hipGetLastError();
}
