// gte_aux.hip — auxiliary kernels of libgte, deliberately in their OWN translation unit:
// adding a kernel to gte_kernels.hip perturbs the register allocation of the step kernel
// compiled next to it (measured: 78 -> 83 VGPRs, occupancy 6 -> 5 waves/SIMD, +4 us per step).
#include "gte_device.h"

namespace gte {

// ---------------------------------------------------------------------------
// Trajectory log (optional, gte_config.log_steps): one row per env after every reset /
// step — what History.add records (reference environments.py:253-264).
struct LogArrays {
  int32_t *idx, *step, *pos, *dsi;
  double *pv, *realpos, *reward;
  uint8_t* flags;
};

__global__ void gte_log_kernel(const EnvRec* rec, const double* reward64, const uint8_t* term,
                               const uint8_t* trunc, int n, int64_t row_base, LogArrays o) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const EnvRec r = rec[e];
  const int64_t k = row_base + e;
  o.idx[k] = r.idx; o.step[k] = r.step; o.pos[k] = r.pos; o.dsi[k] = r.dsi;
  o.pv[k] = r.pv; o.realpos[k] = r.realpos; o.reward[k] = reward64[e];
  o.flags[k] = (uint8_t)((term[e] ? 1 : 0) | (trunc[e] ? 2 : 0));
}

hipError_t launch_log(const EnvRec* rec, const double* reward64, const uint8_t* term,
                      const uint8_t* trunc, int n, int64_t row_base, const LogArrays& o,
                      hipStream_t stream) {
  hipLaunchKernelGGL(gte_log_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rec, reward64, term,
                     trunc, n, row_base, o);
  return hipGetLastError();
}

}  // namespace gte
