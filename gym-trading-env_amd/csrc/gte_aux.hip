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

// ---------------------------------------------------------------------------
// State, returns and observation of a range of envs, packed into (pinned, device-visible)
// host memory: `count` gte_env_snapshot structs, then `count` observations.
struct SnapshotPacked {
  int32_t idx, step, pos, dsi, start, episode, needs_reset, terminated, truncated, reserved;
  double asset, fiat, ia, ifi, pv, realpos, reward;
};

__global__ void gte_snapshot_kernel(const EnvRec* rec, const double* reward64, const uint8_t* term,
                                    const uint8_t* trunc, const float* obs, int64_t obs_elems,
                                    int first, SnapshotPacked* dst, float* dst_obs) {
  const int e = first + blockIdx.x;  // one workgroup per env
  if (threadIdx.x == 0) {
    const EnvRec r = rec[e];
    SnapshotPacked s;
    s.idx = r.idx; s.step = r.step; s.pos = r.pos; s.dsi = r.dsi; s.start = r.start;
    s.episode = r.episode; s.needs_reset = r.needs_reset;
    s.terminated = term[e]; s.truncated = trunc[e]; s.reserved = 0;
    s.asset = r.asset; s.fiat = r.fiat; s.ia = r.ia; s.ifi = r.ifi; s.pv = r.pv;
    s.realpos = r.realpos; s.reward = reward64[e];
    dst[blockIdx.x] = s;
  }
  if (dst_obs)
    for (int64_t i = threadIdx.x; i < obs_elems; i += blockDim.x)
      dst_obs[(int64_t)blockIdx.x * obs_elems + i] = obs[(int64_t)e * obs_elems + i];
}

hipError_t launch_snapshot(const EnvRec* rec, const double* reward64, const uint8_t* term,
                           const uint8_t* trunc, const float* obs, int64_t obs_elems, int first,
                           int count, void* dst, float* dst_obs, hipStream_t stream) {
  hipLaunchKernelGGL(gte_snapshot_kernel, dim3(count), dim3(256), 0, stream, rec, reward64, term,
                     trunc, obs, obs_elems, first, (SnapshotPacked*)dst, dst_obs);
  return hipGetLastError();
}

}  // namespace gte
