// gte_aux.hip — auxiliary kernels of libgte, deliberately in their OWN translation unit:
// adding a kernel to gte_kernels.hip perturbs the register allocation of the step kernel
// compiled next to it (measured: 78 -> 83 VGPRs, occupancy 6 -> 5 waves/SIMD, +4 us per step).
#include "gte_device.h"

namespace gte {

// ---------------------------------------------------------------------------
// Trajectory log (LogArrays, gte_device.h): the row of a reset, of a step of the kernels that do
// not write it themselves (the isolated hot instantiations, the unfused rollout loop).
__global__ void gte_log_kernel(const EnvRec* rec, const double* reward64, const uint8_t* term,
                               const uint8_t* trunc, int n, int64_t row_base, LogArrays o) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const EnvRec r = rec[e];
  const int64_t k = row_base + e;
  o.idx[k] = r.idx; o.step[k] = r.step; o.pos[k] = r.pos; o.dsi[k] = r.dsi;
  // a row a reset wrote carries reward 0 (environments.py:196) — also when the reset ran inside
  // the step that ended the episode (same-step mode: the terminal step's reward is in the return
  // buffers and in final_info, this row already describes the new episode)
  o.pv[k] = r.pv; o.realpos[k] = r.realpos; o.reward[k] = (r.step == 0) ? 0.0 : reward64[e];
  o.asset[k] = r.asset; o.fiat[k] = r.fiat; o.ia[k] = r.ia; o.ifi[k] = r.ifi;
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
// Dynamic features computed OUTSIDE the step kernel (a user's Python callable evaluated
// vectorised over the batch, reference environments.py:152-154): overwrite feature i (bit i of
// `mask`) of the CURRENT row of every env — in the env's dynamic store, from where later windows
// read it, and in the observation the last launch produced.
__global__ void gte_set_dynamic_kernel(const Params p, const float* values, uint32_t mask) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const int32_t idx = p.rec[e].idx;
  const int64_t slot = p.persist ? (int64_t)idx : (int64_t)(idx % p.W);
  float* ring = p.ring + ((int64_t)e * p.depth + slot) * p.nd;
  float* row = p.obs + ((int64_t)e * p.W + (p.W - 1)) * p.Fobs + p.Fs;
  for (int i = 0; i < p.nd; ++i)
    if (mask & (1u << i)) {
      const float v = values[(int64_t)e * p.nd + i];
      ring[i] = v;
      row[i] = v;
    }
}

hipError_t launch_set_dynamic(const Params& p, const float* values, uint32_t mask, hipStream_t stream) {
  hipLaunchKernelGGL(gte_set_dynamic_kernel, dim3((p.N + 255) / 256), dim3(256), 0, stream, p, values, mask);
  return hipGetLastError();
}

// The same from up to GTE_MAX_DYN separate device columns (f32 or f64 [N] each; NULL = leave the
// feature alone): what a vectorised Python callable returns, without the caller packing and
// converting them first (three small torch launches per feature).
struct DynColumns {
  const void* col[GTE_MAX_DYN];
  int32_t is_f64[GTE_MAX_DYN];
};

__global__ void gte_set_dynamic_columns_kernel(const Params p, const DynColumns c) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const int32_t idx = p.rec[e].idx;
  const int64_t slot = p.persist ? (int64_t)idx : (int64_t)(idx % p.W);
  float* ring = p.ring + ((int64_t)e * p.depth + slot) * p.nd;
  float* row = p.obs + ((int64_t)e * p.W + (p.W - 1)) * p.Fobs + p.Fs;
  for (int i = 0; i < p.nd; ++i)
    if (c.col[i]) {
      // (float)double rounds to nearest even, like the reference's cast into its f32 _obs_array
      const float v = c.is_f64[i] ? (float)((const double*)c.col[i])[e] : ((const float*)c.col[i])[e];
      ring[i] = v;
      row[i] = v;
    }
}

hipError_t launch_set_dynamic_columns(const Params& p, const void* const* cols, const int32_t* is_f64,
                                      hipStream_t stream) {
  DynColumns c;
  for (int i = 0; i < GTE_MAX_DYN; ++i) {
    c.col[i] = i < p.nd ? cols[i] : nullptr;
    c.is_f64[i] = i < p.nd ? is_f64[i] : 0;
  }
  hipLaunchKernelGGL(gte_set_dynamic_columns_kernel, dim3((p.N + 255) / 256), dim3(256), 0, stream, p, c);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// A user's reward_function evaluated outside the kernel, for the whole batch: the reference's
// rules around it (environments.py:265-267: no reward when the step terminated; :196: rows a
// reset wrote carry reward 0) and the three places the value lives — the f64 and f32 return
// buffers and the newest row of the trajectory log (`historical_info["reward", -1] = reward`).
// terminal_view (same-step auto-reset): an env that ended shows its TERMINAL row to the callable,
// so the reset row underneath does not zero the reward it RETURNS; the log row itself — the reset
// row of the next episode — keeps the reference's 0.
__global__ void gte_apply_reward_kernel(const Params p, const double* reward, const int32_t* log_step_row,
                                        double* log_reward_row, int terminal_view) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const bool term = p.terminated[e] != 0;
  const bool ended = term || p.truncated[e] != 0;
  bool reset_row = log_step_row[e] == 0;
  if (terminal_view) reset_row = reset_row && !ended;
  const double r = (term || reset_row) ? 0.0 : reward[e];
  p.reward64[e] = r;
  p.reward[e] = (float)r;
  log_reward_row[e] = (log_step_row[e] == 0) ? 0.0 : r;  // the log row of a reset keeps reward 0 (:196)
}

hipError_t launch_apply_reward(const Params& p, const double* reward, const int32_t* log_step_row,
                               double* log_reward_row, int terminal_view, hipStream_t stream) {
  hipLaunchKernelGGL(gte_apply_reward_kernel, dim3((p.N + 255) / 256), dim3(256), 0, stream, p, reward,
                     log_step_row, log_reward_row, terminal_view);
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
