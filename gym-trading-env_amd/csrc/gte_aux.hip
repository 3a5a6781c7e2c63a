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
  LogRow w;
  w.idx = r.idx; w.step = r.step; w.pos = r.pos; w.dsi = r.dsi;
  // a row a reset wrote carries reward 0 (environments.py:196) — also when the reset ran inside
  // the step that ended the episode (same-step mode: the terminal step's reward is in the return
  // buffers and in final_info, this row already describes the new episode)
  w.pv = r.pv; w.realpos = r.realpos; w.reward = (r.step == 0) ? 0.0 : reward64[e];
  w.asset = r.asset; w.fiat = r.fiat; w.ia = r.ia; w.ifi = r.ifi;
  w.flags = (uint8_t)((term[e] ? 1 : 0) | (trunc[e] ? 2 : 0));
#pragma unroll
  for (int i = 0; i < 7; ++i) w.pad[i] = 0;
  o.rows[row_base + e] = w;
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
__global__ void gte_apply_reward_kernel(const Params p, const double* reward, LogRow* newest,
                                        int terminal_view) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const bool term = p.terminated[e] != 0;
  const bool ended = term || p.truncated[e] != 0;
  const bool log_reset_row = newest[e].step == 0;
  bool reset_row = log_reset_row;
  if (terminal_view) reset_row = reset_row && !ended;
  const double r = (term || reset_row) ? 0.0 : reward[e];
  p.reward64[e] = r;
  p.reward[e] = (float)r;
  newest[e].reward = log_reset_row ? 0.0 : r;  // the log row of a reset keeps reward 0 (:196)
}

hipError_t launch_apply_reward(const Params& p, const double* reward, LogRow* newest, int terminal_view,
                               hipStream_t stream) {
  hipLaunchKernelGGL(gte_apply_reward_kernel, dim3((p.N + 255) / 256), dim3(256), 0, stream, p, reward,
                     newest, terminal_view);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The logged EPISODE of each of a list of envs, packed into (pinned, device-visible) host memory
// by one launch: what History holds for them (environments.py:253-264), without one strided copy
// per column and env.  One workgroup per listed env.  The episode = the last run of logged rows
// whose `step` counts ..., s-2, s-1, s (the newest row's step s says how far back it can reach;
// the run is cut where the count breaks, e.g. at the repeated rows of a frozen env, and at the
// front when it is longer than the log or than max_rows).  finished (same-step auto-reset with
// final_obs, right after the step in which the env ended): the newest log row already describes
// the next episode's reset row; the episode that just FINISHED is the rows before it plus the
// terminal row from the env's terminal record, with that step's reward.
struct LogPack {
  int32_t* n_rows;                 // [n_ids]
  int32_t *idx, *step, *pos, *dsi; // [n_ids, max_rows], rows 0 .. n_rows-1 valid, oldest first
  double *pv, *realpos, *reward, *asset, *fiat, *ia, *ifi;
  uint8_t* flags;
};

__global__ __launch_bounds__(256) void gte_pack_log_kernel(const LogArrays log, int N, int L,
                                                           long long rows_written, const int32_t* ids,
                                                           int max_rows, int finished,
                                                           const EnvRec* final_rec, const double* reward64,
                                                           LogPack o) {
  __shared__ int s_start;
  const int j = blockIdx.x, tid = threadIdx.x;
  const int e = ids[j];
  const int have = (int)(rows_written < (long long)L ? rows_written : (long long)L);
  if (have <= 0) { if (tid == 0) o.n_rows[j] = 0; return; }
  // logical row r = 0 .. have-1 (oldest first) lives at physical row (rows_written - have + r) % L
  const long long base = rows_written - have;
  auto at = [&](int r) -> int64_t { return (int64_t)((base + r) % L) * N + e; };
  const bool fin = finished != 0;
  auto step_at = [&](int r) -> int32_t { return (fin && r == have - 1) ? final_rec[e].step : log.rows[at(r)].step; };
  const int32_t s_new = step_at(have - 1);
  int r0 = have - 1 - s_new;
  if (r0 < 0) r0 = 0;
  if (tid == 0) s_start = r0;
  __syncthreads();
  for (int r = r0 + 1 + tid; r < have; r += blockDim.x)
    if (step_at(r - 1) != step_at(r) - 1) atomicMax(&s_start, r);
  __syncthreads();
  int start = s_start;
  int n = have - start;
  if (n > max_rows) { start = have - max_rows; n = max_rows; }
  if (tid == 0) o.n_rows[j] = n;
  for (int r = tid; r < n; r += blockDim.x) {
    const int rr = start + r;
    const int64_t k = at(rr), d = (int64_t)j * max_rows + r;
    if (fin && rr == have - 1) {
      const EnvRec t = final_rec[e];
      o.idx[d] = t.idx; o.step[d] = t.step; o.pos[d] = t.pos; o.dsi[d] = t.dsi;
      o.pv[d] = t.pv; o.realpos[d] = t.realpos; o.reward[d] = reward64[e];
      o.asset[d] = t.asset; o.fiat[d] = t.fiat; o.ia[d] = t.ia; o.ifi[d] = t.ifi;
    } else {
      const LogRow w = log.rows[k];
      o.idx[d] = w.idx; o.step[d] = w.step; o.pos[d] = w.pos; o.dsi[d] = w.dsi;
      o.pv[d] = w.pv; o.realpos[d] = w.realpos; o.reward[d] = w.reward;
      o.asset[d] = w.asset; o.fiat[d] = w.fiat; o.ia[d] = w.ia; o.ifi[d] = w.ifi;
    }
    o.flags[d] = log.rows[k].flags;
  }
}

hipError_t launch_pack_log(const LogArrays& log, int N, int L, long long rows_written, const int32_t* ids,
                           int n_ids, int max_rows, int finished, const EnvRec* final_rec,
                           const double* reward64, const LogPack& o, hipStream_t stream) {
  hipLaunchKernelGGL(gte_pack_log_kernel, dim3(n_ids), dim3(256), 0, stream, log, N, L, rows_written, ids,
                     max_rows, finished, final_rec, reward64, o);
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
