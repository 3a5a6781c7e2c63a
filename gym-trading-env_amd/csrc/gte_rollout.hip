// gte_rollout.hip — K consecutive TradingEnv.step calls (environments.py:233-272) in ONE
// launch, for action sequences known in advance (backtests of precomputed strategies,
// random-policy collection).  Own translation unit: nothing here can perturb the code
// generated for the per-step kernel (gte_hot.hip).
//
// Environments are independent, so a workgroup simply iterates its own <= 64 envs through
// the K steps.  Wave 0 runs phase A (one lane per env) ONE STEP AHEAD into double-buffered job
// records while waves 1-3 copy the current step's windows; the copy index space is handed out
// in chunks through an LDS counter, so wave 0 joins the copy as soon as its phase A is done.
// Compared with K launches of the step kernel there is no dispatch ramp and no tail per step,
// phase A never leaves the memory pipes idle, the W-deep dynamic-feature rings stay in LDS for
// the whole launch (read once, updated in place; phase A still writes them through to HBM),
// and when no per-step observations are asked for the gather runs only for the last step.
//
// Same arithmetic, same auto-reset, same injection queue and limit-order fills as gte_step:
// tests/test_gpu_rollout.py checks a rollout against K single steps bit for bit.
// Shapes: 16-byte vectors, cooperative phase A (4*epw <= 64), W-deep rings (no dyn_persist),
// no final_obs; gte_rollout() falls back to K launches of the step kernel otherwise.
#define GTE_HOT_ONLY 1
#include "gte_kernels.hip"

namespace gte {

struct RolloutArgs {
  const int32_t* actions;  // [K][N]
  int32_t K;
  float* obs;              // [K][N][W][Fobs] or nullptr (last step only, into p.obs)
  float* reward;           // [K][N] or nullptr (p.reward, overwritten every step)
  double* reward64;
  uint8_t* terminated;
  uint8_t* truncated;
  double* valuation;       // [K][N] or nullptr
};

// LDS: two sets of job records (wave 0 runs phase A one step ahead of the gather), two chunk
// counters, one copy of the rings.
// (the set is selected by pointer arithmetic: an array of two WgLds indexed by k & 1 lands in
// scratch memory, and every job read of the copy loop would go through it)
struct RollLds {
  JobRec* job;    // [2][EPB]
  float* cur;     // [2][EPB][GTE_MAX_DYN]
  int32_t* idx;   // [2][EPB]
  int32_t* ctr;   // [2] next unclaimed gather chunk of the step using set i
  float* staged;  // [EPB][W][nd]
  int EPB;
  __device__ WgLds set(int i) const {
    WgLds L;
    L.job = job + i * EPB;
    L.cur = cur + i * EPB * GTE_MAX_DYN;
    L.idx = idx + i * EPB;
    L.fin = nullptr;
    L.staged = staged;
    return L;
  }
};

__device__ inline RollLds carve_roll(unsigned char* b, int EPB) {
  RollLds R;
  R.EPB = EPB;
  R.job = (JobRec*)b;   b += 2 * 16 * EPB;
  R.cur = (float*)b;    b += 2 * 4 * GTE_MAX_DYN * EPB;
  R.idx = (int32_t*)b;  b += 2 * 4 * EPB;
  R.ctr = (int32_t*)b;  b += 16;
  R.staged = (float*)b;
  return R;
}

size_t rollout_lds_bytes(const Params& p) {
  const size_t EPB = (size_t)p.epw * 4;
  return 2 * EPB * (16 + 4 * GTE_MAX_DYN + 4) + 16 + EPB * (size_t)p.W * (size_t)(p.nd ? p.nd : 1) * 4;
}

__device__ inline Params step_params(const Params& p0, const RolloutArgs& r, int k) {
  Params p = p0;
  const int64_t N = p0.N;
  p.actions = r.actions + (int64_t)k * N;
  if (r.reward) p.reward = r.reward + (int64_t)k * N;
  if (r.reward64) p.reward64 = r.reward64 + (int64_t)k * N;
  if (r.terminated) p.terminated = r.terminated + (int64_t)k * N;
  if (r.truncated) p.truncated = r.truncated + (int64_t)k * N;
  if (r.obs) p.obs = r.obs + (int64_t)k * N * (int64_t)p0.W * p0.Fobs;
  return p;
}

#define GTE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int NT>
__global__ __launch_bounds__(256) void gte_rollout_kernel(const Params p0, const RolloutArgs r,
                                                          const uint64_t vpe_magic,
                                                          const uint64_t fv_magic,
                                                          const uint64_t wnd_magic) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gte_smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) p0.term_count_next[0] = 0;
  const int EPB = p0.epw * 4;
  const int wg_first = blockIdx.x * EPB;
  if (wg_first >= p0.N) return;
  const int n_wg = min(EPB, p0.N - wg_first);
  const RollLds R = carve_roll(gte_smem, EPB);
  const int s_first = wib * p0.epw;
  const int n_env = min(p0.epw, n_wg - s_first);

  // env ids of this wave's slots (both job sets), then their rings into LDS: once per launch
  if (lane < p0.epw) {
    const int slot = wg_first + s_first + lane;
    const int32_t env = (lane < n_env) ? (p0.perm ? p0.perm[slot] : slot) : -1;
    R.job[s_first + lane].env = env;
    R.job[EPB + s_first + lane].env = env;
  }
  if (threadIdx.x == 0) { R.ctr[0] = 0; R.ctr[1] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (n_env > 0) stage_raw_rings(p0, R.set(0), s_first, n_env, lane, wnd_magic);

  const bool owns = lane < EPB;           // wave 0: lane = LDS slot
  const bool active = owns && lane < n_wg;
  const int e0 = (wib == 0 && active) ? (p0.perm ? p0.perm[wg_first + lane] : wg_first + lane) : 0;
  const uint32_t VPE = (uint32_t)(p0.W * p0.Fobs) / 4u;
  const uint32_t total = (uint32_t)n_wg * VPE;
  const uint32_t CH = 64u * 4u;           // one pass of the copy loop per claimed chunk

  // phase A of step k for the whole workgroup (wave 0 only), jobs into buf[k & 1]
  auto run_a = [&](int k) {
    const Params p = step_params(p0, r, k);
    ObsJob job;
    double pv = 0.0;
    phase_a<MODE_STEP>(p, e0, active, lane, job, nullptr, /*compact=*/k == r.K - 1, &pv);
    if (owns) publish_job(R.set(k & 1), lane, job);
    if (r.valuation && active) r.valuation[(int64_t)k * p0.N + e0] = pv;
  };

  if (wib == 0) run_a(0);
  GTE_LDS_BARRIER();

  for (int k = 0; k < r.K; ++k) {
    const bool last = (k == r.K - 1);
    const WgLds L = R.set(k & 1);
    // wave 0 runs one step ahead; the other waves start copying step k's windows at once
    if (wib == 0 && !last) run_a(k + 1);
    if ((last || r.obs != nullptr) && !(p0.debug & 1)) {
      const Params p = step_params(p0, r, k);
      for (;;) {  // whoever is free claims the next chunk of the workgroup's index space
        int c = 0;
        if (lane == 0) c = atomicAdd(&R.ctr[k & 1], 1);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(c) * CH;
        if (lo >= total) break;
        phase_b<4, NT, STAGE_RAW, 4>(p, L, 0, n_wg, lane, vpe_magic, fv_magic, lo, lo + CH);
      }
    }
    if (last) break;
    GTE_LDS_BARRIER();  // step k's windows are copied, step k+1's jobs are published
    // step k's current row joins the LDS copy of the rings (phase A wrote the same values to
    // the rings in HBM): slot idx % W = slot0 + W - 1 (mod W)
    if (lane < n_env) {
      const int s = s_first + lane;
      const uint32_t m = L.job[s].meta;
      if (m & 1u) {
        int slot = meta_slot0(m) + p0.W - 1;
        slot -= (slot >= p0.W) ? p0.W : 0;
        for (int i = 0; i < p0.nd; ++i)
          L.staged[(s * p0.W + slot) * p0.nd + i] = L.cur[s * GTE_MAX_DYN + i];
      }
    }
    if (threadIdx.x == 0) R.ctr[k & 1] = 0;  // next used two steps from now
    GTE_LDS_BARRIER();  // rings updated before anyone copies step k+1
  }
}

// Workgroups of the rollout kernel one CU holds at once with p.epw envs per wavefront.
int rollout_blocks_per_cu(const Params& p, int nt) {
  int n = 0;
  const size_t smem = rollout_lds_bytes(p);
  hipError_t e;
  if (nt == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_kernel<2>, 256, smem);
  else if (nt == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_kernel<1>, 256, smem);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_kernel<0>, 256, smem);
  return e == hipSuccess ? n : 0;
}

hipError_t launch_rollout(const Params& p, const RolloutArgs& r, int nt, int blocks, int threads,
                          hipStream_t stream) {
  const size_t smem = rollout_lds_bytes(p);
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  auto magic = [](uint32_t d) { return ((1ull << 40) + d - 1) / d; };
  const uint64_t vm = magic(V / 4), fm = magic((uint32_t)p.Fobs / 4),
                 wm = magic((uint32_t)(p.W * (p.nd ? p.nd : 1)));
  if (nt == 2)
    hipLaunchKernelGGL((gte_rollout_kernel<2>), dim3(blocks), dim3(threads), smem, stream, p, r, vm, fm, wm);
  else if (nt == 1)
    hipLaunchKernelGGL((gte_rollout_kernel<1>), dim3(blocks), dim3(threads), smem, stream, p, r, vm, fm, wm);
  else
    hipLaunchKernelGGL((gte_rollout_kernel<0>), dim3(blocks), dim3(threads), smem, stream, p, r, vm, fm, wm);
  return hipGetLastError();
}

}  // namespace gte
