// gte_rollout.hip — K consecutive TradingEnv.step calls (environments.py:233-272) in ONE
// launch, for action sequences known in advance (backtests of precomputed strategies,
// random-policy collection).  Own translation unit: nothing here can perturb the code
// generated for the per-step kernel (gte_hot.hip).
//
// Environments are independent, so a workgroup simply takes its own envs through the K steps.
// Two kernels:
//
//   gte_rollout_resident_kernel  (the default)  LDS staging of the sliding feature window
//     (_get_obs, environments.py:156-160): each env's W-1 older window rows live in LDS for the
//     whole launch, as a ring indexed by table row; per step ONE new row per env is fetched
//     from the feature table, the observation is emitted LDS -> HBM, and the new row then
//     replaces the oldest one.  The dynamic columns are part of the LDS rows (patched when a
//     row enters), so the emit is a pure copy.  A reset (or a dataset switch) re-anchors the
//     env: its window is refilled from the table.  Wave 0 keeps every env's state in REGISTERS
//     across the K steps (lane = env) and runs phase A one step ahead of the emit; the
//     dynamic ring in HBM and the per-step returns are written through and the record is
//     stored after the last step, so the env's state after the launch is exactly that of K
//     gte_step calls.  LDS holds ~64 envs
//     per CU at the headline shape (W-1 = 19 rows x 128 B = 2 432 B per env), so a batch
//     runs as several rounds of workgroups, each going through all K steps.
//   gte_rollout_kernel  (shapes whose window does not fit: fallback)  gathers every step's
//     windows from the table (L2) like the step kernel does; W-deep dynamic rings in LDS.
//
// Same arithmetic, same auto-reset, same injection queue and limit-order fills as gte_step:
// tests/test_gpu_rollout.py checks a rollout against K single steps bit for bit.
// Shapes: 16-byte vectors, W-deep rings (no dyn_persist), no final_obs; gte_rollout() falls
// back to K launches of the step kernel otherwise.
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
  int32_t epb;             // resident kernel: envs per workgroup
  int32_t n_groups;        // resident kernel: ceil(N / epb) groups of envs, handed out through ...
  int32_t* group_counter;  // ... this device counter (zeroed before the launch)
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


// ---------------------------------------------------------------------------------------
// Window-resident rollout

struct alignas(16) ResAux {
  int32_t n_zero;   // leading window rows whose dynamic columns read as zero (make_job)
  int32_t hslot0;   // slot of the window's first row in the env's W-deep dynamic ring in HBM
  int32_t pad0, pad1;
};

// LDS image of a workgroup of the resident kernel.  JobRec.meta here: bit0 the slot holds an
// env, bit1 (re)fill the env's LDS window from the table this step, bits 2..16 LDS ring slot of
// the window's first row (= table row mod (W-1)).
struct ResLds {
  JobRec* job;   // [EPB]
  float* cur;    // [EPB][GTE_MAX_DYN] dynamic features of the current row
  ResAux* aux;   // [EPB] what a refill needs besides the job
  int32_t* ctr;  // [0] next unclaimed emit chunk, [1] envs to refill this step
  float* win;    // [EPB][W-1][Fobs]
};

__device__ inline ResLds carve_res(unsigned char* b, int EPB) {
  ResLds L;
  L.job = (JobRec*)b;  b += 16 * EPB;
  L.cur = (float*)b;   b += 4 * GTE_MAX_DYN * EPB;
  L.aux = (ResAux*)b;  b += sizeof(ResAux) * EPB;
  L.ctr = (int32_t*)b; b += 16;
  L.win = (float*)b;
  return L;
}

size_t resident_lds_bytes(const Params& p, int epb) {
  return (size_t)epb * (16 + 4 * GTE_MAX_DYN + sizeof(ResAux)) + 16 +
         (size_t)epb * (size_t)(p.W - 1) * (size_t)p.Fobs * 4;
}

// threads that carry the newest window row of the workgroup's envs from the table to the
// observation and then into the LDS ring: waves 1..3, at most RES_NEW vectors each
#define RES_NEW 2
#define RES_OWNERS 192

// One group of r.epb envs through all K steps (a workgroup takes group after group, below).
template <int NT>
__device__ __forceinline__ void resident_group(const Params& p0, const RolloutArgs& r,
                                               const uint64_t vl_magic, const uint64_t fv_magic,
                                               unsigned char* gte_smem, const int group) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wib = tid >> 6;
  const int EPB = r.epb;
  const int wg_first = group * EPB;
  if (wg_first >= p0.N) return;
  const int n_wg = min(EPB, p0.N - wg_first);
  const ResLds L = carve_res(gte_smem, EPB);
  const int W1 = p0.W - 1;
  const uint32_t FV = (uint32_t)p0.Fobs / 4u;  // vectors per row
  const uint32_t VL = (uint32_t)W1 * FV;       // vectors per env held in LDS
  const int64_t V = (int64_t)p0.W * p0.Fobs;   // floats per observation

  if (tid < EPB) {
    const int slot = wg_first + tid;
    L.job[tid].env = (tid < n_wg) ? (p0.perm ? p0.perm[slot] : slot) : -1;
    L.job[tid].meta = 0;
  }
  if (tid == 0) { L.ctr[0] = 0; L.ctr[1] = 0; }

  // ---- wave 0: lane = LDS slot = one env, its state in registers for the whole launch
  const bool owns = wib == 0 && lane < EPB;
  const bool active = owns && lane < n_wg;
  const int e0 = active ? (p0.perm ? p0.perm[wg_first + lane] : wg_first + lane) : 0;
  EnvRegs s = {};
  if (active) load_state(p0, e0, s);
  int32_t act = active ? r.actions[e0] : -1;  // the next step's action, loaded one step ahead
  uint64_t prev_src = 0;
  int32_t prev_nz = -1;
  PriceCarry pc = {0.0, 0.0, -1, 0};
  ObsJob job;

  auto run_a = [&](int k) {  // phase A of step k (wave 0), from and into the registers
    const Params p = step_params(p0, r, k);
    const int32_t a = act;
    if (active && k + 1 < r.K) act = r.actions[(int64_t)(k + 1) * p0.N + e0];
    double pv = 0.0;
    phase_a<MODE_STEP>(p, e0, active, lane, job, nullptr, /*compact=*/k == r.K - 1, &pv, &s, &a,
                       /*write_record=*/k == r.K - 1, &pc);
    if (r.valuation && active) r.valuation[(int64_t)k * p0.N + e0] = pv;
    // An env that re-anchors (reset, dataset switch) will have its window refilled from the
    // table when this job is published, one step from now, by waves 1-3 — a read that misses
    // every cache while the chip is saturated with observation stores, stalling the workgroup.
    // This wave is a step ahead with time to spare: it touches those rows now (whole wave, one
    // env at a time, results discarded) so that the refill finds them in this CU's L1 / L2.
    if (k > 0 && r.obs) {
      const bool inc = (uint64_t)job.src == prev_src + (uint64_t)p0.Fobs * 4u &&
                       job.n_zero == (prev_nz > 0 ? prev_nz - 1 : 0);
      unsigned long long m = __ballot(active && (job.flags & 1) && !inc);
      float4_t sink = (float4_t)0.0f;
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const uint64_t src = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)job.src >> 32), l) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint64_t)job.src, l);
        for (uint32_t q = (uint32_t)lane; q < VL; q += 64u) sink += load_global<float4_t>(src, (int64_t)q);
      }
      asm volatile("" ::"v"(sink));  // keep the loads
    }
  };
  // wave 0 hands the job of the step about to be emitted to the workgroup.  An env whose window
  // moved on by exactly one row keeps its LDS rows; anything else (first step, reset, dataset
  // switch, a frozen finished env) is refilled from the table.
  auto publish = [&](bool all_fill) {
    if (!owns) return;  // (wave 0 only calls this)
    uint32_t meta = 0;
    if (active && (job.flags & 1)) {
      const int32_t first = job.idx - W1;
      const bool inc = !all_fill && (uint64_t)job.src == prev_src + (uint64_t)p0.Fobs * 4u &&
                       job.n_zero == (prev_nz > 0 ? prev_nz - 1 : 0);
      meta = 1u | (inc ? 0u : 2u) | ((uint32_t)(first % W1) << 2);
      prev_src = (uint64_t)job.src;
      prev_nz = job.n_zero;
    }
    L.job[lane].src = (uint64_t)job.src;
    L.job[lane].meta = meta;
#pragma unroll
    for (int i = 0; i < GTE_MAX_DYN; ++i) L.cur[lane * GTE_MAX_DYN + i] = job.cur[i];
    ResAux ax;
    ax.n_zero = job.n_zero; ax.hslot0 = job.slot0; ax.pad0 = 0; ax.pad1 = 0;
    L.aux[lane] = ax;
    const unsigned long long m = __ballot((meta & 2u) != 0u);
    if (lane == 0) L.ctr[1] = __popcll(m);
  };
  const int owner = tid - 64;  // 0..191 in waves 1-3
  // (re)fill the W-1 older rows of the flagged envs from the feature table, dynamic columns
  // resolved the way the step kernel's gather does (zero before the episode start, else the
  // env's ring in HBM); waves 1-3, one env at a time: first step and resets only
  auto fill = [&]() {
    for (int el = 0; el < n_wg; ++el) {
      const JobRec j = L.job[el];
      if (!(j.meta & 2u)) continue;  // workgroup-uniform
      const ResAux ax = L.aux[el];
      const int lslot0 = (int)((j.meta >> 2) & 0x7FFFu);
      const float* ring_e = p0.ring + (int64_t)j.env * p0.W * p0.nd;
      for (uint32_t q = (uint32_t)owner; q < VL; q += (uint32_t)RES_OWNERS) {
        const uint32_t w = fastdiv40(q, fv_magic);
        const uint32_t c = q - w * FV;
        float4_t v = load_global<float4_t>(j.src, (int64_t)q);
        if (c == FV - 1u && p0.nd > 0) {
          float x[GTE_MAX_DYN];
          int hs = ax.hslot0 + (int)w;
          hs -= (hs >= p0.W) ? p0.W : 0;
#pragma unroll
          for (int i = 0; i < GTE_MAX_DYN; ++i)
            x[i] = (i < p0.nd && (int)w >= ax.n_zero) ? ring_e[(int64_t)hs * p0.nd + i] : 0.0f;
          set_tail(v, p0.nd, x);
        }
        int ls = lslot0 + (int)w;
        ls -= (ls >= W1) ? W1 : 0;
        *(float4_t*)(L.win + ((int64_t)el * W1 + ls) * p0.Fobs + c * 4u) = v;
      }
    }
  };

  // newest-row duty of this thread (waves 1..3): vectors nq[i] of the workgroup's n_wg*FV
  float4_t nv[RES_NEW];
  int32_t n_el[RES_NEW], n_c[RES_NEW], n_ls[RES_NEW];
  bool n_ok[RES_NEW];
#pragma unroll
  for (int i = 0; i < RES_NEW; ++i) {
    const uint32_t n = (uint32_t)owner + (uint32_t)i * RES_OWNERS;
    const bool in = owner >= 0 && n < (uint32_t)n_wg * FV;
    const uint32_t el = fastdiv40(in ? n : 0u, fv_magic);
    n_el[i] = (int32_t)el; n_c[i] = (int32_t)((in ? n : 0u) - el * FV); n_ok[i] = in; n_ls[i] = 0;
    nv[i] = (float4_t)0.0f;
  }

  auto emit = [&](int k) {
    float* obs_k = r.obs ? r.obs + (int64_t)k * p0.N * V : p0.obs;
    // the newest row of every env: table -> registers (in flight during the LDS part)
    int32_t n_env[RES_NEW];
#pragma unroll
    for (int i = 0; i < RES_NEW; ++i) {
      n_env[i] = -1;
      if (n_ok[i]) {
        const JobRec j = L.job[n_el[i]];
        if (j.meta & 1u) {
          n_env[i] = j.env;
          n_ls[i] = (int32_t)((j.meta >> 2) & 0x7FFFu);  // the slot its oldest row leaves free
          nv[i] = load_global<float4_t>(j.src, (int64_t)VL + n_c[i]);
        }
      }
    }
    // the W-1 older rows: LDS -> observation, chunks of 64*4 vectors claimed by whoever is free
    const uint32_t total = (uint32_t)n_wg * VL;
    for (;;) {
      int cidx = 0;
      if (lane == 0) cidx = atomicAdd(&L.ctr[0], 1);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane(cidx) * 256u;
      if (lo >= total) break;
      float4_t v[4];
      int64_t dst[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t kq = lo + (uint32_t)u * 64u + (uint32_t)lane;
        const bool in = kq < total;
        const uint32_t kk = in ? kq : 0u;
        const uint32_t el = fastdiv40(kk, vl_magic);
        const uint32_t q = kk - el * VL;
        const uint32_t w = fastdiv40(q, fv_magic);
        const uint32_t c = q - w * FV;
        const JobRec j = L.job[el];
        int ls = (int)((j.meta >> 2) & 0x7FFFu) + (int)w;
        ls -= (ls >= W1) ? W1 : 0;
        ok[u] = in && (j.meta & 1u);
        v[u] = *(const float4_t*)(L.win + ((int64_t)el * W1 + ls) * p0.Fobs + c * 4u);
        dst[u] = (int64_t)j.env * V + (int64_t)q * 4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) store_out<NT>((float4_t*)(obs_k + dst[u]), v[u]);
    }
    // the newest row: dynamic columns from phase A, then to the observation
#pragma unroll
    for (int i = 0; i < RES_NEW; ++i) {
      if (n_env[i] < 0) continue;
      if (n_c[i] == (int32_t)FV - 1 && p0.nd > 0) {
        float x[GTE_MAX_DYN];
#pragma unroll
        for (int d = 0; d < GTE_MAX_DYN; ++d) x[d] = L.cur[n_el[i] * GTE_MAX_DYN + d];
        set_tail(nv[i], p0.nd, x);
      }
      store_out<NT>((float4_t*)(obs_k + (int64_t)n_env[i] * V + ((int64_t)VL + n_c[i]) * 4), nv[i]);
    }
    // (n_ok stays; n_env < 0 marks "nothing loaded" for the ring update below)
#pragma unroll
    for (int i = 0; i < RES_NEW; ++i) if (n_env[i] < 0) n_ls[i] = -1;
  };

  // Roles are disjoint branches: wave 0 only ever runs the state machine, waves 1-3 only ever
  // move windows, so the registers of one role are dead in the other (as one merged loop the
  // kernel needed 151 VGPRs = 3 workgroups per CU; the LDS budget is sized for 4).  Every wave
  // executes the same sequence of workgroup barriers.
  // Without per-step observations only the last step's window is ever needed: wave 0 runs the
  // K state-machine steps back to back from its registers, then one fill + emit.
  const int k_emit0 = r.obs ? 0 : r.K - 1;
  if (wib == 0) {
    for (int k = 0; k <= k_emit0; ++k) run_a(k);
    publish(true);
    GTE_LDS_BARRIER();  // (1) jobs of the first emitted step visible
    if (p0.debug & 1) return;
    GTE_LDS_BARRIER();  // (2) windows filled
    for (int k = k_emit0; k + 1 < r.K; ++k) {
      run_a(k + 1);     // one step ahead of the emit; nothing it touches is read by the emit
      GTE_LDS_BARRIER();  // (3) step k is emitted: jobs may change
      publish(false);
      GTE_LDS_BARRIER();  // (4) step k+1's jobs visible
      if (L.ctr[1] > 0) GTE_LDS_BARRIER();  // (5) refills done (workgroup-uniform)
    }
  } else {
    GTE_LDS_BARRIER();  // (1)
    if (p0.debug & 1) return;
    fill();
    GTE_LDS_BARRIER();  // (2)
    for (int k = k_emit0; k < r.K; ++k) {
      emit(k);
      if (k == r.K - 1) break;
      GTE_LDS_BARRIER();  // (3) step k is emitted: its oldest row may be replaced
#pragma unroll
      for (int i = 0; i < RES_NEW; ++i)
        if (n_ok[i] && n_ls[i] >= 0)
          *(float4_t*)(L.win + ((int64_t)n_el[i] * W1 + n_ls[i]) * p0.Fobs + n_c[i] * 4) = nv[i];
      if (tid == 64) L.ctr[0] = 0;
      GTE_LDS_BARRIER();  // (4) rings updated, step k+1's jobs visible
      if (L.ctr[1] > 0) { // workgroup-uniform: some env was reset, re-anchor its window
        fill();
        GTE_LDS_BARRIER();  // (5)
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// State-only rollout: the steps of a rollout that needs no observation from them (a backtest
// keeps rewards, flags and valuations; only the LAST step's observation exists afterwards, and
// that step runs as an ordinary gte_step launch).  One lane per env, state in registers for all
// n steps, next action loaded one step ahead, no LDS, no barrier: what is left per step is the
// fp64 state machine (prices are carried from step to step, PriceCarry).  Dynamic ring and
// per-step returns are written through exactly as gte_step would, the record once at the end.
__global__ __launch_bounds__(256) void gte_rollout_state_kernel(const Params p0, const RolloutArgs r,
                                                                const int n_steps, const int epw) {
  const int lane = threadIdx.x & 63;
  // epw envs per wavefront (the other lanes idle): the step is a dependent chain of loads and
  // fp64 issue, so two half-filled waves per SIMD overlap where one full wave waits
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) * epw + lane;
  const bool active = lane < epw && slot < p0.N;
  // (no L2-affinity permutation here: it exists for the window gathers, and with env = slot the
  // per-step returns of a wave's envs are neighbours in memory: 4.2 -> 3.95 us per step)
  const int e = active ? slot : 0;
  EnvRegs s = {};
  if (active) load_state(p0, e, s);
  int32_t act = active ? r.actions[e] : -1;
  PriceCarry pc = {0.0, 0.0, -1, 0};
  ObsJob job;
  auto run_a = [&](int k) {
    const Params p = step_params(p0, r, k);
    const int32_t a = act;
    if (active && k + 1 < n_steps) act = r.actions[(int64_t)(k + 1) * p0.N + e];
    double pv = 0.0;
    // the record is stored once, after the last step: its 128 B per env-step of scattered 16-byte
    // stores were most of what this kernel issued
    phase_a<MODE_STEP>(p, e, active, lane, job, nullptr, /*compact=*/false, &pv, &s, &a,
                       /*write_record=*/k == n_steps - 1, &pc);
    if (r.valuation && active) r.valuation[(int64_t)k * p0.N + e] = pv;
  };
  for (int k = 0; k < n_steps; ++k) run_a(k);
}

hipError_t launch_rollout_state(const Params& p, const RolloutArgs& r, int n_steps, int epw,
                                hipStream_t stream) {
  if (!hot_tu_covers(p)) return hipErrorInvalidValue;  // this TU is compiled with GTE_HOT_ONLY (gte_device.h)
  const int waves = (p.N + epw - 1) / epw;
  hipLaunchKernelGGL(gte_rollout_state_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, p, r, n_steps, epw);
  return hipGetLastError();
}

// The launch is a WORK QUEUE over groups of r.epb envs: min(groups, resident slots) workgroups, each
// taking the next group from a device counter until none is left.  Groups are independent (no
// workgroup ever waits for another), so this only removes the quantisation of a plain grid, where
// 6.1 "rounds" of workgroups cost 7: 100 003 envs are 6 251 groups on 1 024 slots.
template <int NT>
__global__ __launch_bounds__(256, 4) void gte_rollout_resident_kernel(const Params p0, const RolloutArgs r,
                                                                   const uint64_t vl_magic,
                                                                   const uint64_t fv_magic) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gte_smem[];
  if (blockIdx.x == 0 && threadIdx.x == 0) p0.term_count_next[0] = 0;
  int32_t* next_group = carve_res(gte_smem, r.epb).ctr + 2;
  for (;;) {
    if (threadIdx.x == 0) *next_group = atomicAdd(r.group_counter, 1);
    GTE_LDS_BARRIER();
    const int g = *next_group;
    if (g >= r.n_groups) break;  // workgroup-uniform: the queue is empty
    resident_group<NT>(p0, r, vl_magic, fv_magic, gte_smem, g);
    GTE_LDS_BARRIER();  // every wave is done with this group's LDS image (and has read `g`)
  }
}

int resident_blocks_per_cu(const Params& p, int epb, int nt) {
  int n = 0;
  const size_t smem = resident_lds_bytes(p, epb);
  hipError_t e = hipSuccess;
  if (smem > 64 * 1024) {  // beyond the default dynamic-LDS limit: opt in (160 KiB per CU on gfx950)
    if (nt == 2) e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    else if (nt == 1) e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    else e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  }
  if (nt == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_resident_kernel<2>, 256, smem);
  else if (nt == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_resident_kernel<1>, 256, smem);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gte_rollout_resident_kernel<0>, 256, smem);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

hipError_t launch_rollout_resident(const Params& p, const RolloutArgs& r, int nt, int blocks,
                                   hipStream_t stream) {
  if (!hot_tu_covers(p)) return hipErrorInvalidValue;  // this TU is compiled with GTE_HOT_ONLY (gte_device.h)
  // identity processing order: the windows live in LDS for the whole launch, so the L2-affinity
  // order has one table row per env-step left to serve, while it scatters every step's return
  // values (28.4 -> 28.0 us per step, profiles/r03_rollout_placement.log)
  Params q = p;
  q.perm = nullptr;
  const size_t smem = resident_lds_bytes(p, r.epb);
  auto magic = [](uint32_t d) { return ((1ull << 40) + d - 1) / d; };
  const uint32_t FV = (uint32_t)p.Fobs / 4u;
  const uint64_t vlm = magic((uint32_t)(p.W - 1) * FV), fm = magic(FV);
  if (smem > 64 * 1024) {  // the geometry search probed other sizes after this one: opt in again
    hipError_t e;
    if (nt == 2) e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    else if (nt == 1) e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    else e = hipFuncSetAttribute((const void*)gte_rollout_resident_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  if (nt == 2)
    hipLaunchKernelGGL((gte_rollout_resident_kernel<2>), dim3(blocks), dim3(256), smem, stream, q, r, vlm, fm);
  else if (nt == 1)
    hipLaunchKernelGGL((gte_rollout_resident_kernel<1>), dim3(blocks), dim3(256), smem, stream, q, r, vlm, fm);
  else
    hipLaunchKernelGGL((gte_rollout_resident_kernel<0>), dim3(blocks), dim3(256), smem, stream, q, r, vlm, fm);
  return hipGetLastError();
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
  if (!hot_tu_covers(p)) return hipErrorInvalidValue;  // this TU is compiled with GTE_HOT_ONLY (gte_device.h)
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
