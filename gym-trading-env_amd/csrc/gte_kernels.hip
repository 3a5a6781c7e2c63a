// gte_kernels.hip — the step / reset kernels of libgte (gfx950 / CDNA4 only).
//
// One launch advances every environment of the shard by one step
// (TradingEnv.step, reference environments.py:233-272) or resets the masked
// ones (TradingEnv.reset, :163-199).  Two phases inside one wavefront, no LDS
// and no workgroup barrier:
//
//   phase A  one lane per environment: the scalar fp64 state machine
//            (_take_action/_trade -> Portfolio.trade_to_position ->
//            update_interest -> valorisation -> done/truncated -> reward),
//            auto-reset with Philox or injected draws, and wave-level
//            compaction of the terminal mask (__ballot + popcount prefix, one
//            atomic per wave).  State is a struct of arrays, so the loads and
//            stores of the active lanes are contiguous.
//   phase B  the whole wave copies the EPW observation windows (_get_obs,
//            :152-160): the window of env e is ONE contiguous block of
//            W*F_obs floats of the row-major feature table, moved with
//            16-byte loads/stores (1 KiB per wave instruction) and patched in
//            flight with the dynamic columns, which come from a small per-env
//            store in HBM (previous rows) and from phase A's registers
//            (current row, broadcast with v_readlane / ds_bpermute).
//
// The kernel is HBM-bound (no contraction, so no MFMA): >= 97 % of its bytes
// are the window gather + observation store.  See DESIGN.md for the roofline.
#include "gte_device.h"

namespace gte {

typedef float float4_t __attribute__((ext_vector_type(4)));

enum { MODE_STEP = 0, MODE_RESET = 1 };

struct EnvRegs {
  int32_t idx, step, pos, dsi, start, episode, needs_reset;
  Portfolio q;
  double pv, realpos;
};

__device__ inline void load_state(const Params& p, int e, EnvRegs& s) {
  s.idx = p.idx[e]; s.step = p.step[e]; s.pos = p.pos[e]; s.dsi = p.dsi[e];
  s.start = p.start[e]; s.episode = p.episode[e]; s.needs_reset = p.needs_reset[e];
  s.q.asset = p.asset[e]; s.q.fiat = p.fiat[e]; s.q.ia = p.ia[e]; s.q.ifi = p.ifi[e];
  s.pv = p.pv[e]; s.realpos = p.realpos[e];
}

__device__ inline void store_state(const Params& p, int e, const EnvRegs& s) {
  p.idx[e] = s.idx; p.step[e] = s.step; p.pos[e] = s.pos; p.dsi[e] = s.dsi;
  p.start[e] = s.start; p.episode[e] = s.episode; p.needs_reset[e] = s.needs_reset;
  p.asset[e] = s.q.asset; p.fiat[e] = s.q.fiat; p.ia[e] = s.q.ia; p.ifi[e] = s.q.ifi;
  p.pv[e] = s.pv; p.realpos[e] = s.realpos;
}

// MultiDatasetTradingEnv.next_dataset, environments.py:380-391
__device__ inline void next_dataset(const Params& p, int e, int32_t inj_ds, EnvRegs& s,
                                    bool& fresh) {
  const int32_t n = p.n_picks[e];
  p.n_picks[e] = n + 1;
  s.dsi = (inj_ds >= 0) ? inj_ds : perm_pick(p, e, n / p.D, n % p.D);
  p.eps_on_ds[e] = 0;               // :381
  if (p.persist) fresh = true;      // _set_df rebuilds _obs_array (:135-141)
}

// TradingEnv.reset, environments.py:163-199 (+ MultiDataset reset :393-400)
__device__ inline void do_reset(const Params& p, int e, int32_t inj_idx, int32_t inj_pos,
                                int32_t inj_ds, EnvRegs& s, bool& fresh) {
  if (p.D > 1) {  // :394-398
    const int32_t n = p.eps_on_ds[e] + 1;
    p.eps_on_ds[e] = n;
    if (n % p.switch_every == 0) next_dataset(p, e, inj_ds, s, fresh);
  }
  uint32_t r[4];
  reset_draws(p, e, s.episode, 0x52534554u, r);
  s.episode += 1;
  s.step = 0;  // :166
  int32_t pi = p.init_pos_index;  // :167
  if (pi < 0) pi = (inj_pos >= 0) ? inj_pos : bounded(r[0], p.P);
  s.pos = pi;
  int32_t idx = p.has_window ? p.W - 1 : 0;  // :171-172
  const DatasetDesc d = p.ds[s.dsi];
  if (p.max_dur > 0) {  // :173-177 randint(low=idx, high=T - max_dur - idx)
    const int32_t low = idx;
    const int32_t high = (int32_t)d.T - p.max_dur - idx;
    idx = (inj_idx >= 0) ? inj_idx : low + bounded(r[1], high - low);
  }
  s.idx = idx;
  s.start = idx;
  const double position = p.positions[pi];  // TargetPortfolio, portfolio.py:59-66
  const double price = d.close[idx];
  s.q.asset = position * p.V0 / price;
  s.q.fiat = (1.0 - position) * p.V0;
  s.q.ia = 0.0;
  s.q.ifi = 0.0;
  s.pv = p.V0;          // :194
  s.realpos = position; // :192
  s.needs_reset = 0;
}

__device__ inline void pop_injection(const Params& p, int e, int32_t& qi, int32_t& qp,
                                     int32_t& qd) {
  qi = qp = qd = -1;
  if (p.q_n <= 0) return;
  const int32_t h = p.q_head[e];
  if (h >= p.q_n) return;
  p.q_head[e] = h + 1;
  const int64_t k = (int64_t)e * p.q_n + h;
  if (p.q_idx) qi = p.q_idx[k];
  if (p.q_pos) qp = p.q_pos[k];
  if (p.q_ds) qd = p.q_ds[k];
}

// Dynamic features of the current row (:153-154) -> the env's store, and the
// description of the window copy for phase B.
__device__ inline void make_job(const Params& p, int e, const EnvRegs& s, bool fresh,
                                ObsJob& job) {
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) {
    float v = 0.0f;
    if (i < p.nd) {
      const double x = (p.dyn_kind[i] == GTE_DYN_REAL_POSITION) ? s.realpos   // :23-24
                                                                : p.positions[s.pos];  // :20-21
      v = (float)x;
      const int64_t slot = p.persist ? (int64_t)s.idx : (int64_t)(s.idx % p.W);
      p.ring[((int64_t)e * p.depth + slot) * p.nd + i] = v;
    }
    job.cur[i] = v;
  }
  const int32_t first = s.idx - p.W + 1;  // first row of the window (:159)
  job.src = p.ds[s.dsi].feat + (int64_t)first * p.Fobs;
  job.slot0 = p.persist ? first : (s.idx + 1) % p.W;
  int32_t nz;
  if (fresh) nz = p.W - 1;            // brand-new _obs_array: only the current row is set
  else if (p.persist) nz = 0;
  else {
    nz = s.start - first;             // rows before the episode start were never written
    nz = nz < 0 ? 0 : (nz > p.W - 1 ? p.W - 1 : nz);
  }
  job.n_zero = nz;
  job.idx = s.idx;
  job.flags = 1 | ((fresh && p.persist) ? 2 : 0);
}

// ---------------------------------------------------------------------------
// phase A

template <int MODE>
__device__ inline void phase_a(const Params& p, int e, bool active, int lane, ObsJob& job) {
  job.src = nullptr; job.slot0 = 0; job.n_zero = 0; job.idx = 0; job.flags = 0;
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) job.cur[i] = 0.0f;
  bool ended = false;

  if (MODE == MODE_RESET) {
    if (active && (p.mask == nullptr || p.mask[e] != 0)) {
      EnvRegs s;
      load_state(p, e, s);
      bool fresh = false;
      const int32_t ii = p.inj_idx ? p.inj_idx[e] : -1;
      const int32_t ip = p.inj_pos ? p.inj_pos[e] : -1;
      const int32_t id = p.inj_ds ? p.inj_ds[e] : -1;
      if (p.D > 1 && p.n_picks[e] == 0) next_dataset(p, e, id, s, fresh);  // ctor pick, :378
      do_reset(p, e, ii, ip, id, s, fresh);
      store_state(p, e, s);
      p.reward[e] = 0.0f; p.reward64[e] = 0.0;
      p.terminated[e] = 0; p.truncated[e] = 0;
      make_job(p, e, s, fresh, job);
    }
    return;
  }

  // MODE_STEP — TradingEnv.step, environments.py:233-272
  if (active) {
    EnvRegs s;
    load_state(p, e, s);
    const int32_t action = p.actions[e];
    bool fresh = false;
    bool stepped = true;
    if (s.needs_reset) {
      if (p.autoreset == GTE_AUTORESET_NEXT_STEP) {
        int32_t qi, qp, qd;
        pop_injection(p, e, qi, qp, qd);
        do_reset(p, e, qi, qp, qd, s, fresh);
        p.reward[e] = 0.0f; p.reward64[e] = 0.0;
        p.terminated[e] = 0; p.truncated[e] = 0;
        stepped = false;
      } else if (s.idx >= (int32_t)p.ds[s.dsi].T - 1) {
        // no auto-reset and no row left: the reference raises IndexError (:239);
        // the batch leaves such an env frozen, flags still raised
        p.reward[e] = 0.0f; p.reward64[e] = 0.0;
        stepped = false;
        ended = true;  // its flags stay raised, so it stays in the terminal list
      }
    }
    if (stepped) {
      const DatasetDesc d = p.ds[s.dsi];
      if (action >= 0) {  // :234 -> :213-215: trade only when the position VALUE differs
        const double position = p.positions[action];
        if (position != p.positions[s.pos]) {
          trade_to_position(s.q, position, d.close[s.idx], p.fees);  // :204-209
          s.pos = action;                                             // :210
        }
      }
      s.idx += 1;   // :235
      s.step += 1;  // :236
      const double price = d.close[s.idx];  // :239
      s.q.ia = pymax0(-s.q.asset) * p.rate;   // update_interest, portfolio.py:44-46
      s.q.ifi = pymax0(-s.q.fiat) * p.rate;
      const double pv = valorisation(s.q, price);  // :241
      const bool done = (pv / p.V0) <= 0.7;        // :246
      bool trunc = s.idx >= (int32_t)d.T - 1;      // :248
      if (p.max_dur > 0 && s.step >= p.max_dur - 1) trunc = true;  // :250
      s.realpos = (s.q.asset - s.q.ia) * price / valorisation(s.q, price);  // :259
      double rew = 0.0;                            // :263, stays 0 when done (:265)
      if (!done) rew = reward_of(p, pv, s.pv);
      s.pv = pv;
      p.reward64[e] = rew;
      p.reward[e] = (float)rew;
      p.terminated[e] = done ? 1 : 0;
      p.truncated[e] = trunc ? 1 : 0;
      ended = done || trunc;
      if (ended) s.needs_reset = 1;
      if (ended && p.autoreset == GTE_AUTORESET_SAME_STEP) {
        int32_t qi, qp, qd;
        pop_injection(p, e, qi, qp, qd);
        do_reset(p, e, qi, qp, qd, s, fresh);
      }
    }
    store_state(p, e, s);
    make_job(p, e, s, fresh, job);
  }

  // terminal-mask compaction: one atomic per wave, ids in lane order within a wave
  const unsigned long long m = __ballot(ended);
  if (m != 0ull) {
    const int cnt = __popcll(m);
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(p.term_count, cnt);
    base = __shfl(base, leader);
    if (ended) {
      const int my = __popcll(m & ((1ull << lane) - 1ull));
      p.term_ids[base + my] = e;
    }
  }
}

// ---------------------------------------------------------------------------
// phase B

__device__ inline uint32_t fastdiv40(uint32_t k, uint64_t magic) {
  return (uint32_t)(((uint64_t)k * magic) >> 40);
}

template <bool NT, typename T>
__device__ inline void store_out(T* dst, const T& v) {
  if (NT) __builtin_nontemporal_store(v, dst);
  else *dst = v;
}

// job fields of the env a lane (flat path) or the whole wave (rows path) works on
struct JobView {
  const float* src;
  int32_t slot0, n_zero, flags;
  float cur[GTE_MAX_DYN];
};

// every lane reads the job of env `el` (per-lane el): ds_bpermute
__device__ inline JobView shuffle_job(const ObsJob& job, int el) {
  JobView b;
  const uint64_t a = (uint64_t)job.src;
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)a, el);
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(a >> 32), el);
  b.src = (const float*)(((uint64_t)hi << 32) | lo);
  b.slot0 = __shfl(job.slot0, el);
  b.n_zero = __shfl(job.n_zero, el);
  b.flags = __shfl(job.flags, el);
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) b.cur[i] = __shfl(job.cur[i], el);
  return b;
}

// the whole wave reads the job of env `el` (wave-uniform el): v_readlane -> SGPRs
__device__ inline JobView broadcast_job(const ObsJob& job, int el) {
  JobView b;
  const uint64_t a = (uint64_t)job.src;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, el);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(a >> 32), el);
  b.src = (const float*)(((uint64_t)hi << 32) | lo);
  b.slot0 = __builtin_amdgcn_readlane(job.slot0, el);
  b.n_zero = __builtin_amdgcn_readlane(job.n_zero, el);
  b.flags = __builtin_amdgcn_readlane(job.flags, el);
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i)
    b.cur[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(job.cur[i]), el));
  return b;
}

// Overwrite the dynamic columns that vector `v` (columns col .. col+VEC-1 of window
// row w) covers.  Every array index below is a compile-time constant after
// unrolling: a run-time index into cur[] or v[] would put them in scratch memory
// (one scratch store per observation store — measured as 2x WRITE_SIZE).
template <int VEC, typename vec_t>
__device__ inline void patch_dynamic(const Params& p, vec_t& v, const float* ring_e, int w,
                                     int col, const JobView& jb) {
  if (col + VEC <= p.Fs) return;  // all static columns
  int32_t slot = jb.slot0 + w;
  if (!p.persist && slot >= p.W) slot -= p.W;
  const bool is_cur = (w == p.W - 1);          // current row: phase A's registers
  const bool from_store = !is_cur && w >= jb.n_zero;  // else: never written -> 0
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) {
    if (i < p.nd) {
      const int c = p.Fs + i - col;  // component of v that holds dynamic feature i
      if (c >= 0 && c < VEC) {
        float x = is_cur ? jb.cur[i] : 0.0f;
        if (from_store) x = ring_e[(int64_t)slot * p.nd + i];
        if (VEC == 1) {
          v = x;
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) v[k] = (c == k) ? x : v[k];
        }
      }
    }
  }
}

// "flat" gather — any shape.  The wave's n_env*VPE vectors form one index space,
// lane l of iteration t handles vector t*64+l, so every wave instruction stores
// 64*VEC*4 contiguous bytes whatever the window size (also when an env's window is
// smaller than one wave instruction, e.g. windows=None).  The env differs per
// lane, so the job fields travel through ds_bpermute.
template <int VEC, bool NT, int U>
__device__ inline void phase_b_flat(const Params& p, int wave_first, int n_env, int lane,
                                    const ObsJob& job, uint64_t vpe_magic, uint64_t fv_magic) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  const uint32_t VPE = V / VEC;                // vectors per env
  const uint32_t FV = (uint32_t)p.Fobs / VEC;  // vectors per row
  const uint32_t total = (uint32_t)n_env * VPE;
  float* const obs0 = p.obs + (int64_t)wave_first * V;

  for (uint32_t k0 = 0; k0 < total; k0 += 64u * U) {
    vec_t v[U];
    JobView jb[U];
    uint32_t jj[U], ee[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t k = k0 + (uint32_t)u * 64u + (uint32_t)lane;
      const bool in = k < total;
      const uint32_t kk = in ? k : 0u;
      ee[u] = fastdiv40(kk, vpe_magic);
      jj[u] = kk - ee[u] * VPE;
      jb[u] = shuffle_job(job, (int)ee[u]);  // all lanes take part in the shuffles
      ok[u] = in && (jb[u].flags & 1);
      if (ok[u]) v[u] = *(const vec_t*)(jb[u].src + (int64_t)jj[u] * VEC);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const uint32_t k = k0 + (uint32_t)u * 64u + (uint32_t)lane;
      const uint32_t w = fastdiv40(jj[u], fv_magic);
      const int col = (int)(jj[u] - w * FV) * VEC;
      const float* ring_e = p.ring + (int64_t)(wave_first + (int)ee[u]) * p.depth * p.nd;
      patch_dynamic<VEC>(p, v[u], ring_e, (int)w, col, jb[u]);
      store_out<NT>((vec_t*)(obs0 + (int64_t)k * VEC), v[u]);
    }
  }
}

// "rows" gather — windows of at least one wave instruction (VPE >= 64 vectors).
// The wave walks its envs one at a time; the env's job sits in SGPRs (v_readlane),
// so a load is `global_load_dwordx4 v, v_off, s[base]`.  Chunks of 64*U vectors are
// software-pipelined: the loads of chunk c+1 are issued before the stores of chunk
// c, so a wave keeps 2*U KiB of loads in flight and never waits for its own stores
// (on CDNA4 vmcnt counts stores and loads together, in order).
template <int VEC, bool NT, int U>
__device__ inline void phase_b_rows(const Params& p, int wave_first, int n_env, int lane,
                                    const ObsJob& job, uint64_t fv_magic) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  const uint32_t VPE = V / VEC;
  const uint32_t FV = (uint32_t)p.Fobs / VEC;
  constexpr uint32_t CH = 64u * U;  // vectors per chunk

  // chunk cursor: (env in wave, first vector of the chunk); wave-uniform
  int el = 0;
  uint32_t j0 = 0;
  auto load_chunk = [&](int e_l, uint32_t j_0, vec_t (&v)[U], JobView& jb) {
    jb = broadcast_job(job, e_l);
    if (!(jb.flags & 1)) return;  // uniform
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = j_0 + (uint32_t)u * 64u + (uint32_t)lane;
      if (j < VPE) v[u] = *(const vec_t*)(jb.src + (int64_t)j * VEC);
    }
  };
  auto store_chunk = [&](int e_l, uint32_t j_0, vec_t (&v)[U], const JobView& jb) {
    if (!(jb.flags & 1)) return;
    float* const dst = p.obs + (int64_t)(wave_first + e_l) * V;
    const float* ring_e = p.ring + (int64_t)(wave_first + e_l) * p.depth * p.nd;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = j_0 + (uint32_t)u * 64u + (uint32_t)lane;
      if (j < VPE) {
        const uint32_t w = fastdiv40(j, fv_magic);
        const int col = (int)(j - w * FV) * VEC;
        patch_dynamic<VEC>(p, v[u], ring_e, (int)w, col, jb);
        store_out<NT>((vec_t*)(dst + (int64_t)j * VEC), v[u]);
      }
    }
  };

  vec_t va[U], vb[U];
  JobView ja, jbn;
  load_chunk(el, j0, va, ja);
  while (el < n_env) {
    int el_n = el;
    uint32_t j0_n = j0 + CH;
    if (j0_n >= VPE) { j0_n = 0; el_n = el + 1; }
    if (el_n < n_env) load_chunk(el_n, j0_n, vb, jbn);
    store_chunk(el, j0, va, ja);
#pragma unroll
    for (int u = 0; u < U; ++u) va[u] = vb[u];
    ja = jbn;
    el = el_n;
    j0 = j0_n;
  }
}

// zero the dynamic store of envs that switched dataset in persist mode (the
// reference rebuilds _obs_array in _set_df), except the current row's slot
__device__ inline void zero_fresh_stores(const Params& p, int wave_first, int n_env, int lane,
                                         const ObsJob& job) {
  for (int el = 0; el < n_env; ++el) {
    const int flags = __builtin_amdgcn_readlane(job.flags, el);
    if (!(flags & 2)) continue;
    const int idx = __builtin_amdgcn_readlane(job.idx, el);
    float* ring_e = p.ring + (int64_t)(wave_first + el) * p.depth * p.nd;
    const int64_t n = p.depth * p.nd;
    const int64_t keep_lo = (int64_t)idx * p.nd, keep_hi = keep_lo + p.nd;
    for (int64_t k = lane; k < n; k += 64)
      if (k < keep_lo || k >= keep_hi) ring_e[k] = 0.0f;
  }
}

// ROWS_U == 0: flat gather; ROWS_U = 1..4: rows gather with chunks of 64*ROWS_U vectors
template <int MODE, int VEC, bool NT, int ROWS_U>
__global__ __launch_bounds__(256) void gte_kernel(const Params p, const uint64_t vpe_magic,
                                                  const uint64_t fv_magic) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // the terminal counter has two slots used alternately, so no memset launch is
  // needed between steps: this launch clears the slot the NEXT launch will use
  if (MODE == MODE_STEP && blockIdx.x == 0 && threadIdx.x == 0) p.term_count_next[0] = 0;
  const int wave_first = wave * p.epw;
  if (wave_first >= p.N) return;  // whole wave exits together
  const int n_env = min(p.epw, p.N - wave_first);
  const int e = wave_first + lane;
  ObsJob job;
  phase_a<MODE>(p, e, lane < n_env, lane, job);
  if (p.persist) zero_fresh_stores(p, wave_first, n_env, lane, job);
  if (ROWS_U == 0) phase_b_flat<VEC, NT, 4>(p, wave_first, n_env, lane, job, vpe_magic, fv_magic);
  else phase_b_rows<VEC, NT, (ROWS_U ? ROWS_U : 1)>(p, wave_first, n_env, lane, job, fv_magic);
}

// ---------------------------------------------------------------------------
// launchers (called from gte_api.hip)

static uint64_t magic40(uint32_t d) { return ((1ull << 40) + d - 1) / d; }

// rows_u: 0 = flat gather, 1..4 = rows gather (only with 16-byte vectors)
template <int MODE>
static hipError_t launch_mode(const Params& p, int vec, bool nt, int rows_u, int blocks,
                              int threads, hipStream_t stream) {
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  const uint64_t vm = magic40(V / vec), fm = magic40((uint32_t)p.Fobs / vec);
#define GTE_LAUNCH(VEC, NT, RU) \
  hipLaunchKernelGGL((gte_kernel<MODE, VEC, NT, RU>), dim3(blocks), dim3(threads), 0, stream, p, vm, fm)
#define GTE_LAUNCH_NT(VEC, RU) do { if (nt) GTE_LAUNCH(VEC, true, RU); else GTE_LAUNCH(VEC, false, RU); } while (0)
  if (vec == 4) {
    switch (rows_u) {
      case 1: GTE_LAUNCH_NT(4, 1); break;
      case 2: GTE_LAUNCH_NT(4, 2); break;
      case 3: GTE_LAUNCH_NT(4, 3); break;
      case 4: GTE_LAUNCH_NT(4, 4); break;
      default: GTE_LAUNCH_NT(4, 0); break;
    }
  } else {
    GTE_LAUNCH_NT(1, 0);
  }
#undef GTE_LAUNCH_NT
#undef GTE_LAUNCH
  return hipGetLastError();
}

hipError_t launch_step(const Params& p, int vec, bool nt, int rows_u, int blocks, int threads,
                       hipStream_t stream) {
  return launch_mode<MODE_STEP>(p, vec, nt, rows_u, blocks, threads, stream);
}

hipError_t launch_reset(const Params& p, int vec, bool nt, int rows_u, int blocks, int threads,
                        hipStream_t stream) {
  return launch_mode<MODE_RESET>(p, vec, nt, rows_u, blocks, threads, stream);
}

}  // namespace gte
