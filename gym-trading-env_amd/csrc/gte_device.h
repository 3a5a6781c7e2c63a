// gte_device.h — device-side pieces of libgte (gfx950 only).
//
// Per-environment fp64 state machine of TradingEnv.step()/reset()
// (reference: src/gym_trading_env/environments.py:163-272 and
// utils/portfolio.py:1-66), written one IEEE-754 double operation per Python
// operation, in the reference's order.  This translation unit is compiled
// with -ffp-contract=off: an FMA would change roundings and break the
// bit-exact agreement of the portfolio state with CPython.
#pragma once
#include <cstddef>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gte.h"

// Wavefronts per workgroup of the step kernel (wave 0 runs phase A for the whole workgroup,
// every wave gathers its own p.epw envs).  4 is the product; other values are for A/B builds
// only (tools/waves_ab.sh): the rollout kernels assume 4.
#ifndef GTE_WAVES
#define GTE_WAVES 4
#endif

namespace gte {

struct DatasetDesc {
  const float* feat;    // f32 [T, F_obs] row-major, dynamic columns zero (_obs_array)
  const double* close;  // f64 [T] (_price_array)
  const double* high;   // f64 [T] or null
  const double* low;
  int64_t T;
};

// Per-env state record: ONE 128-byte line per environment, so that an env reached
// through the L2-affinity permutation costs one line in and one line out (as a
// struct of arrays the same gather touched 13 sectors per env and cancelled the
// gain).  gte_get_state extracts struct-of-arrays views for the host on demand.
struct alignas(128) EnvRec {
  // hot half: what every step rewrites — ONE aligned 64-byte piece, which the step kernel writes as one
  // request per env (through LDS: four lanes per env); as five scattered stores from the lane that
  // stepped the env it was half of what phase A's stores cost (profiles/r03_step5_decoupled.log)
  int32_t idx, step, pos, dsi;               // 16 B
  double asset, fiat, ia, ifi, pv, realpos;  // 48 B
  // cold half: written where it changes (a reset, a limit-order fill, an episode end)
  int32_t start, episode, needs_reset, eps_on_ds;
  int32_t n_picks, q_head, lo_n, pad0;
  int32_t pad1[8];                           // -> 128 B
};
static_assert(offsetof(EnvRec, start) == 64, "EnvRec: the hot half is the first 64 bytes");
static_assert(sizeof(EnvRec) == 128, "EnvRec must be one 128-byte line");

// Trajectory log (optional, gte_config.log_steps): [L, N] ROWS, one 80-byte record per env after
// every reset / step — what History.add records (reference environments.py:253-264).  One record
// per (row, env), not one array per column: the step kernel writes a row through LDS as two
// requests per env, where the twelve columns were twelve scattered stores (one request each: what a
// scattered store costs is per request, profiles/r03_phase_a_stores.log).  Hosts see the columns as
// strided views (gte_log_view: env_stride / row_stride).
struct alignas(16) LogRow {
  int32_t idx, step, pos, dsi;                              // 16 B
  double pv, realpos, reward;                               // 24 B
  double asset, fiat, ia, ifi;  // Portfolio state: get_portfolio_distribution (portfolio.py:49-57)
  uint8_t flags;                // bit0 terminated, bit1 truncated
  uint8_t pad[7];                                           // -> 80 B
};
static_assert(sizeof(LogRow) == 80, "LogRow is five 16-byte pieces");
struct LogArrays {
  LogRow* rows;  // [L, N]
};

// Everything a launch needs; passed by value as the kernel argument.
struct Params {
  // --- configuration (from gte_config)
  int32_t N, D, Fs, nd, Fobs, W, has_window, P;
  int32_t dyn_kind[GTE_MAX_DYN];
  int32_t init_pos_index, max_dur, reward_kind, autoreset, switch_every, persist;
  int64_t depth;  // rows of the per-env dynamic-feature store (W, or max T when persist)
  double fees, rate, V0, rp0, rp1, rp2;
  uint64_t seed;
  int64_t env_id_base;
  // --- resident tables
  const DatasetDesc* ds;
  const double* positions;  // f64 [P]
  // --- per-env state, one 128-byte record per env
  EnvRec* rec;
  float* ring;  // f32 [N, depth, nd]
  // --- pending limit orders (null until the first gte_add_limit_orders)
  int32_t* lo_pos;      // i32 [N, P] target position index, insertion order
  double* lo_limit;     // f64 [N, P]
  uint8_t* lo_persist;  // u8  [N, P]
  // --- outputs
  float* final_obs;  // f32 [N, W, Fobs] or null: terminal observations (same-step mode)
  EnvRec* final_rec; // [N] or null: the record of env e as it was when its episode ended (same-step
                     // mode with final_obs: what `final_info` reports; gte_get_final_state)
  float* obs;
  float* reward;
  double* reward64;
  uint8_t *terminated, *truncated;
  int32_t* term_count;       // slot of the 2-slot counter this launch adds to
  int32_t* term_count_next;  // the other slot: cleared by this launch for the next one
  int32_t* term_ids;
  // --- inputs of this launch
  const int32_t* actions;                      // step: i32 [N] (device)
  const uint8_t* mask;                         // reset: u8 [N] or null
  const int32_t *inj_idx, *inj_pos, *inj_ds;   // reset: i32 [N] or null
  // --- queued draws for auto-resets
  int32_t q_n;
  const int32_t *q_idx, *q_pos, *q_ds;
  // --- geometry
  const int32_t* perm; // processing slot -> env id (L2-affinity order), or null = identity
  int32_t epw;         // environments per wavefront
  int32_t debug;       // gte_config.debug_flags (timing ablations)
  int32_t lean_rows;   // != 0: full waves of 16-byte-vector windows take the lean copy loop (gte_kernels.hip)
  int32_t hot_lds;     // != 0: a step's record stores go through LDS, one 64-byte request per env (gte_kernel)
  // --- trajectory row written by THIS launch's phase A (a gte_step with log_steps > 0; the
  // shared-TU step kernel only).  log.idx == null: none (the host appends it with gte_log_kernel)
  LogArrays log;
  int64_t log_row_base;  // index of env 0 in the row being written: (row % L) * N
};

// What the ISOLATED hot instantiations of the step kernel (gte_hot.hip / gte_hot_nt.hip, compiled
// with GTE_HOT_ONLY) leave OUT of phase A: the terminal records behind `final_info`
// (p.final_rec, environments.py:272 in same-step mode) and the trajectory row written by the
// kernel itself (p.log).  This is the ONE statement of that list: gte_step picks the kernel with
// it, the hot launchers refuse any launch it does not cover (hipErrorInvalidValue -> GTE_ERR_HIP),
// and every `#ifndef GTE_HOT_ONLY` block inside phase A names a field tested here.  Round 2
// shipped a launch predicate that had drifted from the compiled-out store: final_info read
// zero-filled records at every 16-byte-vector shape.
inline bool hot_tu_covers(const Params& p) { return p.final_rec == nullptr && p.log.rows == nullptr; }

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  One block of four draws per
// (global env id, episode, stream).  The reference draws from NumPy's global
// legacy RNG (environments.py:167,174,385); parity runs inject those draws.
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline void reset_draws(const Params& p, int32_t e, int32_t episode,
                                   uint32_t stream, uint32_t out[4]) {
  const int64_t gid = p.env_id_base + e;
  philox4x32_10((uint32_t)gid, (uint32_t)((uint64_t)gid >> 32), (uint32_t)episode, stream,
                (uint32_t)p.seed, (uint32_t)(p.seed >> 32), out);
}

__device__ inline int32_t bounded(uint32_t x, int32_t span) {
  return (int32_t)__umulhi(x, (uint32_t)span);
}

// k-th element of the keyed pseudo-random permutation of [0, D) for pick round
// `round` of env e (MultiDatasetTradingEnv.next_dataset, environments.py:380-388:
// uniform among the least-used datasets == every dataset once per round of D
// picks, in random order).
__device__ inline int32_t perm_pick(const Params& p, int32_t e, int32_t round, int32_t k) {
  const int32_t D = p.D;
  if (D == 1) return 0;
  int b = 32 - __clz(D - 1);
  if (b < 1) b = 1;
  const uint32_t mask = (b == 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  uint32_t r[4];
  reset_draws(p, e, round, 0x44534554u, r);
  uint32_t x = (uint32_t)k;
  const int sh = (b + 1) / 2;
  do {  // a bijection on b bits; cycle-walk into [0, D): terminates, < 2 rounds expected
    x = (x * (r[0] | 1u) + r[1]) & mask;
    x ^= x >> sh;
    x = (x * (r[2] | 1u) + r[3]) & mask;
    x ^= x >> sh;
    x = (x * 0x9E3779B1u + (r[0] >> 7)) & mask;
    x ^= x >> sh;
  } while (x >= (uint32_t)D);
  return (int32_t)x;
}

// ---------------------------------------------------------------------------
// Portfolio arithmetic, utils/portfolio.py

struct Portfolio {
  double asset, fiat, ia, ifi;
};

// Portfolio.valorisation, portfolio.py:7-13 — sum() of a 4-list from int 0, left to right
__device__ inline double valorisation(const Portfolio& q, double price) {
  double s = 0.0 + q.asset * price;
  s = s + q.fiat;
  s = s + (-q.ia * price);
  s = s + (-q.ifi);
  return s;
}

__device__ inline double pymax0(double x) { return x > 0.0 ? x : 0.0; }  // max(0, x)

// Portfolio.trade_to_position, portfolio.py:18-43
__device__ inline void trade_to_position(Portfolio& q, double position, double price,
                                         double fees) {
  const double cur = q.asset * price / valorisation(q, price);  // :20
  double ratio = 1.0;                                            // :21
  if (position <= 0.0 && cur < 0.0) {                            // :22-23
    const double r = position / cur;
    ratio = r < 1.0 ? r : 1.0;
  } else if (position >= 1.0 && cur > 1.0) {                     // :24-25
    const double r = (position - 1.0) / (cur - 1.0);
    ratio = r < 1.0 ? r : 1.0;
  }
  if (ratio < 1.0) {                                             // :26-30
    q.asset = q.asset - (1.0 - ratio) * q.ia;
    q.fiat = q.fiat - (1.0 - ratio) * q.ifi;
    q.ia = ratio * q.ia;
    q.ifi = ratio * q.ifi;
  }
  double trade = position * valorisation(q, price) / price - q.asset;  // :33
  if (trade > 0.0) {                                             // :34-38
    trade = trade / (1.0 - fees + fees * position);
    const double asset_fiat = -trade * price;
    q.asset = q.asset + trade * (1.0 - fees);
    q.fiat = q.fiat + asset_fiat;
  } else {                                                       // :39-43
    trade = trade / (1.0 - fees * position);
    const double asset_fiat = -trade * price;
    q.asset = q.asset + trade;
    q.fiat = q.fiat + asset_fiat * (1.0 - fees);
  }
}

__device__ inline double reward_of(const Params& p, double pv, double pv_prev) {
  const double lr = log(pv / pv_prev);  // basic_reward_function, environments.py:17-18
  if (p.reward_kind == GTE_REWARD_SCALED_LOG_RETURN) return p.rp0 * lr;
  if (p.reward_kind == GTE_REWARD_CLIPPED_LOG_RETURN) {
    const double v = p.rp0 * lr;  // np.clip(v, lo, hi)
    return v < p.rp1 ? p.rp1 : (v > p.rp2 ? p.rp2 : v);
  }
  return lr;
}

// What phase A (one lane per env) hands to the gather (whole waves copy the
// observation windows); published to LDS as a 16-byte JobRec + the current values.
struct ObsJob {
  const float* src;   // first row of the window in the dataset's feature table
  int32_t idx;        // current row
  int32_t slot0;      // slot of the window's first row in the env's dynamic store
  int32_t n_zero;     // leading window rows whose dynamic columns read as zero
  int32_t flags;      // bit0: copy the window; bit1: zero the env's dynamic store
  float cur[GTE_MAX_DYN];  // dynamic features of the current row (f32, :154)
};

// Same-step auto-reset with final_obs: the window of the TERMINAL state, gathered into
// final_obs[env] next to the reset observation.
struct FinalJob {
  const float* src;
  int32_t slot0, n_zero, flags;   // flags bit0: this env ended in this launch
  int32_t clob_slot;              // ring slot the reset's current row overwrote ...
  float clob[GTE_MAX_DYN];        // ... and what it held before
  float cur[GTE_MAX_DYN];         // dynamic features of the terminal row
};

}  // namespace gte
