// gte_kernels.hip — the step / reset kernels of libgte (gfx950 / CDNA4 only).
//
// One launch advances every environment of the shard by one step
// (TradingEnv.step, reference environments.py:233-272) or resets the masked
// ones (TradingEnv.reset, :163-199).  Workgroup = 256 threads = 4 wavefronts =
// 64 environments at windowed shapes; two kinds of work:
//
//   phase A  one lane per environment: the scalar fp64 state machine
//            (_take_action/_trade -> Portfolio.trade_to_position ->
//            limit-order fills -> update_interest -> valorisation ->
//            done/truncated -> reward), auto-reset with Philox or injected
//            draws, wave-level compaction of the terminal mask (__ballot +
//            popcount prefix, one atomic per wave).  The per-env state is one
//            128-byte record, reached through the L2-affinity permutation.
//   gather   _get_obs (:152-160): the window of env e is ONE contiguous block of
//            W*F_obs floats of the row-major feature table, moved with 16-byte
//            loads / non-temporal stores (1 KiB per wave instruction) and patched
//            in flight with the dynamic columns, which the wave first stages in
//            LDS with one coalesced pass over its envs' small dynamic stores.
//
// Kernels:  gte_kernel<MODE,...>     phase A, LDS barrier, gather: every step and reset
//                                    (the headline instantiation is compiled alone in
//                                    gte_hot.hip / gte_hot_nt.hip);
//           gte_rollout.hip          K steps in one launch: gte_rollout_resident_kernel (windows
//                                    resident in LDS), gte_rollout_state_kernel (no observations),
//                                    gte_rollout_kernel (gather per step; shapes LDS cannot hold);
//           gte_affinity_*           counting sort of the envs by table region
//                                    (processing order, speed only);
//           gte_add_orders / gte_extract_state / gte_rewind_queue: small helpers.
//
// The path is HBM-bound (no contraction, so no MFMA): >= 97 % of its bytes are the
// window gather + observation store.  See DESIGN.md for roofline and measurements.
#include "gte_device.h"

namespace gte {

typedef float float4_t __attribute__((ext_vector_type(4)));

enum { MODE_STEP = 0, MODE_RESET = 1 };

// Diagnostic build only (-DGTE_STAMPS, libgte_stamps.so, never shipped): lane 0 of a
// workgroup's wave 0 records s_memrealtime (100 MHz) at a few points of the step kernel into
// the buffer whose address the host passes in p.inj_ds (unused by a step).
#ifdef GTE_STAMPS
#define GTE_STAMP(k)                                                                          \
  do {                                                                                        \
    if (MODE == MODE_STEP && p.inj_ds) {                                                      \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); /* mark data ARRIVAL */      \
      if (threadIdx.x == 0)                                                                   \
        ((unsigned long long*)p.inj_ds)[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                         \
  } while (0)
// slot k = where the calling wave runs instead of a time: HW_ID (SIMD_ID bits 5:4, CU_ID 11:8,
// SH_ID 12, SE_ID 15:13) | XCC_ID << 32
#define GTE_STAMP_HWID(k)                                                                     \
  do {                                                                                        \
    if (MODE == MODE_STEP && p.inj_ds && threadIdx.x == 0)                                    \
      ((unsigned long long*)p.inj_ds)[blockIdx.x * 8 + (k)] =                                 \
          (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                     \
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);             \
  } while (0)
#else
#define GTE_STAMP(k) do {} while (0)
#define GTE_STAMP_HWID(k) do {} while (0)
#endif

// The part of an env's record a step works on, in registers.  The fields only a reset touches
// (episode, eps_on_ds, n_picks, q_head) stay in the record and are read / written there, inside
// the rare reset branches: carried through the fp64 state machine they cost the step kernel
// four more VGPRs, i.e. an occupancy step.
struct EnvRegs {
  int32_t idx, step, pos, dsi, start, needs_reset, lo_n;
  Portfolio q;
  double pv, realpos;
};

__device__ inline void load_state(const Params& p, int e, EnvRegs& s) {
  const EnvRec* r = &p.rec[e];  // 128-byte aligned record: six 16-byte loads
  const int4* ri = reinterpret_cast<const int4*>(r);
  const int4 a = ri[0];  // idx, step, pos, dsi
  const double2* rd = reinterpret_cast<const double2*>(&r->asset);  // offset 16
  const double2 d0 = rd[0], d1 = rd[1], d2 = rd[2];
  const int4 b = ri[4];  // start, (episode), needs_reset, (eps_on_ds)
  const int4 c = ri[5];  // (n_picks), (q_head), lo_n, pad
  s.idx = a.x; s.step = a.y; s.pos = a.z; s.dsi = a.w;
  s.start = b.x; s.needs_reset = b.z; s.lo_n = c.z;
  s.q.asset = d0.x; s.q.fiat = d0.y; s.q.ia = d1.x; s.q.ifi = d1.y;
  s.pv = d2.x; s.realpos = d2.y;
}

// The record's hot half (EnvRec).  start, lo_n and needs_reset are written where they change
// (do_reset, fill_limit_orders, the episode end in phase_a).
__device__ inline void store_state_at(EnvRec* r, const EnvRegs& s) {
  *reinterpret_cast<int4*>(&r->idx) = make_int4(s.idx, s.step, s.pos, s.dsi);
  double2* d = reinterpret_cast<double2*>(&r->asset);
  d[0] = make_double2(s.q.asset, s.q.fiat);
  d[1] = make_double2(s.q.ia, s.q.ifi);
  d[2] = make_double2(s.pv, s.realpos);
}
// The same 64 bytes into the workgroup's LDS image (slot = env of the workgroup): the gather waves
// write them out, four lanes per env (flush_hot_records).  The pointer keeps its address space in its
// type (as a generic pointer these would be flat stores).
typedef int __attribute__((ext_vector_type(4))) int4_t;
typedef double __attribute__((ext_vector_type(2))) double2_t;
typedef unsigned char __attribute__((address_space(3))) * lds_byte_ptr;
__device__ inline void store_state_lds(lds_byte_ptr h, const EnvRegs& s) {
  typedef int4_t __attribute__((address_space(3))) * li4;
  typedef double2_t __attribute__((address_space(3))) * ld2;
  int4_t a = {s.idx, s.step, s.pos, s.dsi};
  double2_t d0 = {s.q.asset, s.q.fiat}, d1 = {s.q.ia, s.q.ifi}, d2 = {s.pv, s.realpos};
  *(li4)h = a;
  *(ld2)(h + 16) = d0;
  *(ld2)(h + 32) = d1;
  *(ld2)(h + 48) = d2;
}
__device__ inline void store_state(const Params& p, int e, const EnvRegs& s) { store_state_at(&p.rec[e], s); }

// MultiDatasetTradingEnv.next_dataset, environments.py:380-391
__device__ inline void next_dataset(const Params& p, int e, int32_t inj_ds, EnvRegs& s,
                                    bool& fresh) {
  EnvRec* r = &p.rec[e];
  const int32_t n = r->n_picks;
  r->n_picks = n + 1;
  s.dsi = (inj_ds >= 0) ? inj_ds : perm_pick(p, e, n / p.D, n % p.D);
  r->eps_on_ds = 0;                 // :381
  if (p.persist) fresh = true;      // _set_df rebuilds _obs_array (:135-141)
}

// TradingEnv.reset, environments.py:163-199 (+ MultiDataset reset :393-400)
__device__ inline void do_reset(const Params& p, int e, int32_t inj_idx, int32_t inj_pos,
                                int32_t inj_ds, EnvRegs& s, bool& fresh) {
  EnvRec* rec = &p.rec[e];
  if (p.D > 1) {  // :394-398
    const int32_t eps = rec->eps_on_ds + 1;
    rec->eps_on_ds = eps;  // (next_dataset, if it runs, clears it afterwards)
    if (eps % p.switch_every == 0) next_dataset(p, e, inj_ds, s, fresh);
  }
  uint32_t r[4];
  const int32_t episode = rec->episode;
  reset_draws(p, e, episode, 0x52534554u, r);
  rec->episode = episode + 1;
  s.step = 0;  // :166
  s.lo_n = 0;  // :168 self._limit_orders = {}
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
  rec->start = idx;
  rec->lo_n = 0;
  const double position = p.positions[pi];  // TargetPortfolio, portfolio.py:59-66
  const double price = d.close[idx];
  s.q.asset = position * p.V0 / price;
  s.q.fiat = (1.0 - position) * p.V0;
  s.q.ia = 0.0;
  s.q.ifi = 0.0;
  s.pv = p.V0;          // :194
  s.realpos = position; // :192
  s.needs_reset = 0;
  rec->needs_reset = 0;
}

// TradingEnv._take_action_order_limit, environments.py:217-223: every pending order
// whose target differs from the current position and whose limit lies inside
// [low, high] of the NEW row trades at the limit price, in insertion order.  A filled
// non-persistent order is removed (the reference deletes it while iterating its dict
// and raises RuntimeError; the intended behaviour is implemented).
__device__ inline void fill_limit_orders(const Params& p, int e, const DatasetDesc* d,
                                         EnvRegs& s) {
  const int n = s.lo_n;
  if (n <= 0) return;
  int32_t* lp = p.lo_pos + (int64_t)e * p.P;
  double* ll = p.lo_limit + (int64_t)e * p.P;
  uint8_t* lper = p.lo_persist + (int64_t)e * p.P;
  const double hi = d->high[s.idx], lo = d->low[s.idx];
  int k = 0;
  for (int j = 0; j < n; ++j) {
    const int32_t pi = lp[j];
    const double limit = ll[j];
    const uint8_t per = lper[j];
    bool keep = true;
    const double position = p.positions[pi];
    if (position != p.positions[s.pos] && limit <= hi && limit >= lo) {
      trade_to_position(s.q, position, limit, p.fees);  // _trade(position, price=limit)
      s.pos = pi;
      if (!per) keep = false;
    }
    if (keep) {
      if (k != j) { lp[k] = pi; ll[k] = limit; lper[k] = per; }
      ++k;
    }
  }
  s.lo_n = k;
  p.rec[e].lo_n = k;
}

#ifndef GTE_HOT_ONLY
// TradingEnv.add_limit_order, environments.py:227-231: `orders[position] = {...}` — an
// existing key (a position VALUE) keeps its place in the iteration order, a new one
// goes last.  One thread per env.
__global__ void gte_add_orders_kernel(const Params p, const int32_t* pos_index,
                                      const double* limit, const uint8_t* persistent) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const int32_t pi = pos_index[e];
  if (pi < 0) return;
  int32_t* lp = p.lo_pos + (int64_t)e * p.P;
  const int n = p.rec[e].lo_n;
  int j = 0;
  while (j < n && p.positions[lp[j]] != p.positions[pi]) ++j;
  if (j == n) p.rec[e].lo_n = n + 1;  // n < P: at most one order per distinct position value
  lp[j] = pi;
  p.lo_limit[(int64_t)e * p.P + j] = limit[e];
  p.lo_persist[(int64_t)e * p.P + j] = persistent ? persistent[e] : 0;
}

#endif  // GTE_HOT_ONLY

__device__ inline void pop_injection(const Params& p, int e, EnvRegs& s, int32_t& qi,
                                     int32_t& qp, int32_t& qd) {
  qi = qp = qd = -1;
  if (p.q_n <= 0) return;
  const int32_t h = p.rec[e].q_head;
  if (h >= p.q_n) return;
  p.rec[e].q_head = h + 1;
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

// Prices a fused rollout carries from step to step (one lane = one env): a step trades at
// close[idx] — the price the previous step valued the portfolio at — and values at close[idx+1],
// which the previous step already asked for; the load that would head every step's dependency
// chain is issued a step early instead.  Invalid (idx < 0) after anything but a plain step.
// What a step returned for one env, for a caller that also writes the trajectory row.
struct StepOut {
  double reward, pv, realpos, asset, fiat, ia, ifi;  // reward of the step; the rest: state after it
  int32_t idx, step, pos, dsi;
  int32_t flags;  // bit0 terminated, bit1 truncated
};

struct PriceCarry {
  double cur, next;  // close[idx], close[idx + 1] of dataset dsi
  int32_t idx, dsi;
};

// compact: add the envs whose episode ended to the terminal list (off for the inner steps
// of a fused rollout, which keeps per-step flags instead); pv_out: the valuation after the step.
// carried: the env's registers live across calls (the fused rollout keeps them there for all K
// steps: no record load per step; the record is still written through); action_in: the action
// was loaded ahead of time.  Both are nullptr — and fold away — in the per-step kernels.
template <int MODE>
__device__ inline void phase_a(const Params& p, int e, bool active, int lane, ObsJob& job,
                               FinalJob* fin = nullptr, bool compact = true,
                               double* pv_out = nullptr, EnvRegs* carried = nullptr,
                               const int32_t* action_in = nullptr, bool write_record = true,
                               PriceCarry* pc = nullptr, StepOut* so = nullptr,
                               lds_byte_ptr hot = nullptr) {
  // write_record = false (fused rollouts, with `carried`): the record is not written through on
  // this step — the caller stores it once, after its last step (fields a reset or a limit-order
  // fill changes are written where they change, whatever this flag says)
  if (fin) fin->flags = 0;
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
      if (p.D > 1 && p.rec[e].n_picks == 0) next_dataset(p, e, id, s, fresh);  // ctor pick, :378
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
    EnvRegs s_own;
    EnvRegs& s = carried ? *carried : s_own;
    if (!carried) load_state(p, e, s);
    int32_t action = action_in ? *action_in : p.actions[e];
    GTE_STAMP(2);  // record + action arrived
    // positions[position_index] raises IndexError in the reference (:234); a device-side
    // action cannot raise, so an out-of-range index is treated as None (hold), never read
    if (action >= p.P) action = -1;
    bool fresh = false;
    bool stepped = true;
    if (s.needs_reset) {
      if (p.autoreset == GTE_AUTORESET_NEXT_STEP) {
        int32_t qi, qp, qd;
        pop_injection(p, e, s, qi, qp, qd);
        do_reset(p, e, qi, qp, qd, s, fresh);
        if (pc) pc->idx = -1;
        p.reward[e] = 0.0f; p.reward64[e] = 0.0;
        p.terminated[e] = 0; p.truncated[e] = 0;
        if (so) { so->reward = 0.0; so->flags = 0; }
        stepped = false;
      } else if (s.idx >= (int32_t)p.ds[s.dsi].T - 1) {
        // no auto-reset and no row left: the reference raises IndexError (:239);
        // the batch leaves such an env frozen, flags still raised
        p.reward[e] = 0.0f; p.reward64[e] = 0.0;
        // its flags stay raised (rewritten, not relied upon: a rollout writes every step's
        // flags to a fresh row): the valuation has not moved since the 0.7 test (:246), and
        // being on the last row is the truncation rule itself (:248)
        p.terminated[e] = (s.pv / p.V0) <= 0.7 ? 1 : 0;
        p.truncated[e] = 1;
        if (so) { so->reward = 0.0; so->flags = ((s.pv / p.V0) <= 0.7 ? 1 : 0) | 2; }
        stepped = false;
        ended = true;  // so it stays in the terminal list
      }
    }
    if (stepped) {
      // only the fields this path needs (the whole 40-byte descriptor held in registers
      // across the fp64 state machine costs the kernel an occupancy step)
      const DatasetDesc* dp = p.ds + s.dsi;
      const double* d_close = dp->close;
      const int32_t d_T = (int32_t)dp->T;
      const bool carried_prices = pc && pc->idx == s.idx && pc->dsi == s.dsi;
      if (action >= 0) {  // :234 -> :213-215: trade only when the position VALUE differs
        const double position = p.positions[action];
        if (position != p.positions[s.pos]) {
          trade_to_position(s.q, position, carried_prices ? pc->cur : d_close[s.idx], p.fees);  // :204-209
          s.pos = action;                                            // :210
        }
      }
      s.idx += 1;   // :235
      s.step += 1;  // :236
      if (p.lo_pos) fill_limit_orders(p, e, dp, s);  // :238
      const double price = carried_prices ? pc->next : d_close[s.idx];  // :239
      if (pc) {  // this step's valuation price is the next step's trade price; ask for the one after
        pc->cur = price;
        pc->idx = s.idx;
        pc->dsi = s.dsi;
        pc->next = (s.idx + 1 < d_T) ? d_close[s.idx + 1] : price;
      }
      GTE_STAMP(3);  // descriptor, positions, trade, price at the new row arrived
      s.q.ia = pymax0(-s.q.asset) * p.rate;   // update_interest, portfolio.py:44-46
      s.q.ifi = pymax0(-s.q.fiat) * p.rate;
      const double pv = valorisation(s.q, price);  // :241
      const bool done = (pv / p.V0) <= 0.7;        // :246
      bool trunc = s.idx >= d_T - 1;               // :248
      if (p.max_dur > 0 && s.step >= p.max_dur - 1) trunc = true;  // :250
      s.realpos = (s.q.asset - s.q.ia) * price / valorisation(s.q, price);  // :259
      double rew = 0.0;                            // :263, stays 0 when done (:265)
      if (!done) rew = reward_of(p, pv, s.pv);
      s.pv = pv;
      p.reward64[e] = rew;
      p.reward[e] = (float)rew;
      p.terminated[e] = done ? 1 : 0;
      p.truncated[e] = trunc ? 1 : 0;
      if (so) { so->reward = rew; so->flags = (done ? 1 : 0) | (trunc ? 2 : 0); }
      ended = done || trunc;
      if (ended) { s.needs_reset = 1; p.rec[e].needs_reset = 1; }
      if (ended && p.autoreset == GTE_AUTORESET_SAME_STEP) {
        // the reference's step() runs _get_obs (:272) before any wrapper resets the env:
        // write the terminal row's dynamic features, remember the terminal window
#ifndef GTE_HOT_ONLY  // p.final_rec: hot_tu_covers() keeps such launches off the isolated TUs
        if (p.final_rec) {  // what the wrapper's `final_info` reports (state before the reset)
          store_state_at(&p.final_rec[e], s);
          p.final_rec[e].needs_reset = s.needs_reset;
          p.final_rec[e].start = s.start;
        }
#endif
        ObsJob term;
        make_job(p, e, s, false, term);
        int32_t qi, qp, qd;
        pop_injection(p, e, s, qi, qp, qd);
        do_reset(p, e, qi, qp, qd, s, fresh);
        if (pc) pc->idx = -1;
        if (fin && p.final_obs) {
          fin->src = term.src; fin->slot0 = term.slot0; fin->n_zero = term.n_zero; fin->flags = 1;
          // the reset's current row is about to overwrite one ring slot the terminal window
          // may still need: keep its old content (this lane wrote/reads it in program order)
          const int64_t cs = p.persist ? (int64_t)s.idx : (int64_t)(s.idx % p.W);
          fin->clob_slot = (int32_t)cs;
#pragma unroll
          for (int i = 0; i < GTE_MAX_DYN; ++i) {
            fin->cur[i] = term.cur[i];
            fin->clob[i] = (i < p.nd) ? p.ring[((int64_t)e * p.depth + cs) * p.nd + i] : 0.0f;
          }
        }
      }
    }
    GTE_STAMP(4);  // state machine done, outputs issued
    if (pv_out) *pv_out = s.pv;
    if (write_record) {
      if (hot) store_state_lds(hot, s);  // (written out by the gather waves: flush_hot_records)
      else store_state(p, e, s);
    }
    if (so) {
      so->idx = s.idx; so->step = s.step; so->pos = s.pos; so->dsi = s.dsi;
      so->pv = s.pv; so->realpos = s.realpos;
      so->asset = s.q.asset; so->fiat = s.q.fiat; so->ia = s.q.ia; so->ifi = s.q.ifi;
    }
    make_job(p, e, s, fresh, job);
    GTE_STAMP(5);  // record, ring and job stores done
  }

  // terminal-mask compaction: one atomic per wave, ids in lane order within a wave
  const unsigned long long m = __ballot(ended);
  if (compact && m != 0ull) {
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

// Observation store policy (gte_config.nontemporal_obs): 0 plain, 1 non-temporal,
// 2 sc1.  tools/store_bench.hip on MI355X, 168 MB store-only: plain 26 us, nt 33 us,
// sc1 23-24 us; plain stores evict the feature table from L2, nt/sc1 do not (an sc1
// store drops the line from L2).  sc1 needs inline asm: the compiler does not count it
// on vmcnt, so every place that relies on "my stores are done" waits explicitly.
template <int NT, typename T>
__device__ inline void store_out(T* dst, const T& v) {
  if constexpr (NT == 2 && sizeof(T) == 16) {
    // no "memory" clobber: nothing in the kernel reads obs back, and volatile asms keep
    // their order among themselves (the s_waitcnt before the barrier stays behind them)
    // s_nop 1: a VMEM store of more than 64 bits reads its data registers for a couple of cycles
    // after it issues; a VALU write to them in that window corrupts the stored value (gfx9 / CDNA
    // hazard "VMEM store > 8 bytes followed by a write of the VGPRs holding the data").  hipcc pads
    // its own stores, but it does not look inside inline asm: without the two wait states here the
    // lean copy loop stored the next address computation's low words in place of x, y.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v));
  } else if constexpr (NT == 2) {
    const float f = __builtin_bit_cast(float, v);  // 1-element vector: plain VGPR operand
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(f));
  } else if constexpr (NT == 1) {
    __builtin_nontemporal_store(v, dst);
  } else {
    *dst = v;
  }
}

// The window's source pointer travels through LDS as a 64-bit integer, which makes the
// compiler forget that it points to global memory: it then emits flat_load, and flat
// loads count on vmcnt AND lgkmcnt (every LDS read in the loop waits for them).  Cast
// back to the global address space explicitly.
template <typename T>
__device__ inline T load_global(uint64_t base, int64_t index) {
  typedef const T __attribute__((address_space(1))) * gptr_t;
  return ((gptr_t)base)[index];
}

// Put nd dynamic values x[0..nd) into vector v, which is the LAST vector of a window row.
// With 16-byte vectors F_obs % 4 == 0 and nd <= 4, so the dynamic columns are exactly the
// last nd components of that vector: a wave-uniform switch, no per-component compares.
__device__ inline void set_tail(float __attribute__((ext_vector_type(4))) & v, int nd,
                                const float x[GTE_MAX_DYN]) {
  switch (nd) {  // wave-uniform
    case 1: v[3] = x[0]; break;
    case 2: v[2] = x[0]; v[3] = x[1]; break;
    case 3: v[1] = x[0]; v[2] = x[1]; v[3] = x[2]; break;
    default: v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; break;
  }
}

// LDS image of a workgroup: the jobs phase A hands to phase B (one per env of the
// workgroup) and, when STAGE, the dynamic-column values of every window row.
// One 16-byte job record per env of the workgroup: a single ds_read_b128 per vector in
// the copy loop (five separate LDS reads measurably throttled the loop).
struct alignas(16) JobRec {
  uint64_t src;   // first row of the window in the feature table
  int32_t env;    // env id processed in this slot (perm[slot], or the slot itself); -1 = none
  uint32_t meta;  // bit0 copy the window, bit1 zero the env's dynamic store,
                  // bits 2..16 n_zero (W < 32768, gte_create checks), bits 17..31 slot0 of the
                  // W-deep ring (meaningless with dyn_persist: dyn_value uses the row itself)
};
__device__ inline uint32_t pack_meta(int flags, int n_zero, int slot0) {
  return (uint32_t)(flags & 3) | ((uint32_t)n_zero << 2) | ((uint32_t)slot0 << 17);
}
__device__ inline int meta_flags(uint32_t m) { return (int)(m & 3u); }
__device__ inline int meta_n_zero(uint32_t m) { return (int)((m >> 2) & 0x7FFFu); }
__device__ inline int meta_slot0(uint32_t m) { return (int)(m >> 17); }

// LDS image of a workgroup: the jobs phase A hands to phase B (one per env of the
// workgroup) and, when staged, the dynamic-column values of every window row.
struct WgLds {
  JobRec* job;      // [EPB]
  unsigned char* hot;  // [EPB][64] the records' hot halves as phase A leaves them (flush_hot_records)
  int32_t* idx;     // [EPB] current row (persist mode's zero-fill needs it)
  float* cur;       // [EPB][GTE_MAX_DYN] dynamic features of the current row
  FinalJob* fin;    // [EPB] terminal windows (only when p.final_obs)
  unsigned char* logrow;  // [EPB][sizeof(LogRow)] the step's trajectory rows (only when p.log.rows; flush_log_rows)
  float* staged;    // [EPB][W][nd]: the raw rings; the lean copy loop resolves a wave's part IN
                    // PLACE into window order (rotation / zero rows / current row applied)
};

__device__ inline WgLds carve_lds(unsigned char* base, int EPB, bool with_final, bool with_log) {
  WgLds L;
  L.job = (JobRec*)base;                   base += 16 * EPB;
  L.hot = base;                            base += 64 * EPB;
  L.cur = (float*)base;                    base += 4 * GTE_MAX_DYN * EPB;
  L.idx = (int32_t*)base;                  base += 4 * EPB;
  L.logrow = base;                         base += with_log ? sizeof(LogRow) * EPB : 0;  // (16-byte aligned here)
  L.fin = (FinalJob*)base;                 base += with_final ? sizeof(FinalJob) * EPB : 0;
  L.staged = (float*)base;
  return L;
}

// The records' hot halves, from the LDS image phase A left (store_state_lds) to the records: four
// lanes per env, i.e. one 64-byte request per env where the lane that stepped the env issued four
// 16-byte stores to 64 different lines each.
__device__ inline void flush_hot_records(const Params& p, const WgLds& L, int s_first, int n_env, int lane) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  for (int i = lane; i < n_env * 4; i += 64) {  // (one pass with 16 envs per wave)
    const int sl = s_first + (i >> 2), part = i & 3;
    const int env = L.job[sl].env;  // (>= 0 for the first n_env slots)
    const f4 v = *reinterpret_cast<const f4*>(L.hot + 64 * sl + 16 * part);
    *(reinterpret_cast<f4*>(&p.rec[env]) + part) = v;
  }
}

// The step's trajectory rows, from the LDS image phase A's lanes left to the log: five lanes per env
// (80 contiguous bytes, two requests per env where the twelve columns were twelve).
__device__ inline void flush_log_rows(const Params& p, const WgLds& L, int s_first, int n_env, int lane) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  for (int i = lane; i < n_env * 5; i += 64) {
    const int q = i / 5, part = i - q * 5;
    const int sl = s_first + q;
    const int env = L.job[sl].env;  // (>= 0 for the first n_env slots)
    const f4 v = *reinterpret_cast<const f4*>(L.logrow + sizeof(LogRow) * sl + 16 * part);
    *(reinterpret_cast<f4*>(&p.log.rows[p.log_row_base + env]) + part) = v;
  }
}

// phase A's lane publishes its env's job (the env id was written at kernel start)
__device__ inline void publish_job(const WgLds& L, int slot, const ObsJob& job) {
  L.job[slot].src = (uint64_t)job.src;
  L.job[slot].meta = pack_meta(job.flags, job.n_zero, job.slot0);
  L.idx[slot] = job.idx;
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) L.cur[slot * GTE_MAX_DYN + i] = job.cur[i];
}

// value of dynamic feature i in window row w of the env in LDS slot `s` (generic form:
// reads the env's store in global memory; used by the persist-mode staging and when
// nothing is staged)
__device__ inline float dyn_value(const Params& p, const WgLds& L, int s, const float* ring_e,
                                  int w, int i) {
  if (w == p.W - 1) return L.cur[s * GTE_MAX_DYN + i];  // current row: from phase A
  const uint32_t m = L.job[s].meta;
  if (w < meta_n_zero(m)) return 0.0f;                   // never written: reads as zero
  int32_t slot;
  if (p.persist) {
    // T-deep column: the slot IS the table row, which does not fit JobRec.meta's 15 bits
    // (rows >= 32768 used to alias): take it from the current row published next to the job
    slot = L.idx[s] - p.W + 1 + w;
  } else {
    slot = meta_slot0(m) + w;
    if (slot >= p.W) slot -= p.W;
  }
  return ring_e[(int64_t)slot * p.nd + i];
}

// The wave gathers, once and coalesced, the dynamic-column values of all window rows
// of its envs into LDS (W*nd floats per env: 160 B at the headline shape), so that the
// copy loop patches from LDS instead of issuing divergent global loads per vector.
__device__ inline void stage_dynamic(const Params& p, const WgLds& L, int s_first,
                                     int n_env, int lane, uint64_t wnd_magic) {
  const uint32_t WND = (uint32_t)(p.W * p.nd);
  const uint32_t total = (uint32_t)n_env * WND;
  for (uint32_t k = (uint32_t)lane; k < total; k += 64u) {
    const uint32_t el = fastdiv40(k, wnd_magic);
    const uint32_t r = k - el * WND;
    const uint32_t w = r / (uint32_t)p.nd;
    const int i = (int)(r - w * (uint32_t)p.nd);
    const int s = s_first + (int)el;
    const float* ring_e = p.ring + (int64_t)L.job[s].env * p.depth * p.nd;
    L.staged[(uint32_t)s * WND + r] = dyn_value(p, L, s, ring_e, (int)w, i);
  }
}

// STAGE_RAW: at kernel start every wave copies the W-deep rings of its envs — one
// contiguous block of EPW*W*nd floats — into LDS (coalesced, and its latency hides
// behind phase A); the slot rotation / zero rows / current row are resolved when a
// vector is patched.  STAGE_LATE (dyn_persist: the rows sit at idx-dependent offsets
// of a T-deep column): gathered after phase A, already resolved (stage_dynamic).
enum { STAGE_NONE = 0, STAGE_RAW = 1, STAGE_LATE = 2 };

__device__ inline void stage_raw_rings(const Params& p, const WgLds& L, int s_first, int n_env,
                                       int lane, uint64_t wnd_magic) {
  const uint32_t WND = (uint32_t)(p.W * p.nd);
  const uint32_t total = (uint32_t)n_env * WND;
  float* dst = L.staged + (uint32_t)s_first * WND;
  for (uint32_t k = (uint32_t)lane; k < total; k += 64u) {
    const uint32_t el = fastdiv40(k, wnd_magic);
    const uint32_t r = k - el * WND;
    dst[k] = p.ring[(int64_t)L.job[s_first + (int)el].env * WND + r];  // depth == W here
  }
}

// Overwrite the dynamic columns that vector `v` (columns col .. col+VEC-1 of window
// row w of the env in LDS slot s, job meta m) covers.  Straight-line code under ONE
// branch: nested divergent branches here cost ~250 instructions and a dozen
// s_waitcnt per vector.  Every index into v is a compile-time constant after
// unrolling: a run-time index would put v in scratch memory (measured: one scratch
// store per observation store, 2x WRITE_SIZE).
template <int VEC, int STAGE, typename vec_t>
__device__ inline void patch_dynamic(const Params& p, const WgLds& L, vec_t& v, int s, uint32_t m,
                                     const float* ring_e, int w, int col) {
  if (col + VEC <= p.Fs || (p.debug & 2)) return;  // all static columns
  const bool is_cur = (w == p.W - 1);
  int32_t slot = meta_slot0(m) + w;
  if (STAGE == STAGE_RAW) slot -= (slot >= p.W) ? p.W : 0;
  const bool zero = !is_cur && w < meta_n_zero(m);
  float x[GTE_MAX_DYN];
#pragma unroll
  for (int i = 0; i < GTE_MAX_DYN; ++i) {
    x[i] = 0.0f;
    if (i < p.nd) {  // wave-uniform
      if (STAGE == STAGE_RAW) {        // raw rings in LDS: pick the address, one LDS read
        const float* a = is_cur ? &L.cur[s * GTE_MAX_DYN + i] : &L.staged[(s * p.W + slot) * p.nd + i];
        x[i] = zero ? 0.0f : *a;
      } else if (STAGE == STAGE_LATE) {  // already resolved per window row
        x[i] = L.staged[(s * p.W + w) * p.nd + i];
      } else {
        x[i] = dyn_value(p, L, s, ring_e, w, i);
      }
    }
  }
  if constexpr (VEC == 4) {
    set_tail(v, p.nd, x);  // col + 4 > Fs  <=>  this is the row's last vector
  } else {
    const int i = col - p.Fs;  // VEC == 1: this element is dynamic feature i
    float r = x[0];
#pragma unroll
    for (int k = 1; k < GTE_MAX_DYN; ++k) r = (i == k) ? x[k] : r;
    v = r;
  }
}

// Window gather.  The wave's n_env*VPE vectors form one index space; lane l of
// iteration t handles vector t*64+l, so every wave instruction loads/stores 64*VEC*4
// contiguous, fully used bytes whatever the window size (also when an env's window is
// smaller than one wave instruction, e.g. windows=None).  The env differs per lane:
// its job is read from LDS.  U independent loads are in flight per lane.
// [k_lo, k_hi): the part of the index space to copy (a chunk claimed by the rollout kernel;
// everything by default).
template <int VEC, int NT, int STAGE, int U>
__device__ inline void phase_b(const Params& p, const WgLds& L, int s_first,
                               int n_env, int lane, uint64_t vpe_magic, uint64_t fv_magic,
                               uint32_t k_lo = 0u, uint32_t k_hi = 0xFFFFFFFFu) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  const uint32_t VPE = V / VEC;                // vectors per env
  const uint32_t FV = (uint32_t)p.Fobs / VEC;  // vectors per row
  const uint32_t total = min((uint32_t)n_env * VPE, k_hi);

  for (uint32_t k0 = k_lo; k0 < total; k0 += 64u * U) {
    vec_t v[U];
    uint32_t jj[U], ee[U], mm[U];
    int32_t env[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t k = k0 + (uint32_t)u * 64u + (uint32_t)lane;
      const bool in = k < total;
      const uint32_t kk = in ? k : 0u;
      ee[u] = fastdiv40(kk, vpe_magic);
      jj[u] = kk - ee[u] * VPE;
      const JobRec j = L.job[s_first + (int)ee[u]];  // one ds_read_b128
      mm[u] = j.meta;
      env[u] = j.env;
      ok[u] = in && (j.meta & 1u);
      if (ok[u]) { if (p.debug & 8) v[u] = (vec_t)(float)jj[u]; else v[u] = load_global<vec_t>(j.src, (int64_t)jj[u]); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const uint32_t w = fastdiv40(jj[u], fv_magic);
      const int col = (int)(jj[u] - w * FV) * VEC;
      const int s = s_first + (int)ee[u];
      const float* ring_e = p.ring + (int64_t)env[u] * p.depth * p.nd;
      patch_dynamic<VEC, STAGE>(p, L, v[u], s, mm[u], ring_e, (int)w, col);
      store_out<NT>((vec_t*)(p.obs + (int64_t)env[u] * V + (int64_t)jj[u] * VEC), v[u]);
    }
  }
}

// ---------------------------------------------------------------------------
// Lean copy loop (round 3).  The SQ counters show the step kernel's SIMDs ~75 % issue-busy: the
// generic loop above spends ~67 VALU instructions per wave-vector (two 40-bit magic divisions,
// 64-bit address arithmetic, per-vector branches, the dynamic-column patch executed by all 64 lanes
// because every wave instruction contains some row's last vector), i.e. the copy is bound by
// instruction issue as much as by memory.  For the common case — a full wave of envs that all copy,
// whole number of passes — this version does the same copy with a third of the instructions:
//   * the dynamic values of every window row are resolved ONCE per env into LDS (rotation, zero
//     rows, current row; already placed in the vector components they occupy), so patching a
//     vector is one ds_read_b128 and a per-lane select, no branch;
//   * a lane walks its vectors k = lane + 64 q with running (env, vector-in-env, row, vector-in-row)
//     counters instead of dividing;
//   * no per-vector validity branches (the caller checks the whole wave once).
// Results are the generic loop's, bit for bit (the parity suite runs through it).
constexpr int LEAN_MAX_ROWS = 512;  // window rows of one wave's envs the in-place resolve holds in registers

template <int ND>
__device__ inline void resolve_dynamic_rows(const Params& p, const WgLds& L, int s_first, int n_env,
                                            int lane, uint64_t wnd_magic) {
  const uint32_t W = (uint32_t)p.W;
  const uint32_t rows = (uint32_t)n_env * W;  // <= LEAN_MAX_ROWS (the caller checks)
  float x[LEAN_MAX_ROWS / 64][ND];
  // every value is read (from the raw rings / the current-row values) before any is written back:
  // a wave's envs own a contiguous part of L.staged that no other wave touches
#pragma unroll
  for (int q = 0; q < LEAN_MAX_ROWS / 64; ++q) {
    if (64u * (uint32_t)q >= rows) break;  // wave-uniform
    const uint32_t r = (uint32_t)lane + 64u * (uint32_t)q;
    const bool in = r < rows;
    const uint32_t rr = in ? r : 0u;
    const uint32_t el = fastdiv40(rr * (uint32_t)ND, wnd_magic);  // rr / W  (wnd_magic divides by W * nd)
    const uint32_t w = rr - el * W;
    const int s = s_first + (int)el;
    const uint32_t m = L.job[s].meta;
    int32_t slot = meta_slot0(m) + (int32_t)w;
    if (slot >= (int32_t)W) slot -= (int32_t)W;
    const bool is_cur = (w == W - 1u);
    const bool zero = !is_cur && (int)w < meta_n_zero(m);
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const float* a = is_cur ? &L.cur[s * GTE_MAX_DYN + i] : &L.staged[((uint32_t)s * W + (uint32_t)slot) * ND + i];
      x[q][i] = (zero || !in) ? 0.0f : *a;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int q = 0; q < LEAN_MAX_ROWS / 64; ++q) {
    const uint32_t r = (uint32_t)lane + 64u * (uint32_t)q;
    if (r < rows) {  // row r of the wave's part, in window order: env r / W, row r % W
#pragma unroll
      for (int i = 0; i < ND; ++i) L.staged[((uint32_t)s_first * W + r) * ND + i] = x[q][i];
    }
  }
}

#ifndef GTE_LEAN_U
#define GTE_LEAN_U 4  // vectors in flight per lane in the lean copy loop
#endif
template <int NT, int ND>
__device__ inline void phase_b_lean(const Params& p, const WgLds& L, int s_first, int n_env, int lane) {
  constexpr int U = GTE_LEAN_U;
  const uint32_t W = (uint32_t)p.W, FV = (uint32_t)p.Fobs / 4u, VPE = W * FV;
  const uint32_t total = (uint32_t)n_env * VPE;          // a multiple of 64 * U (checked by the caller)
  const uint32_t VB = VPE * 16u;                          // bytes per observation
  // running position of this lane's vector: env slot `ee`, vector in env `jj`, row `w`, vector in row `r`
  uint32_t ee = (uint32_t)lane / VPE;                     // VPE >= 64: 0
  uint32_t jj = (uint32_t)lane - ee * VPE;
  uint32_t w = jj / FV, r = jj - w * FV;
  const uint32_t w_inc = 64u / FV, r_inc = 64u - w_inc * FV;  // one step of 64 vectors
  char* const obs = (char*)p.obs;
  // (one fixed pass shape, the loop written out: as a generic lambda with a tail pass for other
  // multiples of 64 the same code compiled 5 % slower at config 5)
  for (uint32_t k0 = 0u; k0 < total; k0 += 64u * U) {
    float4_t v[U];
    float t[U][ND];
    uint32_t jj16[U];
    int32_t env[U];
    bool last[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s_first + (int)ee;
      const JobRec j = L.job[s];                           // one ds_read_b128
#pragma unroll
      for (int c = 0; c < ND; ++c) t[u][c] = L.staged[((uint32_t)s * W + w) * ND + c];  // one LDS read
      env[u] = j.env;
      jj16[u] = jj * 16u;
      last[u] = (r == FV - 1u);
      v[u] = load_global<float4_t>(j.src, (int64_t)jj);
      // advance by 64 vectors
      jj += 64u; w += w_inc; r += r_inc;
      if (r >= FV) { r -= FV; w += 1u; }
      if (jj >= VPE) { jj -= VPE; ee += 1u; w -= W; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4_t o = v[u];
      // the dynamic columns are the last ND components of a row's last vector
#pragma unroll
      for (int c = 0; c < ND; ++c) o[4 - ND + c] = last[u] ? t[u][c] : o[4 - ND + c];
      store_out<NT>((float4_t*)(obs + (uint64_t)(uint32_t)env[u] * VB + jj16[u]), o);
    }
  }
}

// zero the dynamic store of envs that switched dataset in persist mode (the
// reference rebuilds _obs_array in _set_df), except the current row's slot
__device__ inline void zero_fresh_stores(const Params& p, const WgLds& L,
                                         int s_first, int n_env, int lane) {
  for (int el = 0; el < n_env; ++el) {
    const int s = s_first + el;
    if (!(L.job[s].meta & 2u)) continue;
    const int idx = L.idx[s];
    float* ring_e = p.ring + (int64_t)L.job[s].env * p.depth * p.nd;
    const int64_t n = p.depth * p.nd;
    const int64_t keep_lo = (int64_t)idx * p.nd, keep_hi = keep_lo + p.nd;
    for (int64_t k = lane; k < n; k += 64)
      if (k < keep_lo || k >= keep_hi) ring_e[k] = 0.0f;
  }
}

// Terminal observations (same-step auto-reset + final_obs): the wave copies the terminal
// window of each of its envs that ended in this launch into final_obs[env].  Rare, so one
// env at a time.  Earlier rows' dynamic values come from the env's ring in global memory,
// except the terminal row itself (fin.cur) and the slot the reset overwrote (fin.clob).
template <int VEC>
__device__ inline void final_windows(const Params& p, const WgLds& L, int s_first, int n_env,
                                     int lane, uint64_t fv_magic) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const uint32_t V = (uint32_t)(p.W * p.Fobs), VPE = V / VEC, FV = (uint32_t)p.Fobs / VEC;
  // which of the wave's envs ended: one LDS read and a ballot (a loop that looked at one env's
  // flag after the other cost every wave 16 dependent LDS round trips at the tail of the launch:
  // 43.4 us per step against 39.9 without terminal observations, profiles/r02_mode_bench.log)
  unsigned long long ended = __ballot(lane < n_env && (L.fin[s_first + lane].flags & 1));
  while (ended) {  // wave-uniform
    const int el = __ffsll((long long)ended) - 1;
    ended &= ended - 1ull;
    const int s = s_first + el;
    const FinalJob f = L.fin[s];
    const int64_t env = L.job[s].env;
    const float* ring_e = p.ring + env * p.depth * p.nd;
    float* dst = p.final_obs + env * V;
    for (uint32_t j = (uint32_t)lane; j < VPE; j += 64u) {
      vec_t v = load_global<vec_t>((uint64_t)f.src, (int64_t)j);
      const uint32_t w = fastdiv40(j, fv_magic);
      const int col = (int)(j - w * FV) * VEC;
      if (col + VEC > p.Fs) {
        int32_t slot = f.slot0 + (int32_t)w;
        if (!p.persist && slot >= p.W) slot -= p.W;
        float x[GTE_MAX_DYN];
#pragma unroll
        for (int i = 0; i < GTE_MAX_DYN; ++i) {
          x[i] = 0.0f;
          if (i < p.nd) {
            if ((int)w == p.W - 1) x[i] = f.cur[i];
            else if ((int)w < f.n_zero) x[i] = 0.0f;
            else if (slot == f.clob_slot) x[i] = f.clob[i];
            else x[i] = ring_e[(int64_t)slot * p.nd + i];
          }
        }
        if constexpr (VEC == 4) {
          set_tail(v, p.nd, x);
        } else {
          const int i = col - p.Fs;
          float r = x[0];
#pragma unroll
          for (int k = 1; k < GTE_MAX_DYN; ++k) r = (i == k) ? x[k] : r;
          v = r;
        }
      }
      *(vec_t*)(dst + (int64_t)j * VEC) = v;
    }
  }
}

// COOP: wave 0 of the workgroup runs phase A for all 4*EPW (<= 64) envs of the
//       workgroup, one per lane at full lane utilisation (phase A is VALU-issue bound:
//       ~3 000 cycles of fp64 per wave whatever the number of active lanes); otherwise
//       every wave runs phase A for its own EPW envs.
// STAGE: how the dynamic-column values reach the copy loop (STAGE_* above).
#ifndef GTE_GATHER_U
#define GTE_GATHER_U 4  // independent 16-byte loads in flight per lane in the gather
#endif
template <int MODE, int VEC, int NT, bool COOP, int STAGE>
__global__ __launch_bounds__(256) void gte_kernel(const Params p, const uint64_t vpe_magic,
                                                  const uint64_t fv_magic,
                                                  const uint64_t wnd_magic) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gte_smem[];
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;  // wave in block
  // the terminal counter has two slots used alternately, so no memset launch is
  // needed between steps: this launch clears the slot the NEXT launch will use
  if (MODE == MODE_STEP && blockIdx.x == 0 && threadIdx.x == 0) p.term_count_next[0] = 0;
  const int EPB = p.epw * GTE_WAVES;  // envs per workgroup
  const int wg_first = blockIdx.x * EPB;
  if (wg_first >= p.N) return;  // whole workgroup exits together (before any barrier)
  const int n_wg = min(EPB, p.N - wg_first);
  GTE_STAMP(0);
  const WgLds L = carve_lds(gte_smem, EPB, p.final_obs != nullptr, p.log.rows != nullptr);
  const int s_first = wib * p.epw;
  const int n_env = min(p.epw, n_wg - s_first);
  // Which env each slot processes (identity, or the L2-affinity permutation), and the raw
  // rings into LDS.  With cooperative phase A wave 0 goes straight to the state machine
  // (the in-kernel timeline showed it spending 2.8 us staging its own rings first): wave 1
  // covers wave 0's slots as well as its own.
  auto prepare = [&](int first, int count) {
    if (lane < p.epw) {
      const int slot = wg_first + first + lane;
      L.job[first + lane].env = (lane < count) ? (p.perm ? p.perm[slot] : slot) : -1;
    }
    if (STAGE == STAGE_RAW && count > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      stage_raw_rings(p, L, first, count, lane, wnd_magic);
    }
  };
  if (!COOP) {
    prepare(s_first, n_env);
  } else if (wib >= 1) {
    prepare(s_first, n_env);
    if (wib == 1) prepare(0, min(p.epw, n_wg));
  }
  GTE_STAMP(1);  // env ids (perm) + rings arrived
#ifdef GTE_STAMPS_HWID
  GTE_STAMP_HWID(1);  // (diagnostic of the diagnostic: replaces stamp 1)
#endif

  // ---- phase A
  if (!COOP || wib == 0) {  // wave-uniform
    const int s = COOP ? lane : wib * p.epw + lane;  // LDS slot = env within the workgroup
    const bool owns = COOP ? (lane < EPB) : (lane < p.epw);
    const bool active = owns && s < n_wg;
    const int e = active ? (p.perm ? p.perm[wg_first + s] : wg_first + s) : 0;
    ObsJob job;
    FinalJob fin;
    // a step's record stores: into the LDS image (one 64-byte request per env after the barrier)
    const lds_byte_ptr hot = (MODE == MODE_STEP && p.hot_lds) ? (lds_byte_ptr)(L.hot + 64 * s) : (lds_byte_ptr) nullptr;
#ifndef GTE_HOT_ONLY  // p.log: hot_tu_covers() keeps such launches off the isolated TUs
    if (MODE == MODE_STEP && p.log.rows) {
      // gte_step with log_steps: the lane that stepped the env also produces its trajectory row —
      // what History.add records (environments.py:253-264) — from its registers, instead of a
      // second launch reading everything back; the row goes to LDS and the copy waves write it
      // out, five lanes per env (flush_log_rows)
      StepOut so = {};
      phase_a<MODE>(p, e, active, lane, job, p.final_obs ? &fin : nullptr, true, nullptr, nullptr, nullptr,
                    true, nullptr, &so, hot);
      if (active) {
        typedef int4_t __attribute__((address_space(3))) * li4;
        typedef double2_t __attribute__((address_space(3))) * ld2;
        const lds_byte_ptr w = (lds_byte_ptr)(L.logrow + sizeof(LogRow) * s);
        int4_t a = {so.idx, so.step, so.pos, so.dsi};
        double2_t d0 = {so.pv, so.realpos};
        double2_t d1 = {(so.step == 0) ? 0.0 : so.reward, so.asset};  // reset rows: reward 0 (:196)
        double2_t d2 = {so.fiat, so.ia};
        double2_t d3 = {so.ifi, __longlong_as_double((long long)(so.flags & 0xff))};  // flags byte + zero padding
        *(li4)w = a;
        *(ld2)(w + 16) = d0;
        *(ld2)(w + 32) = d1;
        *(ld2)(w + 48) = d2;
        *(ld2)(w + 64) = d3;
      }
    } else
#endif
    phase_a<MODE>(p, e, active, lane, job, p.final_obs ? &fin : nullptr, true, nullptr, nullptr, nullptr, true,
                  nullptr, nullptr, hot);
    if (owns) publish_job(L, s, job);  // slots past the last env get flags = 0
    if (owns && p.final_obs) L.fin[s] = fin;
  }
  // Only LDS has to be visible across the barrier (jobs, env ids, staged rings): nothing
  // after it reads global memory written before it in this launch.  __syncthreads() would
  // also drain wave 0's global stores (record, outputs, ring: 2.5 us in the timeline).
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  GTE_STAMP(6);

  // ---- phase B: each wave writes out the records of its own EPW envs and gathers their windows
  if (MODE == MODE_STEP && p.hot_lds && n_env > 0) flush_hot_records(p, L, s_first, n_env, lane);
#ifndef GTE_HOT_ONLY  // p.log: hot_tu_covers() keeps such launches off the isolated TUs
  if (MODE == MODE_STEP && p.log.rows && n_env > 0) flush_log_rows(p, L, s_first, n_env, lane);
#endif
  if (n_env <= 0 || (p.debug & 1)) return;
  if (p.persist) zero_fresh_stores(p, L, s_first, n_env, lane);
  if (STAGE == STAGE_LATE) {
    stage_dynamic(p, L, s_first, n_env, lane, wnd_magic);
    // LDS operations of one wave execute in order; this only stops the compiler from
    // moving the LDS reads of phase B above the staging writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  bool lean = false;
  if constexpr (MODE == MODE_STEP && VEC == 4 && STAGE == STAGE_RAW) {
    // the lean loop takes whole waves of envs that all copy, in whole passes of U wave instructions
    // (windows of at least one wave instruction: the running counters wrap at most once per step of 64)
    if (p.lean_rows > 0 && n_env == p.epw && n_env * p.W <= LEAN_MAX_ROWS && p.W * p.Fobs / 4 >= 64 &&
        ((uint32_t)n_env * (uint32_t)(p.W * p.Fobs / 4)) % (64u * GTE_LEAN_U) == 0u && !p.debug &&
        __ballot(lane < n_env && !(L.job[s_first + (lane < n_env ? lane : 0)].meta & 1u)) == 0ull) {
      switch (p.nd) {  // wave-uniform, outside the loops
        case 1: resolve_dynamic_rows<1>(p, L, s_first, n_env, lane, wnd_magic); break;
        case 2: resolve_dynamic_rows<2>(p, L, s_first, n_env, lane, wnd_magic); break;
        case 3: resolve_dynamic_rows<3>(p, L, s_first, n_env, lane, wnd_magic); break;
        default: resolve_dynamic_rows<4>(p, L, s_first, n_env, lane, wnd_magic); break;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (this wave's own LDS writes, read back below)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      switch (p.nd) {
        case 1: phase_b_lean<NT, 1>(p, L, s_first, n_env, lane); break;
        case 2: phase_b_lean<NT, 2>(p, L, s_first, n_env, lane); break;
        case 3: phase_b_lean<NT, 3>(p, L, s_first, n_env, lane); break;
        default: phase_b_lean<NT, 4>(p, L, s_first, n_env, lane); break;
      }
      lean = true;
    }
  }
  if (!lean) phase_b<VEC, NT, STAGE, GTE_GATHER_U>(p, L, s_first, n_env, lane, vpe_magic, fv_magic);
  GTE_STAMP(7);
  if (MODE == MODE_STEP && p.final_obs) final_windows<VEC>(p, L, s_first, n_env, lane, fv_magic);
}

#ifndef GTE_HOT_ONLY
// ---------------------------------------------------------------------------
// L2-affinity permutation.  Workgroups are dealt round-robin over the 8 XCDs, each
// with a private 4 MiB L2 (workgroup b and b+8 share one; observed behaviour, used
// for speed only).  A 12.8 MB feature table does not fit one L2, and with random
// starts every XCD reads all of it (measured: 146 MB of L2 misses per step).  If the
// envs processed by XCD x all sit in the x-th eighth of the (dataset, row) space,
// each L2 only has to hold 1/8 of the table and the window reads hit it (measured
// 42 us vs 54 us per step, profiles/r01_tune_affinity_potential.log).  perm[slot] =
// env is a counting sort of the envs by (dataset, row bucket), laid out so that the
// r-th env in sorted order goes to the r-th slot in XCD-major order (slot_of_rank,
// built on the host).  Results do not depend on the permutation; only speed does.
__global__ void gte_affinity_hist_kernel(const Params p, int32_t* hist, int n_bins_per_ds) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const int d = p.rec[e].dsi;
  const int64_t T = p.ds[d].T;
  int b = (int)(((int64_t)p.rec[e].idx * n_bins_per_ds) / T);
  b = b < 0 ? 0 : (b >= n_bins_per_ds ? n_bins_per_ds - 1 : b);
  atomicAdd(&hist[d * n_bins_per_ds + b], 1);
}

// exclusive scan of `hist` (n_bins <= 1024 * per_thread) by one workgroup, into `cursor`; the
// histogram is left ZEROED for the next re-sort (no memset launch per re-sort: the array is
// zero-filled when it is allocated, and this kernel is its only reader)
__global__ __launch_bounds__(1024) void gte_affinity_scan_kernel(int32_t* hist, int32_t* cursor, int n_bins) {
  __shared__ int32_t part[1024];
  const int t = threadIdx.x;
  const int per = (n_bins + 1023) / 1024;
  const int lo = t * per, hi = min(n_bins, lo + per);
  int32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += hist[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const int32_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int32_t run = part[t] - sum;  // exclusive prefix of this thread's chunk
  for (int i = lo; i < hi; ++i) {
    const int32_t c = hist[i];
    hist[i] = 0;
    cursor[i] = run;
    run += c;
  }
}

__global__ void gte_affinity_scatter_kernel(const Params p, int32_t* cursor, int n_bins_per_ds,
                                            const int32_t* slot_of_rank, int32_t* perm_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.N) return;
  const int d = p.rec[e].dsi;
  const int64_t T = p.ds[d].T;
  int b = (int)(((int64_t)p.rec[e].idx * n_bins_per_ds) / T);
  b = b < 0 ? 0 : (b >= n_bins_per_ds ? n_bins_per_ds - 1 : b);
  const int rank = atomicAdd(&cursor[d * n_bins_per_ds + b], 1);
  perm_out[slot_of_rank[rank]] = e;
}

hipError_t launch_affinity_rebuild(const Params& p, int32_t* bins, int n_bins_per_ds,
                                   const int32_t* slot_of_rank, int32_t* perm_out,
                                   hipStream_t stream) {
  // bins: [0, n_bins) the histogram (zero between re-sorts), [n_bins, 2 n_bins) the scatter's cursors
  const int n_bins = p.D * n_bins_per_ds;
  const int blocks = (p.N + 255) / 256;
  hipLaunchKernelGGL(gte_affinity_hist_kernel, dim3(blocks), dim3(256), 0, stream, p, bins,
                     n_bins_per_ds);
  hipLaunchKernelGGL(gte_affinity_scan_kernel, dim3(1), dim3(1024), 0, stream, bins, bins + n_bins, n_bins);
  hipLaunchKernelGGL(gte_affinity_scatter_kernel, dim3(blocks), dim3(256), 0, stream, p, bins + n_bins,
                     n_bins_per_ds, slot_of_rank, perm_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// struct-of-arrays views of the state for the host (gte_get_state)
struct StateSoA {
  int32_t *idx, *step, *pos, *dsi, *start, *episode, *needs_reset;
  double *asset, *fiat, *ia, *ifi, *pv, *realpos;
};

__global__ void gte_extract_state_kernel(const EnvRec* rec, int n, StateSoA o) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const EnvRec r = rec[e];
  o.idx[e] = r.idx; o.step[e] = r.step; o.pos[e] = r.pos; o.dsi[e] = r.dsi;
  o.start[e] = r.start; o.episode[e] = r.episode; o.needs_reset[e] = r.needs_reset;
  o.asset[e] = r.asset; o.fiat[e] = r.fiat; o.ia[e] = r.ia; o.ifi[e] = r.ifi;
  o.pv[e] = r.pv; o.realpos[e] = r.realpos;
}

hipError_t launch_extract_state(const EnvRec* rec, int n, const StateSoA& o, hipStream_t stream) {
  hipLaunchKernelGGL(gte_extract_state_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rec, n, o);
  return hipGetLastError();
}

__global__ void gte_rewind_queue_kernel(EnvRec* rec, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) rec[e].q_head = 0;
}

hipError_t launch_rewind_queue(EnvRec* rec, int n, hipStream_t stream) {
  hipLaunchKernelGGL(gte_rewind_queue_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rec, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers (called from gte_api.hip)

static uint64_t magic40(uint32_t d) { return ((1ull << 40) + d - 1) / d; }

size_t lds_bytes(const Params& p, int stage) {
  const size_t EPB = (size_t)p.epw * GTE_WAVES;
  size_t b = EPB * (16 + 64 + 4 * GTE_MAX_DYN + 4) + (p.final_obs ? EPB * sizeof(FinalJob) : 0) +
             (p.log.rows ? EPB * sizeof(LogRow) : 0);
  if (stage) b += EPB * (size_t)p.W * (size_t)(p.nd ? p.nd : 1) * 4;
  return b;
}

template <int MODE>
static hipError_t launch_mode(const Params& p, int vec, int nt, bool coop, int stage,
                              int blocks, int threads, hipStream_t stream) {
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  const uint64_t vm = magic40(V / vec), fm = magic40((uint32_t)p.Fobs / vec);
  const uint64_t wm = magic40((uint32_t)(p.W * (p.nd ? p.nd : 1)));
  const size_t smem = lds_bytes(p, stage);
#define GTE_L(VEC, NT, CO, ST) \
  hipLaunchKernelGGL((gte_kernel<MODE, VEC, NT, CO, ST>), dim3(blocks), dim3(threads), smem, \
                     stream, p, vm, fm, wm)
#define GTE_L_ST(VEC, NT, CO) do { if (stage == STAGE_RAW) GTE_L(VEC, NT, CO, STAGE_RAW); \
    else if (stage == STAGE_LATE) GTE_L(VEC, NT, CO, STAGE_LATE); else GTE_L(VEC, NT, CO, STAGE_NONE); } while (0)
#define GTE_L_CO(VEC, NT) do { if (coop) GTE_L_ST(VEC, NT, true); else GTE_L_ST(VEC, NT, false); } while (0)
#define GTE_L_NT(VEC) do { if (nt == 2) GTE_L_CO(VEC, 2); else if (nt == 1) GTE_L_CO(VEC, 1); \
                           else GTE_L_CO(VEC, 0); } while (0)
  if (vec == 4) GTE_L_NT(4); else GTE_L_NT(1);
#undef GTE_L_NT
#undef GTE_L_CO
#undef GTE_L_ST
#undef GTE_L
  return hipGetLastError();
}

hipError_t launch_add_orders(const Params& p, const int32_t* pos_index, const double* limit,
                             const uint8_t* persistent, hipStream_t stream) {
  hipLaunchKernelGGL(gte_add_orders_kernel, dim3((p.N + 255) / 256), dim3(256), 0, stream, p,
                     pos_index, limit, persistent);
  return hipGetLastError();
}

hipError_t launch_step(const Params& p, int vec, int nt, bool coop, int stage, int blocks,
                       int threads, hipStream_t stream) {
  return launch_mode<MODE_STEP>(p, vec, nt, coop, stage, blocks, threads, stream);
}

hipError_t launch_reset(const Params& p, int vec, int nt, bool coop, int stage, int blocks,
                        int threads, hipStream_t stream) {
  return launch_mode<MODE_RESET>(p, vec, nt, coop, stage, blocks, threads, stream);
}

#endif  // GTE_HOT_ONLY

}  // namespace gte
