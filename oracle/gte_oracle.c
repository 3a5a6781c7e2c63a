/*
 * gte_oracle.c — ORACLE.  TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT.
 *
 * A scalar, one-env-at-a-time, fp64 CPU restatement in plain C of the
 * reference's step()/reset() hot path (ten2net/Gym-Trading-Env,
 * src/gym_trading_env/environments.py + utils/portfolio.py), extended to a
 * batch of independent environments exactly the way libgte batches them.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the reported CPU baseline.  The product
 * (libgte.so, HIP) never links, imports or falls back to this file.
 *
 * Parity status: PINNED.  the .npz files in tests/golden/ hold per-step outputs captured
 * from the reference itself, imported unchanged in the build container
 * (tests/golden/make_golden.py); tests/test_oracle_golden.py replays them
 * through this file: idx/step/position/done/truncated bit-exact, fp64 values
 * to <= 1e-12 relative.  The reference ships no tests of its own.
 *
 * Every function cites the reference lines it follows.  Arithmetic is
 * written one IEEE-754 double operation per Python operation, in the same
 * order, and the file is compiled with -ffp-contract=off so that no FMA
 * changes a rounding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gte.h"

typedef struct gto_env {
  gte_config cfg;
  int32_t N, D, Fs, nd, Fobs, W, has_window;
  int64_t depth; /* rows of the per-env dynamic-feature store */
  float** feat;  /* [D] -> f32 [T, Fobs] */
  double** close;
  int64_t* T;
  /* state */
  int32_t *idx, *step, *pos, *ds, *start, *episode, *needs_reset, *eps_on_ds,
      *n_picks;
  double *asset, *fiat, *ia, *ifi, *pv, *realpos;
  float* ring; /* [N, depth, nd] */
  /* outputs */
  float* final_obs; /* [N, W, Fobs] terminal observations (same-step mode, cfg.final_obs) */
  float* obs;
  float* reward;
  double* reward64;
  uint8_t *terminated, *truncated;
  int32_t term_count;
  int32_t* term_ids;
  /* pending limit orders, insertion-ordered like the reference's dict (:227-231) */
  double** high;
  double** low;
  int32_t* lo_n;      /* [N] */
  int32_t* lo_pos;    /* [N, P] target position index */
  double* lo_limit;   /* [N, P] */
  uint8_t* lo_persist;/* [N, P] */
  /* queued draws for auto-resets */
  int32_t q_n;
  int32_t *q_idx, *q_pos, *q_ds, *q_head;
  char err[256];
} gto_env;

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) — the batch's own reset RNG; the
 * reference draws from the global legacy NumPy RNG (environments.py:167,174,
 * 385), which a batch cannot reproduce: parity runs inject the draws.        */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                          uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void gto_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  philox4x32_10(ctr, key, out);
}

/* one block of four draws per (global env id, episode, stream) */
static void reset_draws(const gto_env* E, int32_t e, int32_t episode,
                        uint32_t stream, uint32_t out[4]) {
  int64_t gid = E->cfg.env_id_base + e;
  uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)((uint64_t)gid >> 32),
                     (uint32_t)episode, stream};
  uint32_t key[2] = {(uint32_t)E->cfg.seed, (uint32_t)(E->cfg.seed >> 32)};
  philox4x32_10(ctr, key, out);
}

/* uniform integer in [0, span) from 32 random bits (multiply-shift) */
static int32_t bounded(uint32_t x, int32_t span) {
  return (int32_t)(((uint64_t)x * (uint64_t)(uint32_t)span) >> 32);
}

/* k-th element of the pseudo-random permutation of [0, D) used for pick round
 * `round` of env e: a keyed bijection on b bits (odd multiply, add, xorshift)
 * with cycle walking.  Restates MultiDatasetTradingEnv.next_dataset
 * (environments.py:380-388): "uniform among the least-used datasets" visits
 * every dataset once per round of D picks in uniformly random order. */
static int32_t perm_pick(const gto_env* E, int32_t e, int32_t round, int32_t k) {
  int32_t D = E->D;
  if (D == 1) return 0;
  int b = 1;
  while ((1 << b) < D) ++b;
  uint32_t mask = (b == 32) ? 0xFFFFFFFFu : ((1u << b) - 1u);
  uint32_t r[4];
  reset_draws(E, e, round, 0x44534554u /* 'DSET' */, r);
  uint32_t x = (uint32_t)k;
  int sh = (b + 1) / 2;
  do {
    x = (x * (r[0] | 1u) + r[1]) & mask;
    x ^= x >> sh;
    x = (x * (r[2] | 1u) + r[3]) & mask;
    x ^= x >> sh;
    x = (x * 0x9E3779B1u + (r[0] >> 7)) & mask;
    x ^= x >> sh;
  } while (x >= (uint32_t)D);
  return (int32_t)x;
}

/* ------------------------------------------------------------------------ */
/* Portfolio arithmetic — utils/portfolio.py                                  */

/* Portfolio.valorisation, portfolio.py:7-13: Python's sum() over the list
 * [asset*price, fiat, -interest_asset*price, -interest_fiat] starts from
 * int 0 and adds left to right. */
static double valorisation(double asset, double fiat, double ia, double ifi,
                           double price) {
  double s = 0.0 + asset * price;
  s = s + fiat;
  s = s + (-ia * price);
  s = s + (-ifi);
  return s;
}

/* Python max(0, x): x only when x > 0, else the int 0 */
static double pymax0(double x) { return x > 0.0 ? x : 0.0; }

/* Portfolio.trade_to_position, portfolio.py:18-43 */
static void trade_to_position(double* asset, double* fiat, double* ia,
                              double* ifi, double position, double price,
                              double fees) {
  /* :20 current_position = self.position(price)  (portfolio.py:16-17) */
  double cur = *asset * price / valorisation(*asset, *fiat, *ia, *ifi, price);
  double ratio = 1.0; /* :21 */
  if (position <= 0.0 && cur < 0.0) { /* :22-23 min(1, position/current) */
    double q = position / cur;
    ratio = q < 1.0 ? q : 1.0;
  } else if (position >= 1.0 && cur > 1.0) { /* :24-25 */
    double q = (position - 1.0) / (cur - 1.0);
    ratio = q < 1.0 ? q : 1.0;
  }
  if (ratio < 1.0) { /* :26-30 */
    *asset = *asset - (1.0 - ratio) * *ia;
    *fiat = *fiat - (1.0 - ratio) * *ifi;
    *ia = ratio * *ia;
    *ifi = ratio * *ifi;
  }
  /* :33 asset_trade = position * valorisation(price) / price - asset */
  double trade =
      position * valorisation(*asset, *fiat, *ia, *ifi, price) / price - *asset;
  if (trade > 0.0) { /* :34-38 */
    trade = trade / (1.0 - fees + fees * position);
    double asset_fiat = -trade * price;
    *asset = *asset + trade * (1.0 - fees);
    *fiat = *fiat + asset_fiat;
  } else { /* :39-43 */
    trade = trade / (1.0 - fees * position);
    double asset_fiat = -trade * price;
    *asset = *asset + trade;
    *fiat = *fiat + asset_fiat * (1.0 - fees);
  }
}

/* ------------------------------------------------------------------------ */

static float* ring_row(const gto_env* E, int32_t e, int64_t row) {
  return E->ring + ((int64_t)e * E->depth + (row % E->depth)) * E->nd;
}

/* TradingEnv._get_obs, environments.py:152-160: write the dynamic features of
 * the current row (cast to f32, :154), return row idx or rows idx-W+1..idx. */
static void get_obs(gto_env* E, int32_t e) {
  const int32_t d = E->ds[e];
  const int32_t idx = E->idx[e];
  float cur[GTE_MAX_DYN];
  for (int i = 0; i < E->nd; ++i) {
    double v = (E->cfg.dyn_kind[i] == GTE_DYN_REAL_POSITION)
                   ? E->realpos[e]                      /* :23-24 */
                   : E->cfg.positions[E->pos[e]];       /* :20-21 */
    cur[i] = (float)v;
  }
  if (E->nd > 0) memcpy(ring_row(E, e, idx), cur, sizeof(float) * E->nd);
  float* o = E->obs + (int64_t)e * E->W * E->Fobs;
  for (int32_t w = 0; w < E->W; ++w) {
    int64_t row = (int64_t)idx - E->W + 1 + w;
    const float* src = E->feat[d] + row * E->Fobs;
    memcpy(o, src, sizeof(float) * E->Fs);
    for (int i = 0; i < E->nd; ++i) {
      float v;
      if (row == idx) v = cur[i];
      else if (!E->cfg.dyn_persist && row < E->start[e]) v = 0.0f;
      else v = ring_row(E, e, row)[i];
      o[E->Fs + i] = v;
    }
    o += E->Fobs;
  }
}

/* MultiDatasetTradingEnv.next_dataset, environments.py:380-391 */
static void next_dataset(gto_env* E, int32_t e, int32_t inj_ds) {
  int32_t n = E->n_picks[e]++;
  int32_t d = (inj_ds >= 0) ? inj_ds : perm_pick(E, e, n / E->D, n % E->D);
  E->ds[e] = d;
  E->eps_on_ds[e] = 0; /* :381 */
  if (E->cfg.dyn_persist && E->nd > 0) /* _set_df rebuilds _obs_array :135-141 */
    memset(E->ring + (int64_t)e * E->depth * E->nd, 0,
           sizeof(float) * E->depth * E->nd);
}

/* TradingEnv.reset, environments.py:163-199 (+ MultiDataset reset :393-400) */
static void do_reset(gto_env* E, int32_t e, int32_t inj_idx, int32_t inj_pos,
                     int32_t inj_ds) {
  const gte_config* c = &E->cfg;
  if (E->D > 1) { /* :394-398 */
    E->eps_on_ds[e] += 1;
    if (E->eps_on_ds[e] % c->episodes_between_dataset_switch == 0)
      next_dataset(E, e, inj_ds);
  }
  const int32_t d = E->ds[e];
  uint32_t r[4];
  reset_draws(E, e, E->episode[e], 0x52534554u /* 'RSET' */, r);
  E->episode[e] += 1;
  E->step[e] = 0; /* :166 */
  E->lo_n[e] = 0; /* :168 self._limit_orders = {} */
  int32_t p = c->initial_position_index; /* :167 */
  if (p < 0) p = (inj_pos >= 0) ? inj_pos : bounded(r[0], c->n_positions);
  E->pos[e] = p;
  int32_t idx = E->has_window ? E->W - 1 : 0; /* :171-172 */
  if (c->max_episode_duration > 0) {          /* :173-177 randint(low, high) */
    int32_t low = idx;
    int32_t high = (int32_t)E->T[d] - c->max_episode_duration - idx;
    idx = (inj_idx >= 0) ? inj_idx : low + bounded(r[1], high - low);
  }
  E->idx[e] = idx;
  E->start[e] = idx;
  /* TargetPortfolio, portfolio.py:59-66 */
  double position = c->positions[p];
  double price = E->close[d][idx];
  E->asset[e] = position * c->portfolio_initial_value / price;
  E->fiat[e] = (1.0 - position) * c->portfolio_initial_value;
  E->ia[e] = 0.0;
  E->ifi[e] = 0.0;
  E->pv[e] = c->portfolio_initial_value; /* :194 */
  E->realpos[e] = position;              /* :192 */
  E->needs_reset[e] = 0;
  get_obs(E, e); /* :199 */
}

static double reward_of(const gte_config* c, double pv, double pv_prev) {
  double lr = log(pv / pv_prev); /* basic_reward_function :17-18 */
  switch (c->reward_kind) {
    case GTE_REWARD_SCALED_LOG_RETURN:
      return c->reward_param0 * lr;
    case GTE_REWARD_CLIPPED_LOG_RETURN: {
      double v = c->reward_param0 * lr; /* np.clip(v, lo, hi) */
      double lo = c->reward_param1, hi = c->reward_param2;
      return v < lo ? lo : (v > hi ? hi : v);
    }
    default:
      return lr;
  }
}

static int32_t pop_injection(gto_env* E, int32_t e, int32_t* qi, int32_t* qp,
                             int32_t* qd) {
  *qi = *qp = *qd = -1;
  if (E->q_n <= 0 || E->q_head[e] >= E->q_n) return 0;
  int64_t k = (int64_t)e * E->q_n + E->q_head[e]++;
  if (E->q_idx) *qi = E->q_idx[k];
  if (E->q_pos) *qp = E->q_pos[k];
  if (E->q_ds) *qd = E->q_ds[k];
  return 1;
}

/* TradingEnv.step, environments.py:233-272, for env e.  Returns 1 when the
 * env ended (done or truncated) on this call. */
static int step_one(gto_env* E, int32_t e, int32_t action) {
  const gte_config* c = &E->cfg;
  if (E->needs_reset[e]) {
    if (c->autoreset == GTE_AUTORESET_NEXT_STEP) {
      int32_t qi, qp, qd;
      pop_injection(E, e, &qi, &qp, &qd);
      do_reset(E, e, qi, qp, qd);
      E->reward[e] = 0.0f;
      E->reward64[e] = 0.0;
      E->terminated[e] = 0;
      E->truncated[e] = 0;
      return 0;
    }
    /* DISABLED: like the reference, a finished env keeps stepping while rows
     * remain (the reference raises IndexError past the last row :239; the
     * batch freezes such an env instead) */
    if (E->idx[e] >= E->T[E->ds[e]] - 1) {
      E->reward[e] = 0.0f;
      E->reward64[e] = 0.0;
      return 0;
    }
  }
  const int32_t d = E->ds[e];
  /* :234 _take_action -> :213-215 trade only when the position VALUE differs */
  if (action >= 0) {
    double position = c->positions[action];
    if (position != c->positions[E->pos[e]]) {
      trade_to_position(&E->asset[e], &E->fiat[e], &E->ia[e], &E->ifi[e],
                        position, E->close[d][E->idx[e]], c->trading_fees);
      E->pos[e] = action; /* :210 */
    }
  }
  E->idx[e] += 1;  /* :235 */
  E->step[e] += 1; /* :236 */
  /* :238 _take_action_order_limit, :217-223: every pending order whose target differs
   * from the current position and whose limit lies inside [low, high] of the NEW row
   * trades at the limit price.  A filled non-persistent order is removed (the
   * reference deletes it while iterating its dict and crashes with RuntimeError;
   * the intended behaviour is restated here). */
  if (E->lo_n[e] > 0 && E->high[d] && E->low[d]) {
    const int P = c->n_positions;
    int32_t* lp = E->lo_pos + (int64_t)e * P;
    double* ll = E->lo_limit + (int64_t)e * P;
    uint8_t* lper = E->lo_persist + (int64_t)e * P;
    const double hi = E->high[d][E->idx[e]], lo = E->low[d][E->idx[e]];
    int n = E->lo_n[e], k = 0;
    for (int j = 0; j < n; ++j) {
      int keep = 1;
      double position = c->positions[lp[j]];
      if (position != c->positions[E->pos[e]] && ll[j] <= hi && ll[j] >= lo) {
        trade_to_position(&E->asset[e], &E->fiat[e], &E->ia[e], &E->ifi[e], position, ll[j],
                          c->trading_fees);
        E->pos[e] = lp[j];
        if (!lper[j]) keep = 0;
      }
      if (keep) { lp[k] = lp[j]; ll[k] = ll[j]; lper[k] = lper[j]; ++k; }
    }
    E->lo_n[e] = k;
  }
  double price = E->close[d][E->idx[e]]; /* :239 */
  /* Portfolio.update_interest, portfolio.py:44-46 (assignment, not +=) */
  E->ia[e] = pymax0(-E->asset[e]) * c->borrow_interest_rate;
  E->ifi[e] = pymax0(-E->fiat[e]) * c->borrow_interest_rate;
  double pv = valorisation(E->asset[e], E->fiat[e], E->ia[e], E->ifi[e], price); /* :241 */
  int done = (pv / c->portfolio_initial_value) <= 0.7; /* :246 */
  int trunc = E->idx[e] >= E->T[d] - 1;                /* :248 */
  if (c->max_episode_duration > 0 &&
      E->step[e] >= c->max_episode_duration - 1) /* :250 */
    trunc = 1;
  /* :259 Portfolio.real_position, portfolio.py:14-15 */
  E->realpos[e] = (E->asset[e] - E->ia[e]) * price /
                  valorisation(E->asset[e], E->fiat[e], E->ia[e], E->ifi[e], price);
  double rew = 0.0; /* :263, stays 0 when done (:265) */
  if (!done) rew = reward_of(c, pv, E->pv[e]);
  E->pv[e] = pv;
  E->reward64[e] = rew;
  E->reward[e] = (float)rew;
  E->terminated[e] = (uint8_t)done;
  E->truncated[e] = (uint8_t)trunc;
  int ended = done || trunc;
  if (ended) E->needs_reset[e] = 1;
  if (ended && c->autoreset == GTE_AUTORESET_SAME_STEP) {
    /* the reference's step() runs _get_obs at :272 before any wrapper resets the env: the
     * terminal row's dynamic features are written (matters for dyn_persist), and that
     * observation is what Gymnasium / SB3 report as the final one */
    get_obs(E, e);
    if (c->final_obs) {
      memcpy(E->final_obs + (int64_t)e * E->W * E->Fobs, E->obs + (int64_t)e * E->W * E->Fobs,
             sizeof(float) * E->W * E->Fobs);
    }
    int32_t qi, qp, qd;
    pop_injection(E, e, &qi, &qp, &qd);
    do_reset(E, e, qi, qp, qd); /* writes the reset observation */
  } else {
    get_obs(E, e); /* :272 */
  }
  return ended;
}

/* ------------------------------------------------------------------------ */
/* batch API (ctypes-bound by oracle/oracle.py)                               */

static void* zalloc(size_t n) { return calloc(n ? n : 1, 1); }

gto_env* gto_create(const gte_config* cfg) {
  if (!cfg || cfg->abi_version != GTE_ABI_VERSION ||
      cfg->struct_bytes != (int32_t)sizeof(gte_config))
    return NULL;
  gto_env* E = (gto_env*)zalloc(sizeof(gto_env));
  E->cfg = *cfg;
  E->N = cfg->n_envs;
  E->D = cfg->n_datasets;
  E->Fs = cfg->n_static;
  E->nd = cfg->n_dyn;
  E->Fobs = E->Fs + E->nd;
  E->has_window = cfg->window > 0;
  E->W = E->has_window ? cfg->window : 1;
  E->feat = (float**)zalloc(sizeof(float*) * E->D);
  E->high = (double**)zalloc(sizeof(double*) * E->D);
  E->low = (double**)zalloc(sizeof(double*) * E->D);
  E->close = (double**)zalloc(sizeof(double*) * E->D);
  E->T = (int64_t*)zalloc(sizeof(int64_t) * E->D);
  size_t N = (size_t)E->N;
#define I32(name) E->name = (int32_t*)zalloc(sizeof(int32_t) * N)
#define F64(name) E->name = (double*)zalloc(sizeof(double) * N)
  I32(idx); I32(step); I32(pos); I32(ds); I32(start); I32(episode);
  I32(needs_reset); I32(eps_on_ds); I32(n_picks); I32(term_ids); I32(q_head); I32(lo_n);
  E->lo_pos = (int32_t*)zalloc(sizeof(int32_t) * N * cfg->n_positions);
  E->lo_limit = (double*)zalloc(sizeof(double) * N * cfg->n_positions);
  E->lo_persist = (uint8_t*)zalloc(N * cfg->n_positions);
  F64(asset); F64(fiat); F64(ia); F64(ifi); F64(pv); F64(realpos); F64(reward64);
#undef I32
#undef F64
  E->reward = (float*)zalloc(sizeof(float) * N);
  E->terminated = (uint8_t*)zalloc(N);
  E->truncated = (uint8_t*)zalloc(N);
  E->obs = (float*)zalloc(sizeof(float) * N * E->W * E->Fobs);
  E->final_obs = (float*)zalloc(sizeof(float) * N * E->W * E->Fobs);
  return E;
}

int gto_upload_dataset(gto_env* E, int32_t d, const float* feat,
                       const double* close, int64_t T) {
  if (!E || d < 0 || d >= E->D || T <= 0) return -1;
  free(E->feat[d]);
  free(E->close[d]);
  E->feat[d] = (float*)malloc(sizeof(float) * T * E->Fobs);
  E->close[d] = (double*)malloc(sizeof(double) * T);
  memcpy(E->feat[d], feat, sizeof(float) * T * E->Fobs);
  memcpy(E->close[d], close, sizeof(double) * T);
  E->T[d] = T;
  return 0;
}

/* allocate the dynamic-feature store once every dataset is known; MultiDataset
 * construction consumes pick 0 for every env (environments.py:378) */
static int finalize(gto_env* E) {
  if (E->ring || E->depth) return 0;
  int64_t maxT = 0;
  for (int d = 0; d < E->D; ++d) {
    if (E->T[d] <= 0) return -1;
    if (E->T[d] > maxT) maxT = E->T[d];
  }
  E->depth = E->cfg.dyn_persist ? maxT : E->W;
  E->ring = (float*)zalloc(sizeof(float) * (size_t)E->N * E->depth *
                           (E->nd ? E->nd : 1));
  return 0;
}

int gto_upload_high_low(gto_env* E, int32_t d, const double* high, const double* low) {
  if (!E || d < 0 || d >= E->D || E->T[d] <= 0) return -1;
  free(E->high[d]); free(E->low[d]);
  E->high[d] = (double*)malloc(sizeof(double) * E->T[d]);
  E->low[d] = (double*)malloc(sizeof(double) * E->T[d]);
  memcpy(E->high[d], high, sizeof(double) * E->T[d]);
  memcpy(E->low[d], low, sizeof(double) * E->T[d]);
  return 0;
}

/* TradingEnv.add_limit_order, environments.py:227-231, one order per env with
 * pos_index[e] >= 0: `self._limit_orders[position] = {...}` — an existing key keeps its
 * place in the iteration order, a new key goes last. */
int gto_add_limit_orders(gto_env* E, const int32_t* pos_index, const double* limit,
                         const uint8_t* persistent) {
  const int P = E->cfg.n_positions;
  for (int32_t e = 0; e < E->N; ++e) {
    int32_t pi = pos_index[e];
    if (pi < 0) continue;
    if (pi >= P) return -1;
    int32_t* lp = E->lo_pos + (int64_t)e * P;
    int n = E->lo_n[e], j = 0;
    /* dict keys are position VALUES */
    while (j < n && E->cfg.positions[lp[j]] != E->cfg.positions[pi]) ++j;
    if (j == n) { if (n >= P) return -1; E->lo_n[e] = n + 1; }
    lp[j] = pi;
    E->lo_limit[(int64_t)e * P + j] = limit[e];
    E->lo_persist[(int64_t)e * P + j] = persistent ? persistent[e] : 0;
  }
  return 0;
}

int gto_reset(gto_env* E, const uint8_t* mask, const int32_t* inj_idx,
              const int32_t* inj_pos, const int32_t* inj_ds) {
  if (finalize(E)) return -1;
  for (int32_t e = 0; e < E->N; ++e) {
    if (mask && !mask[e]) continue;
    if (E->D > 1 && E->n_picks[e] == 0) /* the constructor's next_dataset() :378 */
      next_dataset(E, e, inj_ds ? inj_ds[e] : -1);
    do_reset(E, e, inj_idx ? inj_idx[e] : -1, inj_pos ? inj_pos[e] : -1,
             inj_ds ? inj_ds[e] : -1);
    E->reward[e] = 0.0f;
    E->reward64[e] = 0.0;
    E->terminated[e] = 0;
    E->truncated[e] = 0;
  }
  E->term_count = 0;
  return 0;
}

int gto_set_autoreset_injection(gto_env* E, int32_t n, const int32_t* qi,
                                const int32_t* qp, const int32_t* qd) {
  free(E->q_idx); free(E->q_pos); free(E->q_ds);
  E->q_idx = E->q_pos = E->q_ds = NULL;
  E->q_n = n;
  memset(E->q_head, 0, sizeof(int32_t) * E->N);
  size_t bytes = sizeof(int32_t) * (size_t)E->N * (n > 0 ? n : 0);
  if (n > 0 && qi) { E->q_idx = (int32_t*)malloc(bytes); memcpy(E->q_idx, qi, bytes); }
  if (n > 0 && qp) { E->q_pos = (int32_t*)malloc(bytes); memcpy(E->q_pos, qp, bytes); }
  if (n > 0 && qd) { E->q_ds = (int32_t*)malloc(bytes); memcpy(E->q_ds, qd, bytes); }
  return 0;
}

/* one step of every env; `threads` > 1 splits the env range with OpenMP (the
 * envs are independent, so the result does not depend on the thread count) */
int gto_step(gto_env* E, const int32_t* actions, int32_t threads) {
  if (!E->ring && finalize(E)) return -1;
  int32_t N = E->N;
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int32_t e = 0; e < N; ++e) step_one(E, e, actions[e]);
  int32_t n = 0;
  for (int32_t e = 0; e < N; ++e)
    if (E->terminated[e] || E->truncated[e]) E->term_ids[n++] = e;
  E->term_count = n;
  return 0;
}

#define GETTER(type, name, field) \
  type* gto_get_##name(gto_env* E) { return E->field; }
GETTER(float, obs, obs)
GETTER(float, final_obs, final_obs)
GETTER(float, reward, reward)
GETTER(double, reward64, reward64)
GETTER(uint8_t, terminated, terminated)
GETTER(uint8_t, truncated, truncated)
GETTER(int32_t, term_ids, term_ids)
GETTER(int32_t, idx, idx)
GETTER(int32_t, step, step)
GETTER(int32_t, position_index, pos)
GETTER(int32_t, dataset_index, ds)
GETTER(int32_t, start_idx, start)
GETTER(int32_t, episode, episode)
GETTER(int32_t, needs_reset, needs_reset)
GETTER(double, asset, asset)
GETTER(double, fiat, fiat)
GETTER(double, interest_asset, ia)
GETTER(double, interest_fiat, ifi)
GETTER(double, portfolio_valuation, pv)
GETTER(double, real_position, realpos)
int32_t gto_term_count(gto_env* E) { return E->term_count; }

/* known-answer hook for the Portfolio arithmetic alone */
void gto_portfolio_trade(double* state4, double position, double price,
                         double fees, double rate, double next_price,
                         double* valuation, double* real_position) {
  trade_to_position(&state4[0], &state4[1], &state4[2], &state4[3], position,
                    price, fees);
  state4[2] = pymax0(-state4[0]) * rate;
  state4[3] = pymax0(-state4[1]) * rate;
  *valuation = valorisation(state4[0], state4[1], state4[2], state4[3], next_price);
  *real_position = (state4[0] - state4[2]) * next_price / *valuation;
}

void gto_destroy(gto_env* E) {
  if (!E) return;
  for (int d = 0; d < E->D; ++d) { free(E->feat[d]); free(E->close[d]); free(E->high[d]); free(E->low[d]); }
  free(E->feat); free(E->close); free(E->T); free(E->high); free(E->low);
  free(E->lo_n); free(E->lo_pos); free(E->lo_limit); free(E->lo_persist);
  free(E->idx); free(E->step); free(E->pos); free(E->ds); free(E->start);
  free(E->episode); free(E->needs_reset); free(E->eps_on_ds); free(E->n_picks);
  free(E->asset); free(E->fiat); free(E->ia); free(E->ifi); free(E->pv);
  free(E->realpos); free(E->ring); free(E->obs); free(E->final_obs); free(E->reward);
  free(E->reward64); free(E->terminated); free(E->truncated); free(E->term_ids);
  free(E->q_idx); free(E->q_pos); free(E->q_ds); free(E->q_head);
  free(E);
}
