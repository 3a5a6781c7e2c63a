/* c_abi_demo.c — libgte from plain C (no Python, no torch): what a C host needs to run the
 * batched TradingEnv.step()/reset() hot path and the sharded run's return exchange.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lgym-trading-env_amd/csrc -lgte \
 *       -Wl,-rpath,$PWD/gym-trading-env_amd/csrc -lm -o /tmp/c_abi_demo && /tmp/c_abi_demo [n_envs] [steps]
 *
 * One process = one GPU = one shard.  Here world = 1 (the RCCL all-gather degenerates to a copy
 * but runs the real path); with more ranks, rank 0's id travels over the caller's own transport
 * (MPI_Bcast, a socket, a file) — see INTEGRATION.md §5.
 * Exit codes: 0 ok, 3 no usable gfx950 device (libgte has no CPU fallback), 1 anything else. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gte.h"

#define CHECK(call)                                                            \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != GTE_OK) {                                                       \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, gte_last_error());        \
      return rc_ == GTE_ERR_NO_DEVICE ? 3 : 1;                                 \
    }                                                                          \
  } while (0)

static double uniform(unsigned long long* s) { /* xorshift64*: data + actions, nothing else */
  *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
  return (double)((*s * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 200;
  const int T = 20000, Fs = 14, nd = 2, W = 8, Fobs = Fs + nd;
  unsigned long long seed = 88172645463325252ull;

  /* TradingEnv(df, positions=[-1,0,1], windows=8, trading_fees=1e-4, borrow_interest_rate=3e-6,
   *            max_episode_duration=100)  (environments.py:79-93) for N envs */
  gte_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = GTE_ABI_VERSION;
  cfg.struct_bytes = (int32_t)sizeof cfg;
  cfg.device = 0;
  cfg.n_envs = N;
  cfg.n_datasets = 1;
  cfg.n_static = Fs;
  cfg.n_dyn = nd;
  cfg.dyn_kind[0] = GTE_DYN_LAST_POSITION;
  cfg.dyn_kind[1] = GTE_DYN_REAL_POSITION;
  cfg.window = W;
  cfg.n_positions = 3;
  cfg.positions[0] = -1; cfg.positions[1] = 0; cfg.positions[2] = 1;
  cfg.trading_fees = 1e-4;
  cfg.borrow_interest_rate = 3e-6;
  cfg.portfolio_initial_value = 1000;
  cfg.initial_position_index = -1; /* 'random' */
  cfg.max_episode_duration = 100;
  cfg.reward_kind = GTE_REWARD_LOG_RETURN;
  cfg.autoreset = GTE_AUTORESET_NEXT_STEP;
  cfg.episodes_between_dataset_switch = 1;
  cfg.seed = 7;
  cfg.nontemporal_obs = 3; /* automatic store policy */

  gte_env* env = NULL;
  CHECK(gte_create(&cfg, &env));

  /* _set_df (environments.py:128-143): f32 [T, F_obs] row-major with zero dynamic columns, f64 close */
  float* feat = calloc((size_t)T * Fobs, sizeof *feat);
  double* close = malloc(sizeof *close * T);
  double lp = log(100.0);
  for (int t = 0; t < T; ++t) {
    lp += (uniform(&seed) - 0.5) * 4e-3;
    close[t] = exp(lp);
    for (int c = 0; c < Fs; ++c) feat[(size_t)t * Fobs + c] = (float)(uniform(&seed) - 0.5);
  }
  CHECK(gte_upload_dataset(env, 0, feat, close, NULL, NULL, T));

  /* the sharded run's exchange: one communicator per env, here a group of one */
  uint8_t id[GTE_COMM_ID_BYTES];
  CHECK(gte_comm_unique_id(id));
  CHECK(gte_comm_init(env, id, /*rank=*/0, /*world=*/1));

  CHECK(gte_reset(env, NULL, NULL, NULL, NULL));
  int32_t* actions = malloc(sizeof *actions * N);
  uint8_t* returns = malloc((size_t)6 * N); /* reward f32 [N] | terminated u8 [N] | truncated u8 [N] */
  gte_env_snapshot snap;
  float* obs0 = malloc(sizeof *obs0 * W * Fobs);
  double reward_sum = 0.0;
  long ended = 0;
  CHECK(gte_timer_start(env));
  for (int k = 0; k < steps; ++k) {
    for (int e = 0; e < N; ++e) actions[e] = (int32_t)(uniform(&seed) * 3.0);
    CHECK(gte_step(env, actions, /*actions_on_device=*/0));
    const void* gathered = NULL; /* u8 [world][6N] on the device */
    CHECK(gte_allgather_returns(env, NULL, /*mode=*/0, &gathered));
    if (k % 50 == 48 || k == steps - 1) { /* a learner would consume them on the device */
      CHECK(gte_copy_to_host(env, gathered, returns, (uint64_t)6 * N));
      const float* reward = (const float*)returns;
      for (int e = 0; e < N; ++e) {
        reward_sum += reward[e];
        ended += returns[4 * (size_t)N + e] | returns[5 * (size_t)N + e];
      }
    }
  }
  float ms = 0.f;
  CHECK(gte_timer_stop(env, &ms));
  CHECK(gte_read_env(env, 0, &snap, obs0)); /* state + returns + observation of env 0, one transfer */
  printf("c_abi_demo ok: %d envs x %d steps in %.2f ms = %.3g env-steps/s; env 0: idx %d step %d "
         "valuation %.4f obs[last row][0] %.5f; sampled reward sum %.6f, %ld episode ends seen\n",
         N, steps, ms, (double)N * steps / (ms * 1e-3), snap.idx, snap.step, snap.portfolio_valuation,
         obs0[(W - 1) * Fobs], reward_sum, ended);
  if (!(snap.portfolio_valuation > 0.0) || obs0[(W - 1) * Fobs] != feat[(size_t)snap.idx * Fobs]) {
    fprintf(stderr, "inconsistent results\n");
    return 1;
  }
  CHECK(gte_comm_destroy(env));
  gte_destroy(env);
  free(feat); free(close); free(actions); free(returns); free(obs0);
  return 0;
}
