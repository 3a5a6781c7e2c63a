/*
 * gte.h — C ABI of libgte: the MI355X (gfx950) batched trading-environment
 * hot path.
 *
 * The reference (ten2net/Gym-Trading-Env) is pure Python and has no FFI; the
 * boundary it exposes is the Python class API of
 * src/gym_trading_env/environments.py.  This header is the C ABI that sits
 * directly under that class API: each entry point names the reference method
 * (file:line) whose per-environment work it performs for a whole batch of
 * environments resident in HBM.  The Python host layer
 * (the Python files under gym-trading-env_amd/) binds these with ctypes;
 * INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature
 *     (a hipStream_t travels as void*);
 *   - every function returns 0 on success or a negative gte_status; the
 *     message for the last failure on the calling thread is gte_last_error();
 *   - the library owns all state, dataset and (unless gte_bind_outputs is
 *     used) output buffers; pointers handed out stay valid until
 *     gte_destroy; the host never frees them;
 *   - all work is stream-ordered on the env's stream (gte_set_stream) and
 *     asynchronous: gte_step returns after the launch;
 *   - there is NO CPU fallback: every entry point that needs the device fails
 *     with GTE_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef GTE_H_
#define GTE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTE_ABI_VERSION 2
#define GTE_MAX_POSITIONS 32
#define GTE_MAX_DYN 4

typedef enum gte_status {
  GTE_OK = 0,
  GTE_ERR_INVALID = -1,    /* bad argument / config (message says which)     */
  GTE_ERR_NO_DEVICE = -2,  /* no usable HIP device: there is no CPU fallback */
  GTE_ERR_HIP = -3,        /* a HIP runtime call failed                      */
  GTE_ERR_STATE = -4,      /* call order (e.g. step before upload/reset)     */
  GTE_ERR_OOM = -5
} gte_status;

/* dynamic features computable on device
 * (environments.py:20-24 dynamic_feature_last_position_taken / _real_position) */
typedef enum gte_dyn_kind {
  GTE_DYN_LAST_POSITION = 0,
  GTE_DYN_REAL_POSITION = 1
} gte_dyn_kind;

/* rewards computable on device (environments.py:17-18 basic_reward_function;
 * the clipped / scaled forms are the fork's luckymodel/envs/env.py:16-18) */
typedef enum gte_reward_kind {
  GTE_REWARD_LOG_RETURN = 0,         /* ln(pv_t / pv_{t-1})                    */
  GTE_REWARD_SCALED_LOG_RETURN = 1,  /* reward_param0 * ln(pv_t / pv_{t-1})    */
  GTE_REWARD_CLIPPED_LOG_RETURN = 2  /* clip(param0*ln(..), param1, param2)    */
} gte_reward_kind;

/* what a finished environment does on later steps.  The reference env never
 * resets itself; Gymnasium's vector wrappers do
 * (docs/source/vectorize_env.rst:17-33). */
typedef enum gte_autoreset {
  GTE_AUTORESET_DISABLED = 0,  /* caller resets (gte_reset with a mask)        */
  GTE_AUTORESET_NEXT_STEP = 1, /* the step after a terminal one resets: it
                                  ignores the action and returns the reset obs,
                                  reward 0, flags false (Gymnasium >= 1.0)     */
  GTE_AUTORESET_SAME_STEP = 2  /* terminal step returns the reset obs; flags
                                  and reward are the terminal step's           */
} gte_autoreset;

/* Constructor arguments of TradingEnv (environments.py:79-93) for a batch. */
typedef struct gte_config {
  int32_t abi_version;      /* = GTE_ABI_VERSION                               */
  int32_t struct_bytes;     /* = sizeof(gte_config)                            */
  int32_t device;           /* HIP device ordinal                              */
  int32_t n_envs;           /* N environments in this shard                    */
  int32_t n_datasets;       /* D resident datasets (1 for TradingEnv)          */
  int32_t n_static;         /* F_s: static feature columns (:130-133)          */
  int32_t n_dyn;            /* dynamic features (:135-138), <= GTE_MAX_DYN     */
  int32_t dyn_kind[GTE_MAX_DYN];
  int32_t window;           /* W = `windows`; 0 means windows=None (:156-160)  */
  int32_t n_positions;      /* P = len(positions) (:98)                        */
  double  positions[GTE_MAX_POSITIONS];
  double  trading_fees;            /* :102 */
  double  borrow_interest_rate;    /* :103 */
  double  portfolio_initial_value; /* :104 */
  int32_t initial_position_index;  /* index into positions, or -1 = 'random' (:167) */
  int32_t max_episode_duration;    /* 0 = 'max' (:173,:250)                    */
  int32_t reward_kind;
  int32_t autoreset;
  double  reward_param0;
  double  reward_param1;
  double  reward_param2;
  /* MultiDatasetTradingEnv (:365-400) */
  int32_t episodes_between_dataset_switch; /* >= 1                             */
  int32_t dyn_persist;      /* 1: keep a full T-deep dynamic-feature column per
                               env across episodes, exactly like the in-place
                               write into _obs_array (:153-154); 0: W-deep ring,
                               rows before the episode start read as zero      */
  uint64_t seed;            /* device Philox key for reset draws               */
  int64_t  env_id_base;     /* global id of env 0 of this shard (RNG streams
                               are keyed by global env id, so a sharded run
                               equals the unsharded one)                       */
  int32_t envs_per_wave;    /* 0 = choose automatically (so that all workgroups are
                               resident at once where possible), else 1..64       */
  int32_t nontemporal_obs;  /* observation store policy: 0 plain, 1 non-temporal, 2 sc1
                               (1 and 2 keep the feature table in L2; see
                               store_out in csrc/gte_kernels.hip), 3 = automatic:
                               sc1 while the observation buffer fits the Infinity
                               Cache (<= 190 MB), non-temporal beyond             */
  int32_t kernel_variant;   /* 0 = auto.  Bits for A/B timing of the kernel structure:
                               1 = every wave runs phase A for its own envs (no
                               cooperative phase A), 2 = no LDS staging of the
                               dynamic columns, 4 = retired (was the overlapped step
                               kernel, measured slower, removed in round 2; ignored),
                               64 = launch the shared-TU instantiation of the hot
                               kernel instead of the isolated one (gte_hot.hip),
                               128 = gte_rollout runs as one launch per step even
                               where the fused kernels apply, 256 = gte_rollout uses
                               the gather-per-step fused kernel instead of the
                               window-resident one, 1024 = gte_step always appends
                               the trajectory row with a separate small launch
                               (default: the step kernel writes it, through LDS; 2048
                               is accepted and means the default), 4096 = always the
                               generic copy loop (not the lean one that full waves of
                               16-byte-vector windows take), 8192 = the lane that
                               stepped an env stores its record itself (default: the
                               record's per-step half goes through LDS and is
                               written by the copy waves, one 64-byte request per env) */
  int32_t debug_flags;      /* timing ablations only (results become wrong):
                               1 = skip the observation gather, 2 = skip the
                               dynamic-column patch, 8 = skip the window loads
                               (stores only)                                      */
  int32_t affinity_period;  /* L2-affinity processing order: every this many steps the
                               envs are re-sorted by (dataset, table region) so that
                               each XCD's L2 serves one region (speed only; results
                               do not depend on it).  0 = default (max_episode_duration
                               / 16 within [8, 128]; 128 for 'max'), -1 = off        */
  int32_t log_steps;        /* L > 0: keep the last L steps of every env in a device
                               trajectory log (what History records each step,
                               environments.py:253-264); 0 = off                  */
  int32_t reserved2;
  int32_t final_obs;        /* 1 (needs autoreset = same-step): keep the terminal
                               observation of every env that ends, in
                               gte_outputs.final_obs (Gymnasium `final_observation`,
                               SB3 `terminal_observation`)                        */
} gte_config;

/* Device pointers of the per-step return values of TradingEnv.step
 * (environments.py:272) for the whole batch. */
typedef struct gte_outputs {
  float*   obs;        /* f32 [N, W, F_obs] (or [N, F_obs] when window == 0)   */
  float*   reward;     /* f32 [N]                                              */
  double*  reward64;   /* f64 [N]  (the value the f32 one was rounded from)    */
  uint8_t* terminated; /* u8  [N]  `done` (:246)                               */
  uint8_t* truncated;  /* u8  [N]  (:248-251)                                  */
  int32_t* term_count; /* i32 [2]  two slots used alternately (so that no clearing
                          launch sits between steps); term_count[term_slot] is
                          the number of envs whose flags are raised after the
                          last step                                            */
  int32_t* term_ids;   /* i32 [N]  their ids, compacted (order unspecified)    */
  int64_t  obs_elems_per_env; /* W*F_obs                                       */
  int32_t  term_slot;  /* which slot the last launch used (set by gte_get_outputs) */
  int32_t  reserved0;
  float*   final_obs;  /* f32 [N, W, F_obs] or NULL: row e holds the terminal observation
                          of env e after a step in which e ended (same-step mode with
                          gte_config.final_obs); other rows keep older contents     */
} gte_outputs;

/* Device pointers of the per-env state (struct of arrays), i.e. the fields of
 * TradingEnv/Portfolio that History logs each step (:253-264). */
typedef struct gte_state_view {
  int32_t* idx;            /* _idx (:235)                                      */
  int32_t* step;           /* _step (:236)                                     */
  int32_t* position_index; /* index of _position in positions (:210)           */
  int32_t* dataset_index;  /* resident dataset the env trades                  */
  int32_t* start_idx;      /* _idx at the last reset (for Market Return :281)  */
  int32_t* episode;        /* number of resets so far                          */
  int32_t* needs_reset;    /* 1 after a terminal step until the env is reset   */
  double*  asset;          /* Portfolio.asset (portfolio.py:3)                 */
  double*  fiat;           /* Portfolio.fiat                                   */
  double*  interest_asset; /* Portfolio.interest_asset                         */
  double*  interest_fiat;  /* Portfolio.interest_fiat                          */
  double*  portfolio_valuation; /* valorisation at the current row (:241)      */
  double*  real_position;  /* Portfolio.real_position at the current row (:259)*/
} gte_state_view;

typedef struct gte_env gte_env;

/* TradingEnv.__init__ (environments.py:79-125): validates the arguments and
 * allocates HBM for N envs and D datasets.  No data yet. */
int gte_create(const gte_config* cfg, gte_env** out);

/* TradingEnv._set_df (environments.py:128-143): uploads one staged dataset.
 * feat  : host f32 [T, F_obs] row-major, F_obs = n_static + n_dyn, the n_dyn
 *         trailing columns zero — exactly `_obs_array` (:141);
 * close : host f64 [T] — `_price_array` (:143);
 * high/low : host f64 [T] or NULL (only the limit-order path reads them, :221). */
int gte_upload_dataset(gte_env* env, int32_t ds, const float* feat,
                       const double* close, const double* high,
                       const double* low, int64_t T);

/* TradingEnv.reset (environments.py:163-199) for the envs with mask[i] != 0
 * (mask == NULL: all).  All pointers are HOST arrays of length N or NULL.
 * inj_idx / inj_pos_index / inj_dataset >= 0 replace the random draws of
 * :174, :167 and :385 (parity mode); NULL or negative entries use the device
 * Philox stream. */
int gte_reset(gte_env* env, const uint8_t* mask, const int32_t* inj_idx,
              const int32_t* inj_pos_index, const int32_t* inj_dataset);

/* Queue values for the draws of later auto-resets: host i32 [N, n_episodes]
 * arrays (or NULL) indexed by [env][k], k = 0 for the first auto-reset after
 * this call.  Consumed in order; when exhausted the Philox stream takes over. */
int gte_set_autoreset_injection(gte_env* env, int32_t n_episodes,
                                const int32_t* inj_idx,
                                const int32_t* inj_pos_index,
                                const int32_t* inj_dataset);

/* TradingEnv.step (environments.py:233-272) for all N envs in one launch:
 * _take_action/_trade (:204-215) -> Portfolio.trade_to_position
 * (portfolio.py:18-43) -> idx/step advance (:235-236) -> update_interest
 * (portfolio.py:44-46) -> valorisation (portfolio.py:7-13) -> done/truncated
 * (:244-251) -> reward (:17-18,:265-267) -> _get_obs (:152-160).
 * actions: i32 [N] position indices, -1 = None (hold, :234); a host pointer,
 * or a device pointer when actions_on_device != 0.
 * Stream capture: with device-resident actions and log_steps == 0 everything the call enqueues
 * can be captured into a HIP graph by the owner of the env's stream (gte_set_stream).  The
 * terminal counter alternates between its two slots per launch: capture an EVEN number of steps
 * and replay the graph only when gte_get_outputs().term_slot equals its value at capture time
 * (it does after any number of replays and after an even number of eager steps). */
int gte_step(gte_env* env, const int32_t* actions, int32_t actions_on_device);

/* TradingEnv.add_limit_order (environments.py:227-231) for every env with
 * pos_index[i] >= 0 (HOST arrays of length N; persistent == NULL means all
 * non-persistent).  A pending order fills in gte_step at the new row when
 * low <= limit <= high and its target differs from the current position, trading
 * at the limit price (:217-223); gte_reset clears an env's orders (:168).  Needs
 * high/low in every uploaded dataset.  A filled non-persistent order is removed
 * (the reference deletes it while iterating and raises RuntimeError). */
int gte_add_limit_orders(gte_env* env, const int32_t* pos_index, const double* limit,
                         const uint8_t* persistent);

/* Dynamic features the device does not compute (a user's Python callable evaluated vectorised
 * over the batch, environments.py:152-154 `_obs_array[_idx, F_s + i] = f(history)`): overwrite
 * feature i, for every bit i of `mask`, of the CURRENT row of every env with values_device
 * (DEVICE f32 [N, n_dyn]) — in the env's dynamic-feature store, from where the windows of later
 * steps read it, and in the observation the last gte_step / gte_reset produced.  Stream-ordered;
 * call it after every step and reset. */
int gte_set_dynamic_features(gte_env* env, const float* values_device, uint32_t mask);
/* The same from n_dyn separate DEVICE columns, columns_device[i] = f32 or f64 [N] (is_f64[i]) or
 * NULL to leave feature i alone: what a vectorised callable returns per feature, cast to f32 by
 * the kernel the way the reference's assignment into its f32 _obs_array casts (:153-154).
 * Both arrays are HOST arrays of n_dyn entries. */
int gte_set_dynamic_columns(gte_env* env, const void* const* columns_device, const int32_t* is_f64);

/* Device trajectory log (gte_config.log_steps = L): after every gte_reset / gte_step one row per
 * env is appended (by the step kernel itself, or by a small launch).  The log is ONE array of
 * 80-byte records [L, N] (the step kernel writes a row as two requests per env); the pointers
 * below address column c of row 0, env 0, and element (r, e) of a column lives `row_stride`
 * bytes per row and `env_stride` bytes per env further: p + (r % L) * row_stride + e *
 * env_stride.  `rows` counts the rows written so far (the newest is rows - 1).  An episode of
 * env e is the run of rows whose `step` goes 0, 1, 2, ... (step 0 = the reset row). */
typedef struct gte_log_view {
  int32_t* idx;             /* i32 [L, N] _idx  (strided, see above)               */
  int32_t* step;            /* i32 [L, N] _step                                   */
  int32_t* position_index;  /* i32 [L, N]                                         */
  int32_t* dataset_index;   /* i32 [L, N]                                         */
  double*  portfolio_valuation; /* f64 [L, N]                                     */
  double*  real_position;   /* f64 [L, N]                                         */
  double*  reward;          /* f64 [L, N]                                         */
  uint8_t* flags;           /* u8  [L, N] bit0 terminated, bit1 truncated         */
  int64_t  rows;            /* rows written since gte_create                      */
  int32_t  L;
  int32_t  N;
  double*  asset;           /* f64 [L, N] Portfolio.asset / fiat / interest_asset /  */
  double*  fiat;            /*   interest_fiat: what `portfolio_distribution_*`      */
  double*  interest_asset;  /*   derives from (portfolio.py:49-57, History columns   */
  double*  interest_fiat;   /*   environments.py:262)                                */
  int64_t  env_stride;      /* bytes from env e to env e + 1 of the same row          */
  int64_t  row_stride;      /* bytes from row r to row r + 1 of the same env          */
} gte_log_view;
int gte_get_log(gte_env* env, gte_log_view* out);
/* the last `n` (<= L) rows of ONE env, oldest first, into host arrays of length n (any may
 * be NULL); returns the number of rows copied through *n_out */
int gte_read_log(gte_env* env, int32_t env_id, int32_t n, int32_t* idx, int32_t* step,
                 int32_t* position_index, int32_t* dataset_index, double* portfolio_valuation,
                 double* real_position, double* reward, uint8_t* flags, int32_t* n_out);
/* the Portfolio columns of the same rows (asset, fiat, interest_asset, interest_fiat) */
int gte_read_log_portfolio(gte_env* env, int32_t env_id, int32_t n, double* asset, double* fiat,
                           double* interest_asset, double* interest_fiat, int32_t* n_out);
/* The logged EPISODE of each of n_ids envs (HOST array env_ids) in ONE transfer: one kernel packs
 * them into pinned host memory, one stream synchronisation — what History holds for those envs
 * (environments.py:253-264; the reference's add_metric / get_metrics functions and save_for_render
 * read it, :274-307).  Episode of an env = the last run of its logged rows whose `step` counts up
 * by one to the newest row (cut at the front when longer than the log or than max_rows; max_rows
 * <= 0 means L).  finished != 0 (same-step auto-reset with final_obs, right after the step in
 * which the envs ended): the episode that just FINISHED — the rows before the newest one (which
 * already is the next episode's reset row) plus the terminal row from the env's terminal record
 * with that step's reward.  The arrays are [n_ids, max_rows] row-major, rows 0 .. n_rows[j]-1 of
 * env j valid, oldest first; they live in library-owned host memory and stay valid until the next
 * gte_read_log_envs / gte_destroy. */
typedef struct gte_log_batch {
  int32_t n_ids, max_rows;
  const int32_t* n_rows;            /* i32 [n_ids] */
  const int32_t* idx;               /* i32 [n_ids, max_rows] */
  const int32_t* step;
  const int32_t* position_index;
  const int32_t* dataset_index;
  const double*  portfolio_valuation; /* f64 [n_ids, max_rows] */
  const double*  real_position;
  const double*  reward;
  const double*  asset;
  const double*  fiat;
  const double*  interest_asset;
  const double*  interest_fiat;
  const uint8_t* flags;             /* u8 [n_ids, max_rows] bit0 terminated, bit1 truncated */
} gte_log_batch;
int gte_read_log_envs(gte_env* env, const int32_t* env_ids, int32_t n_ids, int32_t max_rows,
                      int32_t finished, gte_log_batch* out);
/* Overwrite the `reward` column of the NEWEST log row with device values f64 [N] — what the
 * reference does with a custom reward_function: `historical_info["reward", -1] = reward`
 * (environments.py:265-267).  Stream-ordered. */
int gte_set_log_reward(gte_env* env, const double* reward_device);
/* A custom reward_function's values for the whole batch (DEVICE f64 [N]), with the reference's
 * rules around them applied by one kernel: reward 0 where the step terminated (environments.py
 * :265: the function is not called when done) and on rows a reset wrote (:196), then written to
 * the f32 and f64 return buffers and to the newest log row (:267).  terminal_view != 0 (same-step
 * auto-reset): an env that ended was shown its terminal row, so its reset row does not zero the
 * reward.  Needs log_steps > 0.  Stream-ordered. */
int gte_apply_reward(gte_env* env, const double* reward_device, int32_t terminal_view);

/* Per-step results of gte_rollout, all device pointers, all optional (NULL = not kept).
 * Row k holds what the k-th gte_step of the sequence would have produced. */
typedef struct gte_rollout_bufs {
  float* obs;           /* f32 [K][N][W][F_obs]; NULL: only the last step's observation is
                           produced, in the env's own obs buffer (gte_get_outputs) */
  float* reward;        /* f32 [K][N] */
  double* reward64;     /* f64 [K][N] */
  uint8_t* terminated;  /* u8  [K][N] */
  uint8_t* truncated;   /* u8  [K][N] */
  double* valuation;    /* f64 [K][N]: portfolio_valuation after each step (environments.py:241) */
} gte_rollout_bufs;

/* n_steps consecutive TradingEnv.step calls (environments.py:233-272) for action sequences
 * known in advance — `actions` is a DEVICE pointer to int32 [n_steps][N], -1 = hold — fused
 * into one launch (gte_rollout.hip).  State, auto-resets, injected draws, limit-order fills and
 * the results are exactly those of n_steps gte_step calls; afterwards the env's own reward /
 * flag buffers and terminal list describe the last step, and its obs buffer holds the last
 * observation unless bufs->obs was given (then that is row n_steps-1 of bufs->obs).  Shapes the
 * fused kernel does not cover (dyn_persist, final_obs, log_steps, scalar-vector layouts) run as
 * n_steps launches of the step kernel with the same results.  Per-step observation rows are
 * written with non-temporal stores under the automatic store policy (they are a stream).
 * Tuning only (results do not depend on it): the environment variable GTE_RESIDENT_EPB = 1..64
 * fixes the envs per workgroup pass of the window-resident kernel instead of the geometry
 * search in gte_api.hip (profiles/r02_resident_epb.log). */
int gte_rollout(gte_env* env, const int32_t* actions, int32_t n_steps, const gte_rollout_bufs* bufs);

/* Where the results of the last gte_step / gte_reset live (device pointers). */
int gte_get_outputs(gte_env* env, gte_outputs* out);
/* Same-step auto-reset with gte_config.final_obs: struct-of-arrays snapshot (device pointers,
 * extracted like gte_get_state) of every env's record AS IT WAS WHEN ITS LAST EPISODE ENDED —
 * the state `TradingEnv.step` reported in `info` before the wrapper reset the env
 * (Gymnasium `final_info`).  Row e is meaningful for the envs listed in term_ids after a step.
 * `episode` / `needs_reset` of the view are not meaningful here. */
int gte_get_final_state(gte_env* env, gte_state_view* out);

/* Snapshot of the per-env state as struct-of-arrays device buffers.  Internally the
 * state is one 128-byte record per env; this call enqueues a small extraction
 * kernel on the env's stream, so the views reflect every launch enqueued before
 * it.  Call it again after later steps (the pointers stay the same). */
int gte_get_state(gte_env* env, gte_state_view* out);

/* Everything the reference's step()/reset() hand back for ONE env (environments.py:253-272:
 * the History row and the return tuple), fetched with one device->host transfer. */
typedef struct gte_env_snapshot {
  int32_t idx, step, position_index, dataset_index;
  int32_t start_idx, episode, needs_reset, terminated, truncated, reserved;
  double asset, fiat, interest_asset, interest_fiat;   /* Portfolio, portfolio.py:1-6 */
  double portfolio_valuation, real_position;           /* :241, :259 */
  double reward;                                       /* f64, :265-267 */
} gte_env_snapshot;

/* State + returns + observations of envs first .. first+count-1 after the last step/reset,
 * into HOST memory: a small kernel (one workgroup per env) packs them into pinned host memory
 * on the env's stream, then the stream is synchronised once.  `out` receives `count` snapshots,
 * `obs` (count * W*F_obs floats) may be NULL.  This is what the host-array ("numpy") mode of
 * the Python classes calls once per step instead of one copy per field. */
int gte_read_envs(gte_env* env, int32_t first, int32_t count, gte_env_snapshot* out, float* obs);
/* The same without the final copy: *out / *obs point INTO the library's pinned host staging
 * buffer (`count` snapshots; `count` * W*F_obs floats, or NULL when want_obs == 0) and stay
 * valid until the next gte_read_envs / gte_read_envs_view / gte_read_env / gte_destroy on this
 * env.  For host-array consumers that can live with buffers being reused every step (what
 * Gymnasium's vector envs call copy=False). */
int gte_read_envs_view(gte_env* env, int32_t first, int32_t count, int32_t want_obs,
                       const gte_env_snapshot** out, const float** obs);
/* gte_read_envs for one env (the N=1 drop-in TradingEnv's per-step call). */
int gte_read_env(gte_env* env, int32_t env_index, gte_env_snapshot* out, float* obs);

/* Use caller-owned device buffers for the outputs (e.g. torch tensors that are
 * then all-gathered over RCCL).  NULL members keep the library's buffer.  The library's own
 * observation buffers ([N, W, F_obs], and the final_obs one) are allocated on first need —
 * the first gte_reset or gte_get_outputs that finds none bound — so a caller that binds its
 * own before resetting never pays for a second copy. */
int gte_bind_outputs(gte_env* env, const gte_outputs* bufs);
/* Redirect ONLY the per-step returns (reward f32[N], terminated u8[N], truncated u8[N]) of
 * the steps enqueued AFTER this call; nothing is synchronised and nothing already
 * enqueued changes.  A sharded run rotates between two (or more) caller-owned return
 * buffers with it, so that the RCCL all-gather of step t's returns (environments.py:272,
 * the `reward, done, truncated` a caller gets back) overlaps step t+1 on another stream.
 * All three pointers are required; the caller keeps the buffers alive while steps that
 * write them or collectives that read them are in flight. */
int gte_bind_returns(gte_env* env, float* reward, uint8_t* terminated, uint8_t* truncated);

/* ---- multi-GPU: the ONE exchange of the sharded path.  Environments shard over
 * the GPUs of a node with no data-path collective; what a caller of the reference's vector env
 * gets back from step() — `reward, done, truncated` for ALL environments (environments.py:272,
 * docs/source/vectorize_env.rst:55-65) — is assembled by an RCCL all-gather over xGMI of the
 * packed per-shard records (reward f32 [N] | terminated u8 [N] | truncated u8 [N], the layout the
 * step kernel writes), and on request of the observations.  One process per GPU; RCCL is bound
 * at run time (no link-time dependency). */
#define GTE_COMM_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId) and hands it to every rank by any host channel */
int gte_comm_unique_id(uint8_t* id_out /* [GTE_COMM_ID_BYTES] */);
/* every rank, collectively: one communicator per env (equal n_envs on every rank) */
int gte_comm_init(gte_env* env, const uint8_t* id, int32_t rank, int32_t world);
/* all-gather `bytes_per_rank` bytes from src_device into dst_device [world * bytes_per_rank]
 * (rank r's block at offset r * bytes_per_rank).  mode 0: on the env's stream — stream-ordered
 * between two steps, no host synchronisation (the synchronous per-step form); mode 1: on the
 * library's communication stream behind an event on the env's stream, overlapping the launches
 * enqueued afterwards (the caller keeps src intact meanwhile: gte_bind_returns rotates return
 * buffers; a block of K rotated rows is one contiguous src) — join with gte_comm_wait(env, back)
 * (orders the env's stream after the mode-1 gather issued `back` gathers ago, 0 = the last one,
 * up to 3: with two rotating return buffers, `gte_comm_wait(env, 1)` before step t makes the
 * gather of step t-2 release the buffer step t rewrites while the gather of step t-1 still
 * overlaps it) or gte_comm_synchronize (blocks the host). */
int gte_allgather(gte_env* env, const void* src_device, void* dst_device,
                  uint64_t bytes_per_rank, int32_t mode);
/* the packed returns of the last step: 6N bytes per rank -> u8 [world, 6N] in dst_device, or in
 * a library-owned buffer when dst_device is NULL (*gathered receives the address used) */
int gte_allgather_returns(gte_env* env, void* dst_device, int32_t mode, const void** gathered);
/* the observations of the last step -> f32 [world * N, W, F_obs] (xGMI-bound at the headline
 * shape: 168 MB per rank and step) */
int gte_allgather_obs(gte_env* env, float* dst_device, int32_t mode);
int gte_comm_wait(gte_env* env, int32_t back);
int gte_comm_synchronize(gte_env* env);
int gte_comm_destroy(gte_env* env); /* also done by gte_destroy */

/* Run on exactly this hipStream_t; NULL is HIP's null (default) stream, which is
 * what PyTorch's default stream is.  A new env runs on a private non-blocking
 * stream until this is called; gte_use_own_stream goes back to it. */
int gte_set_stream(gte_env* env, void* hip_stream);
int gte_use_own_stream(gte_env* env);
int gte_synchronize(gte_env* env);

/* HIP-event timing on the env's stream, for bench.py's roofline figure. */
int gte_timer_start(gte_env* env);
/* elapsed_ms != NULL: record the end event (unless already marked), wait for it, return the
 * span.  elapsed_ms == NULL: only record the end event, asynchronously ("mark"); a later call
 * with a pointer reads it — so that a wall-clock bracket around the same steps does not also
 * pay for the event wait. */
int gte_timer_stop(gte_env* env, float* elapsed_ms);

/* Synchronous device -> host copies (the N=1 drop-in and the tests use them;
 * the batched path keeps everything on the device). */
int gte_read_obs(gte_env* env, int32_t first_env, int32_t n, float* host_dst);
/* device_src must be a pointer obtained from gte_get_outputs / gte_get_state
 * (plus an offset inside that array); waits for the env's stream first. */
int gte_copy_to_host(gte_env* env, const void* device_src, void* host_dst,
                     uint64_t bytes);

/* kernel geometry actually used (for DESIGN.md / bench output).  *vector_bytes = bytes per
 * copy vector + 1000 * flags: bit 0 cooperative phase A, bits 1-2 staging of the dynamic
 * columns (0 none, 1 raw rings in LDS, 2 resolved in LDS), bit 3 unused, bits 4-5
 * the observation store policy in use (0 plain, 1 nt, 2 sc1; never 3), bits 6-9 the resident
 * workgroups per CU the automatic geometry was sized for (0 = not applicable). */
int gte_get_launch_info(gte_env* env, int32_t* envs_per_wave,
                        int32_t* threads_per_block, int32_t* n_blocks,
                        int32_t* vector_bytes);

void gte_destroy(gte_env* env);
const char* gte_last_error(void);
int gte_abi_version(void);
int gte_device_count(void); /* usable HIP devices (0 on a CPU-only host)    */

#ifdef __cplusplus
}
#endif
#endif /* GTE_H_ */
