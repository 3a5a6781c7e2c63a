// gte_api.hip — host side of libgte: the C ABI declared in include/gte.h.
//
// Owns the HBM-resident datasets, per-env state and outputs, validates the
// arguments the way the reference constructor / reset do
// (environments.py:79-125,163-199) and launches the kernels of
// gte_kernels.hip.  No CPU path exists: without a gfx950 device every entry
// point that needs one fails with GTE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gte_device.h"

namespace gte {
hipError_t launch_step(const Params& p, int vec, int nt, bool coop, int stage, int blocks,
                       int threads, hipStream_t stream);
hipError_t launch_reset(const Params& p, int vec, int nt, bool coop, int stage, int blocks,
                        int threads, hipStream_t stream);
size_t lds_bytes(const Params& p, int stage);
hipError_t launch_step_hot(const Params& p, int blocks, int threads, size_t smem, hipStream_t stream);
hipError_t launch_step_hot_nt(const Params& p, int blocks, int threads, size_t smem, hipStream_t stream);
int hot_blocks_per_cu(size_t smem);
int hot_blocks_per_cu_nt(size_t smem);
struct RolloutArgs {  // mirrors gte_rollout.hip
  const int32_t* actions; int32_t K; float* obs; float* reward; double* reward64;
  uint8_t* terminated; uint8_t* truncated; double* valuation; int32_t epb;
  int32_t n_groups; int32_t* group_counter;
};
size_t resident_lds_bytes(const Params& p, int epb);
int resident_blocks_per_cu(const Params& p, int epb, int nt);
hipError_t launch_rollout_resident(const Params& p, const RolloutArgs& r, int nt, int blocks,
                                   hipStream_t stream);
hipError_t launch_rollout_state(const Params& p, const RolloutArgs& r, int n_steps, int epw,
                                hipStream_t stream);
hipError_t launch_set_dynamic_columns(const Params& p, const void* const* cols, const int32_t* is_f64,
                                      hipStream_t stream);
hipError_t launch_apply_reward(const Params& p, const double* reward, LogRow* newest, int terminal_view,
                               hipStream_t stream);
hipError_t launch_rollout(const Params& p, const RolloutArgs& r, int nt, int blocks, int threads,
                          hipStream_t stream);
int rollout_blocks_per_cu(const Params& p, int nt);
struct StateSoA {
  int32_t *idx, *step, *pos, *dsi, *start, *episode, *needs_reset;
  double *asset, *fiat, *ia, *ifi, *pv, *realpos;
};
hipError_t launch_extract_state(const EnvRec* rec, int n, const StateSoA& o, hipStream_t stream);
hipError_t launch_rewind_queue(EnvRec* rec, int n, hipStream_t stream);
hipError_t launch_log(const EnvRec* rec, const double* reward64, const uint8_t* term,
                      const uint8_t* trunc, int n, int64_t row_base, const LogArrays& o,
                      hipStream_t stream);
hipError_t launch_snapshot(const EnvRec* rec, const double* reward64, const uint8_t* term,
                           const uint8_t* trunc, const float* obs, int64_t obs_elems, int first,
                           int count, void* dst, float* dst_obs, hipStream_t stream);
struct LogPack {
  int32_t* n_rows;
  int32_t *idx, *step, *pos, *dsi;
  double *pv, *realpos, *reward, *asset, *fiat, *ia, *ifi;
  uint8_t* flags;
};
hipError_t launch_pack_log(const LogArrays& log, int N, int L, long long rows_written, const int32_t* ids,
                           int n_ids, int max_rows, int finished, const EnvRec* final_rec,
                           const double* reward64, const LogPack& o, hipStream_t stream);
const char* rccl_load();
const char* rccl_error(int code);
int rccl_unique_id(uint8_t* out128);
int rccl_comm_init(void** comm, const uint8_t* id128, int rank, int world);
int rccl_allgather_bytes(void* comm, const void* src, void* dst, size_t bytes, hipStream_t stream);
int rccl_comm_destroy(void* comm);
hipError_t launch_set_dynamic(const Params& p, const float* values, uint32_t mask, hipStream_t stream);
hipError_t launch_affinity_rebuild(const Params& p, int32_t* bins, int n_bins_per_ds,
                                   const int32_t* slot_of_rank, int32_t* perm_out,
                                   hipStream_t stream);
hipError_t launch_add_orders(const Params& p, const int32_t* pos_index, const double* limit,
                             const uint8_t* persistent, hipStream_t stream);
}  // namespace gte

using gte::DatasetDesc;
using gte::EnvRec;
using gte::Params;

static thread_local std::string g_err = "";

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(e_ == hipErrorOutOfMemory ? GTE_ERR_OOM : GTE_ERR_HIP,          \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                 \
  } while (0)

struct gte_env {
  gte_config cfg;
  Params p;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<void*> allocs;       // everything freed in gte_destroy
  std::vector<DatasetDesc> h_ds;   // host copy of the descriptor table
  std::vector<void*> ds_allocs[4]; // per dataset: feat, close, high, low
  DatasetDesc* d_ds = nullptr;
  gte_outputs owned;
  // staging buffers for host-side arguments
  int32_t* d_actions = nullptr;
  uint8_t* d_mask = nullptr;
  int32_t *d_inj_idx = nullptr, *d_inj_pos = nullptr, *d_inj_ds = nullptr;
  int32_t *d_q_idx = nullptr, *d_q_pos = nullptr, *d_q_ds = nullptr;
  double* d_lo_limit_in = nullptr;  // staging for gte_add_limit_orders
  uint8_t* d_lo_persist_in = nullptr;
  bool finalized = false;
  bool was_reset = false;
  int vec = 1, blocks = 0, threads = 256;
  bool coop = false;       // wave 0 of a workgroup runs phase A for the whole workgroup
  int stage = 0;           // dynamic columns: 0 global, 1 raw rings in LDS, 2 resolved in LDS
  int32_t* term_base = nullptr;  // the two-slot terminal counter in use (owned or bound)
  int term_slot = 0;       // slot the last launch added to
  // L2-affinity processing order (gte_kernels.hip, "L2-affinity permutation")
  int affinity_period = 0;       // 0 = off
  int steps_since_rebuild = 0;
  int n_bins_per_ds = 0;
  int32_t* d_perm = nullptr;
  int32_t* d_slot_of_rank = nullptr;
  int32_t* d_bins = nullptr;
  gte::StateSoA soa = {};  // host-facing struct-of-arrays mirrors (gte_get_state)
  gte::StateSoA fsoa = {}; // the same for the terminal records (gte_get_final_state)
  gte::LogArrays log = {}; // device trajectory log (cfg.log_steps rows per env)
  int64_t log_rows = 0;
  int rollout_epw = 0;     // envs per wavefront of the fused rollout kernel (0 = not chosen yet)
  int resident_slots[3] = {0, 0, 0};  // workgroups of that geometry the chip holds at once
  int32_t* d_group_counter = nullptr; // the resident kernel's work queue (next group of envs)
  int resident_epb[3] = {0, 0, 0};  // envs per workgroup of the window-resident rollout kernel per
                                    // store policy (0 = not chosen yet, -1 = shape not covered)
  int hot_per_cu = 0;      // resident workgroups per CU the geometry was sized for (0 = n/a)
  bool timer_marked = false;  // gte_timer_stop(NULL) recorded the end event already
  bool store_auto = false; // the observation store policy was chosen here (cfg said 3)
  // multi-GPU return exchange (gte_comm.hip): one RCCL communicator per env
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ready = nullptr;
  hipEvent_t comm_done[4] = {nullptr, nullptr, nullptr, nullptr};  // ring: one per overlapped gather
  int64_t comm_seq = 0;                                              // overlapped gathers so far
  uint8_t* gathered_returns = nullptr;  // u8 [world, 6N], library-owned destination
  void* h_snap = nullptr;  // pinned host memory for gte_read_envs: snapshots, then observations
  size_t h_snap_bytes = 0;
  bool view_reads = false; // gte_read_envs_view has been used: the buffer is kept at full size
  void* h_logpack = nullptr;  // pinned host memory for gte_read_log_envs
  size_t h_logpack_bytes = 0;
};

template <typename T>
static int dev_alloc(gte_env* E, T** out, size_t count, bool zero = true) {
  void* ptr = nullptr;
  size_t bytes = sizeof(T) * (count ? count : 1);
  HIPCHK(hipMalloc(&ptr, bytes));
  if (getenv("GTE_DEBUG_ALLOC")) fprintf(stderr, "[gte] alloc %p %zu bytes\n", ptr, bytes);
  E->allocs.push_back(ptr);
  if (zero) HIPCHK(hipMemset(ptr, 0, bytes));
  *out = (T*)ptr;
  return GTE_OK;
}

#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != GTE_OK) return rc_; \
  } while (0)

static int validate(const gte_config* c) {
  if (!c) return fail(GTE_ERR_INVALID, "config is NULL");
  if (c->abi_version != GTE_ABI_VERSION || c->struct_bytes != (int32_t)sizeof(gte_config))
    return fail(GTE_ERR_INVALID, "ABI mismatch: header version %d / %d bytes, library %d / %zu",
                c->abi_version, c->struct_bytes, GTE_ABI_VERSION, sizeof(gte_config));
  if (c->n_envs <= 0) return fail(GTE_ERR_INVALID, "n_envs must be > 0");
  if (c->n_datasets <= 0) return fail(GTE_ERR_INVALID, "n_datasets must be > 0");
  if (c->n_static < 0 || c->n_dyn < 0 || c->n_dyn > GTE_MAX_DYN)
    return fail(GTE_ERR_INVALID, "n_static >= 0 and 0 <= n_dyn <= %d required", GTE_MAX_DYN);
  if (c->n_static + c->n_dyn <= 0) return fail(GTE_ERR_INVALID, "observation has no columns");
  if (c->n_static + c->n_dyn > 2048) return fail(GTE_ERR_INVALID, "F_obs > 2048 unsupported");
  for (int i = 0; i < c->n_dyn; ++i)
    if (c->dyn_kind[i] != GTE_DYN_LAST_POSITION && c->dyn_kind[i] != GTE_DYN_REAL_POSITION)
      return fail(GTE_ERR_INVALID, "dyn_kind[%d] = %d unknown", i, c->dyn_kind[i]);
  if (c->window < 0) return fail(GTE_ERR_INVALID, "window must be >= 0");
  if (c->window >= 32768)  // JobRec.meta keeps the zero-row count of a window in 15 bits
    return fail(GTE_ERR_INVALID, "window must be < 32768");
  const int64_t W = c->window > 0 ? c->window : 1;
  if (W * (c->n_static + c->n_dyn) > (1 << 18))
    return fail(GTE_ERR_INVALID, "window * F_obs > 2^18 floats unsupported");
  if (c->n_positions <= 0 || c->n_positions > GTE_MAX_POSITIONS)
    return fail(GTE_ERR_INVALID, "1..%d positions required", GTE_MAX_POSITIONS);
  if (c->initial_position_index < -1 || c->initial_position_index >= c->n_positions)
    return fail(GTE_ERR_INVALID,
                "The 'initial_position' parameter must be 'random' or a position mentionned "
                "in the 'position' parameter.");  // environments.py:106
  if (c->max_episode_duration < 0 || c->max_episode_duration == 1)
    return fail(GTE_ERR_INVALID, "max_episode_duration must be 0 ('max') or >= 2");
  if (!(c->portfolio_initial_value > 0.0))
    return fail(GTE_ERR_INVALID, "portfolio_initial_value must be > 0");
  if (c->reward_kind < 0 || c->reward_kind > GTE_REWARD_CLIPPED_LOG_RETURN)
    return fail(GTE_ERR_INVALID, "reward_kind %d unknown", c->reward_kind);
  if (c->autoreset < 0 || c->autoreset > GTE_AUTORESET_SAME_STEP)
    return fail(GTE_ERR_INVALID, "autoreset %d unknown", c->autoreset);
  if (c->episodes_between_dataset_switch < 1)
    return fail(GTE_ERR_INVALID, "episodes_between_dataset_switch must be >= 1");
  if (c->nontemporal_obs < 0 || c->nontemporal_obs > 3)
    return fail(GTE_ERR_INVALID, "nontemporal_obs must be 0 (plain), 1 (nt), 2 (sc1) or 3 (automatic)");
  if (c->log_steps < 0) return fail(GTE_ERR_INVALID, "log_steps must be >= 0");
  if (c->final_obs && c->autoreset != GTE_AUTORESET_SAME_STEP)
    return fail(GTE_ERR_INVALID, "final_obs needs autoreset = same-step");
  if (c->envs_per_wave < 0 || c->envs_per_wave > 64)
    return fail(GTE_ERR_INVALID, "envs_per_wave must be 0 (automatic) or 1..64");
  return GTE_OK;
}

static int require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(GTE_ERR_NO_DEVICE,
                "no HIP device available (%s): libgte has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "0 devices");
  if (device < 0 || device >= n)
    return fail(GTE_ERR_NO_DEVICE, "device %d out of range (have %d)", device, n);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GTE_ERR_NO_DEVICE, "device %d is %s; libgte is built for gfx950 (MI355X) only",
                device, prop.gcnArchName);
  return GTE_OK;
}

extern "C" {

int gte_abi_version(void) { return GTE_ABI_VERSION; }

const char* gte_last_error(void) { return g_err.c_str(); }

int gte_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gte_create(const gte_config* cfg, gte_env** out) {
  if (!out) return fail(GTE_ERR_INVALID, "out is NULL");
  *out = nullptr;
  TRY(validate(cfg));
  TRY(require_device(cfg->device));
  HIPCHK(hipSetDevice(cfg->device));
  gte_env* E = new (std::nothrow) gte_env();
  if (!E) return fail(GTE_ERR_OOM, "host allocation failed");
  E->cfg = *cfg;
  memset(&E->p, 0, sizeof(Params));
  memset(&E->owned, 0, sizeof(gte_outputs));
  Params& p = E->p;
  p.N = cfg->n_envs;
  p.D = cfg->n_datasets;
  p.Fs = cfg->n_static;
  p.nd = cfg->n_dyn;
  p.Fobs = p.Fs + p.nd;
  p.has_window = cfg->window > 0;
  p.W = p.has_window ? cfg->window : 1;
  p.P = cfg->n_positions;
  for (int i = 0; i < GTE_MAX_DYN; ++i) p.dyn_kind[i] = cfg->dyn_kind[i];
  p.init_pos_index = cfg->initial_position_index;
  p.max_dur = cfg->max_episode_duration;
  p.reward_kind = cfg->reward_kind;
  p.autoreset = cfg->autoreset;
  p.switch_every = cfg->episodes_between_dataset_switch;
  p.persist = cfg->dyn_persist ? 1 : 0;
  p.fees = cfg->trading_fees;
  p.rate = cfg->borrow_interest_rate;
  p.V0 = cfg->portfolio_initial_value;
  p.rp0 = cfg->reward_param0;
  p.rp1 = cfg->reward_param1;
  p.rp2 = cfg->reward_param2;
  p.seed = cfg->seed;
  p.env_id_base = cfg->env_id_base;

  int rc = GTE_OK;
  auto chk = [&](int r) { if (rc == GTE_OK) rc = r; };
  const size_t N = (size_t)p.N;
  chk(hipStreamCreateWithFlags(&E->own_stream, hipStreamNonBlocking) == hipSuccess
          ? GTE_OK : fail(GTE_ERR_HIP, "hipStreamCreate failed"));
  E->stream = E->own_stream;
  chk(hipEventCreate(&E->ev0) == hipSuccess && hipEventCreate(&E->ev1) == hipSuccess
          ? GTE_OK : fail(GTE_ERR_HIP, "hipEventCreate failed"));
  chk(dev_alloc(E, &p.rec, N));  // zero-filled: every counter starts at 0
  chk(dev_alloc(E, &E->soa.idx, N)); chk(dev_alloc(E, &E->soa.step, N));
  chk(dev_alloc(E, &E->soa.pos, N)); chk(dev_alloc(E, &E->soa.dsi, N));
  chk(dev_alloc(E, &E->soa.start, N)); chk(dev_alloc(E, &E->soa.episode, N));
  chk(dev_alloc(E, &E->soa.needs_reset, N));
  chk(dev_alloc(E, &E->soa.asset, N)); chk(dev_alloc(E, &E->soa.fiat, N));
  chk(dev_alloc(E, &E->soa.ia, N)); chk(dev_alloc(E, &E->soa.ifi, N));
  chk(dev_alloc(E, &E->soa.pv, N)); chk(dev_alloc(E, &E->soa.realpos, N));
  // (the library's own observation buffers are allocated on first use, ensure_owned_obs: a caller
  // that binds its own — the torch path — never pays for a second copy of [N, W, F_obs])
  if (cfg->final_obs) {
    chk(dev_alloc(E, &p.final_rec, N));
    chk(dev_alloc(E, &E->fsoa.idx, N)); chk(dev_alloc(E, &E->fsoa.step, N));
    chk(dev_alloc(E, &E->fsoa.pos, N)); chk(dev_alloc(E, &E->fsoa.dsi, N));
    chk(dev_alloc(E, &E->fsoa.start, N)); chk(dev_alloc(E, &E->fsoa.episode, N));
    chk(dev_alloc(E, &E->fsoa.needs_reset, N));
    chk(dev_alloc(E, &E->fsoa.asset, N)); chk(dev_alloc(E, &E->fsoa.fiat, N));
    chk(dev_alloc(E, &E->fsoa.ia, N)); chk(dev_alloc(E, &E->fsoa.ifi, N));
    chk(dev_alloc(E, &E->fsoa.pv, N)); chk(dev_alloc(E, &E->fsoa.realpos, N));
  }
  {  // reward f32 [N] | terminated u8 [N] | truncated u8 [N] in ONE buffer of 6N bytes: the
     // layout a sharded run all-gathers as it is (gte_allgather_returns), no packing kernel
    uint8_t* packed = nullptr;
    chk(dev_alloc(E, &packed, 6 * N + 16));
    E->owned.reward = (float*)packed;
    E->owned.terminated = packed ? packed + 4 * N : nullptr;
    E->owned.truncated = packed ? packed + 5 * N : nullptr;
  }
  chk(dev_alloc(E, &E->owned.reward64, N));
  chk(dev_alloc(E, &E->owned.term_count, 2)); chk(dev_alloc(E, &E->owned.term_ids, N));
  chk(dev_alloc(E, &E->d_actions, N)); chk(dev_alloc(E, &E->d_mask, N));
  chk(dev_alloc(E, &E->d_inj_idx, N)); chk(dev_alloc(E, &E->d_inj_pos, N));
  chk(dev_alloc(E, &E->d_inj_ds, N));
  if (cfg->log_steps > 0) {
    const size_t LN = (size_t)cfg->log_steps * N;
    chk(dev_alloc(E, &E->log.rows, LN));  // 80 B per (row, env)
  }
  double* d_pos = nullptr;
  chk(dev_alloc(E, &d_pos, GTE_MAX_POSITIONS));
  chk(dev_alloc(E, &E->d_ds, (size_t)p.D));
  if (rc == GTE_OK &&
      hipMemcpy(d_pos, cfg->positions, sizeof(double) * p.P, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(GTE_ERR_HIP, "copying positions failed");
  if (rc != GTE_OK) {
    std::string keep = g_err;
    gte_destroy(E);
    g_err = keep;
    return rc;
  }
  p.positions = d_pos;
  p.ds = E->d_ds;
  E->owned.obs_elems_per_env = (int64_t)p.W * p.Fobs;
  // the observation buffers come later (ensure_owned_obs / gte_bind_outputs); while the launch
  // geometry is worked out below, a placeholder stands for "terminal observations are kept"
  // (the LDS image has a FinalJob per env then, lds_bytes)
  p.final_obs = cfg->final_obs ? (float*)(uintptr_t)16 : nullptr;
  p.obs = nullptr; p.reward = E->owned.reward; p.reward64 = E->owned.reward64;
  p.terminated = E->owned.terminated; p.truncated = E->owned.truncated;
  E->term_base = E->owned.term_count; p.term_ids = E->owned.term_ids;
  E->h_ds.assign((size_t)p.D, DatasetDesc{nullptr, nullptr, nullptr, nullptr, 0});
  for (auto& v : E->ds_allocs) v.assign((size_t)p.D, nullptr);

  // observation store policy (gte_kernels.hip, store_out): while the observation buffer fits
  // the 256 MB Infinity Cache next to the feature table, sc1 stores keep it there (65 536 envs,
  // 168 MB: 42.5 us vs 45.8 us with nt); beyond that the stores are a pure stream and
  // non-temporal ones win (81 920 envs, 210 MB: 48 us vs 58 us; 262 144 envs: 162 us vs 261 us)
  E->store_auto = E->cfg.nontemporal_obs == 3;
  if (E->store_auto)
    E->cfg.nontemporal_obs = ((size_t)N * p.W * p.Fobs * sizeof(float) > ((size_t)190 << 20)) ? 1 : 2;

  // launch geometry: EPW environments per wavefront, 4 wavefronts per workgroup
  E->vec = (p.Fobs % 4 == 0) ? 4 : 1;
  const int64_t vpe = (int64_t)p.W * p.Fobs / E->vec;  // vectors per env
  int epw = cfg->envs_per_wave;
  if (epw == 0) {
    // windows of at least one wave instruction: 16 envs per wave (64 per workgroup, cooperative
    // phase A) measured best at every size tried; tiny windows may pack up to 64 per wave
    epw = (vpe >= 64) ? 16 : 64;
    // enough wavefronts to fill 256 CUs x 16 waves ...
    while (epw > 1 && ((int64_t)p.N + epw - 1) / epw < 4096) epw >>= 1;
    // ... but at least one full wave instruction (64 vectors) of copy work per wavefront
    // (small windows are latency-bound: 16 envs/wave with cooperative phase A measured
    // 5.4 us vs 7.5 us at 64 envs/wave on config 2, profiles/r01_tune_c2.log)
    while (epw < 64 && (int64_t)epw * vpe < 64) epw <<= 1;
    // Windowed shapes on the hot kernel.  Two regimes, both measured:
    //  * every workgroup resident at once, at least 16 waves on every CU: the launch ends when the
    //    BUSIEST CU is done (workgroups are dealt evenly: ceil(wgs / CUs) per CU), so take the
    //    envs-per-wave with the fewest envs on that CU; ties go to the bigger workgroup (fewer
    //    phase-A waves).  Config 3, us per step by envs on the busiest CU: 256 (16 per wave) 39.5,
    //    260 (13) 40.4, 264 (11) 40.5, 280 (14) 40.7, 288 (12) 42.0, 300 (15) 42.0; config 5:
    //    128 (8 per wave, 4 workgroups per CU) 36.3, 16 per wave on 2 per CU 43.5
    //    (profiles/r02_tune_epw.log, r02_waves_ab.log, r02_c5_sweep.log).
    //  * more workgroups than the chip holds (the observations stream to HBM): small workgroups —
    //    the phase A of those that start later hides behind the copies of those already running
    //    and the tail is one small workgroup.  Config 3 (profiles/r02_tune_epw_repeat.log, 3 passes
    //    each): 262 144 envs 16 per wave 156.4, 12: 154.7, 8: 149.2, 6: 143.8, 4: 142.6, 3: 145.5,
    //    2: 169.0; 131 072 envs 16: 86.7, 11: 83.3, 8: 80.1, 6: 79.0, 4: 78.2, 3: 79.8; 100 003 envs
    //    16: 74.5, 9: 62.1, 6: 60.0, 4: 61.6 -> about 640 vectors of copy work per wave (4 envs
    //    of 20x32), 960 below 120 000 envs.
    const bool hot_shape = E->vec == 4 && vpe >= 64 && p.nd > 0 && !p.persist && !cfg->final_obs &&
                           (E->cfg.nontemporal_obs == 1 || E->cfg.nontemporal_obs == 2) &&
                           !(cfg->kernel_variant & (1 | 2));
    if (hot_shape) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
        const int64_t n_cu = prop.multiProcessorCount;
        int some_per_cu = 0;  // (residency of any candidate: the occupancy queries work)
        int64_t fewest = 0;
        int one_round_epw = 0, one_round_per_cu = 0;
        for (int e = 64 / GTE_WAVES; e >= 1; --e) {
          if ((int64_t)e * vpe < 64) break;
          Params q = p;
          q.epw = e;  // registers AND the workgroup's LDS (which shrinks with e) bound residency
          const size_t smem = gte::lds_bytes(q, 1);
          const int per_cu = E->cfg.nontemporal_obs == 1 ? gte::hot_blocks_per_cu_nt(smem)
                                                          : gte::hot_blocks_per_cu(smem);
          if (per_cu <= 0) break;
          if (e <= 8 || !some_per_cu) some_per_cu = per_cu;
          const int64_t wgs = ((int64_t)p.N + GTE_WAVES * e - 1) / (GTE_WAVES * e);
          const int64_t busiest = (wgs + n_cu - 1) / n_cu;  // workgroups on the busiest CU
          if (getenv("GTE_DEBUG_GEOMETRY"))
            fprintf(stderr, "[gte] envs/wave %2d: LDS %5zu B, %d workgroups/CU resident, %lld workgroups, "
                            "%lld on the busiest CU\n", e, smem, per_cu, (long long)wgs, (long long)busiest);
          if (busiest <= per_cu && busiest * GTE_WAVES >= 16 && (fewest == 0 || busiest * e < fewest)) {
            fewest = busiest * e;
            one_round_epw = e;
            one_round_per_cu = per_cu;
          }
        }
        if (one_round_epw) {
          epw = one_round_epw;
          E->hot_per_cu = one_round_per_cu;
        } else if (some_per_cu) {
          E->hot_per_cu = some_per_cu;
          const int64_t target = (int64_t)p.N >= 120000 ? 640 : 960;
          int e = (int)((target + vpe / 2) / vpe);
          e = e < 1 ? 1 : (e > 64 / GTE_WAVES ? 64 / GTE_WAVES : e);
          while (e < 64 / GTE_WAVES && (int64_t)e * vpe < 64) ++e;
          epw = e;
          // Round 3: where the lean copy loop applies (whole passes of 4 wave instructions per wave,
          // gte_kernels.hip) it beats the small workgroups above — 262 144 envs: 4 per wave 149 us,
          // 8: 139.8, 16: 137.8; 131 072: 4: 80, 8: 74, 16: 71-78; 100 003: 6: 62, 8: 58.7, 16: 60-67
          // (profiles/r03_epw_hbm.log) — so take the smallest envs-per-wave the lean loop accepts, twice
          // that from 200 000 envs on.
          if (!(cfg->kernel_variant & 4096)) {
            int lean_e = 0;
            for (int c = 1; c <= 64 / GTE_WAVES; ++c)
              if (((int64_t)c * vpe) % 256 == 0 && (int64_t)c * p.W <= 512) { lean_e = c; break; }
            if (lean_e) {
              int c = lean_e;
              while (c * 2 <= 64 / GTE_WAVES && (int64_t)c * vpe < 1280 && (int64_t)c * 2 * p.W <= 512) c *= 2;
              if ((int64_t)p.N >= 200000 && c * 2 <= 64 / GTE_WAVES && (int64_t)c * 2 * p.W <= 512) c *= 2;
              epw = c;
            }
          }
        }
      }
    }
  }
  while (epw > 1 && (int64_t)epw * vpe > (1 << 20)) epw >>= 1;  // keeps the index math in range
  // the LDS-staged dynamic columns must fit comfortably: shrink the workgroup's envs
  while (epw > 1 && (int64_t)epw * GTE_WAVES * p.W * (p.nd ? p.nd : 1) * 4 > 32 * 1024) epw >>= 1;
  p.epw = epw;
  p.debug = cfg->debug_flags;
  E->coop = (epw * GTE_WAVES <= 64) && !(cfg->kernel_variant & 1);
  E->stage = (p.nd > 0 && gte::lds_bytes(p, 1) <= 48 * 1024 && !(cfg->kernel_variant & 2))
                 ? (p.persist ? 2 : 1) : 0;
  // the lean copy loop (gte_kernels.hip): 16-byte vectors with the raw rings staged in LDS
  p.lean_rows = (E->vec == 4 && E->stage == 1 && !p.persist && !(cfg->kernel_variant & 4096)) ? 1 : 0;
  p.hot_lds = (cfg->kernel_variant & 8192) ? 0 : 1;  // (8192: A/B, the stepping lane stores its record itself)
  const int64_t waves = ((int64_t)p.N + epw - 1) / epw;
  E->threads = 64 * GTE_WAVES;
  E->blocks = (int)((waves + GTE_WAVES - 1) / GTE_WAVES);
  // L2-affinity order: only worth it when every XCD gets several workgroups and the
  // windows are big enough to be bandwidth-bound
  {
    // Default re-sort period: the order decays as envs jump to new start rows, i.e. with the
    // reset rate, about 1 / max_episode_duration of the envs per step once the episodes are out
    // of phase.  Measured at duration 500, episodes staggered (profiles/r02_tune_affinity_period.log):
    // every 128 steps 42.4 us per step, 64: 41.3, 32: 40.6, 16: 40.5, 8: 41.6 (a re-sort was four
    // small launches then, ~25 us) -> re-sort after ~6 % of the envs have moved; 128 when episodes
    // only end by the drawdown rule or at the end of the data.
    int auto_period = 128;
    if (p.max_dur > 0) {
      auto_period = p.max_dur / 16;
      auto_period = auto_period < 8 ? 8 : (auto_period > 128 ? 128 : auto_period);
    }
    const int period = cfg->affinity_period == 0 ? auto_period : cfg->affinity_period;
    const int EPB = epw * GTE_WAVES;
    const int n_wg = (p.N + EPB - 1) / EPB;
    if (period > 0 && n_wg >= 64 && vpe * E->vec * 4 >= 512) {
      // slots in XCD-major order: workgroup b runs on XCD b % 8 (round-robin dispatch)
      std::vector<int32_t> slot_of_rank;
      slot_of_rank.reserve((size_t)p.N);
      for (int x = 0; x < 8; ++x)
        for (int b = x; b < n_wg; b += 8)
          for (int sl = b * EPB; sl < (b + 1) * EPB && sl < p.N; ++sl) slot_of_rank.push_back(sl);
      int nb = 8192 / p.D;
      nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
      if (const char* o = getenv("GTE_AFFINITY_BINS")) {  // tuning: total bins of the counting sort
        nb = atoi(o) / p.D;
        nb = nb < 1 ? 1 : nb;
      }
      E->n_bins_per_ds = nb;
      int rc2 = GTE_OK;
      if (rc2 == GTE_OK) rc2 = dev_alloc(E, &E->d_perm, N, false);
      if (rc2 == GTE_OK) rc2 = dev_alloc(E, &E->d_slot_of_rank, N, false);
      if (rc2 == GTE_OK) rc2 = dev_alloc(E, &E->d_bins, (size_t)2 * p.D * nb);  // histogram (zero-filled) + cursors
      if (rc2 == GTE_OK && hipMemcpy(E->d_slot_of_rank, slot_of_rank.data(), sizeof(int32_t) * N,
                                     hipMemcpyHostToDevice) != hipSuccess)
        rc2 = fail(GTE_ERR_HIP, "copying slot_of_rank failed");
      if (rc2 != GTE_OK) { std::string keep = g_err; gte_destroy(E); g_err = keep; return rc2; }
      E->affinity_period = period;
    }
  }
  // the zero-fills above ran on the null stream; the env's stream is non-blocking
  if (hipDeviceSynchronize() != hipSuccess) {
    gte_destroy(E);
    return fail(GTE_ERR_HIP, "hipDeviceSynchronize failed after allocation");
  }
  p.final_obs = nullptr;  // (placeholder above)
  *out = E;
  return GTE_OK;
}

int gte_upload_dataset(gte_env* E, int32_t d, const float* feat, const double* close,
                       const double* high, const double* low, int64_t T) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  const Params& p = E->p;
  if (d < 0 || d >= p.D) return fail(GTE_ERR_INVALID, "dataset index %d out of range", d);
  if (!feat || !close) return fail(GTE_ERR_INVALID, "feat and close are required");
  if (T <= 0 || T > 0x7FFFFFF0ll) return fail(GTE_ERR_INVALID, "T out of range");
  // the conditions under which the reference's reset/step are defined (:171-177)
  const int64_t idx0 = p.has_window ? p.W - 1 : 0;
  if (T < idx0 + 2) return fail(GTE_ERR_INVALID, "dataset of %lld rows is too short for windows=%d",
                                (long long)T, p.W);
  if (p.max_dur > 0 && T - p.max_dur - idx0 <= idx0)
    return fail(GTE_ERR_INVALID, "low >= high: %lld rows cannot host episodes of %d steps",
                (long long)T, p.max_dur);
  if (E->finalized && p.persist && T > p.depth)
    return fail(GTE_ERR_STATE, "dyn_persist: cannot upload a longer dataset after the first reset");
  // After a reset envs may sit anywhere in the old table: a shorter replacement would leave
  // their next step reading rows past the new allocation.  (The reference's own _set_df is only
  // ever followed by reset(); MultiDatasetTradingEnv's drop-in builds a fresh batch per dataset.)
  if (E->was_reset && T < E->h_ds[d].T)
    return fail(GTE_ERR_STATE, "dataset %d: cannot replace %lld rows by %lld after gte_reset "
                "(running environments may be past the new end); create a new env instead",
                d, (long long)E->h_ds[d].T, (long long)T);
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(hipStreamSynchronize(E->stream));
  const void* srcs[4] = {feat, close, high, low};
  const size_t bytes[4] = {sizeof(float) * (size_t)T * p.Fobs, sizeof(double) * (size_t)T,
                           sizeof(double) * (size_t)T, sizeof(double) * (size_t)T};
  // new buffers first; the old ones stay in the descriptor table (and valid) until the new
  // table has been published, so a failure on the way leaves the env exactly as it was
  void* dev[4] = {nullptr, nullptr, nullptr, nullptr};
  auto undo = [&]() { for (void* q : dev) if (q) (void)hipFree(q); };
  for (int k = 0; k < 4; ++k) {
    if (!srcs[k]) continue;
    hipError_t e = hipMalloc(&dev[k], bytes[k]);
    if (e == hipSuccess && getenv("GTE_DEBUG_ALLOC"))
      fprintf(stderr, "[gte] dataset %d column %d at %p, %zu bytes\n", d, k, dev[k], bytes[k]);
    if (e == hipSuccess) e = hipMemcpy(dev[k], srcs[k], bytes[k], hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      undo();
      return fail(e == hipErrorOutOfMemory ? GTE_ERR_OOM : GTE_ERR_HIP, "uploading dataset %d: %s", d,
                  hipGetErrorString(e));
    }
  }
  const DatasetDesc before = E->h_ds[d];
  E->h_ds[d] = DatasetDesc{(const float*)dev[0], (const double*)dev[1], (const double*)dev[2],
                           (const double*)dev[3], T};
  const hipError_t e = hipMemcpy(E->d_ds, E->h_ds.data(), sizeof(DatasetDesc) * p.D, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    E->h_ds[d] = before;
    undo();
    return fail(GTE_ERR_HIP, "publishing the descriptor of dataset %d: %s", d, hipGetErrorString(e));
  }
  for (int k = 0; k < 4; ++k) {
    if (E->ds_allocs[k][d]) (void)hipFree(E->ds_allocs[k][d]);
    E->ds_allocs[k][d] = dev[k];
  }
  return GTE_OK;
}

static int finalize(gte_env* E) {
  if (E->finalized) return GTE_OK;
  Params& p = E->p;
  int64_t maxT = 0;
  for (int d = 0; d < p.D; ++d) {
    if (E->h_ds[d].T <= 0) return fail(GTE_ERR_STATE, "dataset %d was never uploaded", d);
    if (E->h_ds[d].T > maxT) maxT = E->h_ds[d].T;
  }
  p.depth = p.persist ? maxT : p.W;
  // The L2-affinity order pays when each XCD's 4 MiB L2 can hold its 1/8 share of the feature
  // tables (config 3: 12.8 MB).  Tables far beyond the 32 MiB of aggregate L2 (config 5: 1.6 GB
  // per GPU, L2 hit rate 0.16 either way) only pay for the re-sorts and for scattered
  // observation stores: 39.4 us per step with the order, 36.8 without
  // (profiles/r02_c5_sweep.log).  Automatic (affinity_period = 0) turns it off there.
  if (E->affinity_period > 0 && E->cfg.affinity_period == 0) {
    size_t table_bytes = 0;
    for (int d = 0; d < p.D; ++d) table_bytes += (size_t)E->h_ds[d].T * p.Fobs * sizeof(float);
    if (table_bytes > ((size_t)64 << 20)) E->affinity_period = 0;
  }
  TRY(dev_alloc(E, &p.ring, (size_t)p.N * p.depth * (p.nd ? p.nd : 1)));
  HIPCHK(hipDeviceSynchronize());
  E->finalized = true;
  return GTE_OK;
}

// injected draws come from the caller: refuse values the kernel would use as
// out-of-range row / position / dataset indices
static int check_injection(const gte_env* E, size_t count, const int32_t* idx, const int32_t* pos,
                           const int32_t* ds) {
  const Params& p = E->p;
  int64_t maxT = 0, minT = INT64_MAX;
  for (int d = 0; d < p.D; ++d) {
    if (E->h_ds[d].T > maxT) maxT = E->h_ds[d].T;
    if (E->h_ds[d].T > 0 && E->h_ds[d].T < minT) minT = E->h_ds[d].T;
  }
  const int64_t idx0 = p.has_window ? p.W - 1 : 0;
  for (size_t i = 0; i < count; ++i) {
    if (pos && pos[i] >= p.P) return fail(GTE_ERR_INVALID, "injected position index %d >= %d", pos[i], p.P);
    if (ds && ds[i] >= p.D) return fail(GTE_ERR_INVALID, "injected dataset %d >= %d", ds[i], p.D);
    if (idx && idx[i] >= 0) {
      // an env needs at least one row after its start row; its dataset is known only when
      // it is injected too, otherwise the shortest dataset bounds it
      const int64_t T = (ds && ds[i] >= 0) ? E->h_ds[ds[i]].T : minT;
      if (idx[i] < idx0 || idx[i] > T - 2)
        return fail(GTE_ERR_INVALID, "injected start row %d outside [%lld, %lld]", idx[i],
                    (long long)idx0, (long long)(T - 2));
    }
  }
  return GTE_OK;
}

// append one trajectory row per env (after a reset or a step)
static int append_log(gte_env* E) {
  if (E->cfg.log_steps <= 0) return GTE_OK;
  const Params& p = E->p;
  const int64_t row_base = (E->log_rows % E->cfg.log_steps) * (int64_t)p.N;
  HIPCHK(gte::launch_log(p.rec, p.reward64, p.terminated, p.truncated, p.N, row_base, E->log, E->stream));
  E->log_rows += 1;
  return GTE_OK;
}

static int stage(gte_env* E, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, E->stream));
  return GTE_OK;
}

// The library's own observation buffers, on first need: nobody bound one (gte_bind_outputs) by
// the first gte_reset / gte_get_outputs, or a binding was withdrawn.
static int ensure_owned_obs(gte_env* E) {
  Params& p = E->p;
  const size_t elems = (size_t)p.N * (size_t)p.W * (size_t)p.Fobs;
  bool fresh = false;
  if (!p.obs) {
    if (!E->owned.obs) { TRY(dev_alloc(E, &E->owned.obs, elems)); fresh = true; }
    p.obs = E->owned.obs;
  }
  if (E->cfg.final_obs && !p.final_obs) {
    if (!E->owned.final_obs) { TRY(dev_alloc(E, &E->owned.final_obs, elems)); fresh = true; }
    p.final_obs = E->owned.final_obs;
  }
  if (fresh) HIPCHK(hipDeviceSynchronize());  // zero-filled on the null stream; E->stream is non-blocking
  return GTE_OK;
}

int gte_reset(gte_env* E, const uint8_t* mask, const int32_t* inj_idx,
              const int32_t* inj_pos_index, const int32_t* inj_dataset) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  HIPCHK(hipSetDevice(E->cfg.device));
  TRY(ensure_owned_obs(E));
  TRY(finalize(E));
  TRY(check_injection(E, (size_t)E->p.N, inj_idx, inj_pos_index, inj_dataset));
  Params p = E->p;
  const size_t N = (size_t)p.N;
  p.mask = nullptr; p.inj_idx = p.inj_pos = p.inj_ds = nullptr;
  if (mask) { TRY(stage(E, E->d_mask, mask, N)); p.mask = E->d_mask; }
  if (inj_idx) { TRY(stage(E, E->d_inj_idx, inj_idx, 4 * N)); p.inj_idx = E->d_inj_idx; }
  if (inj_pos_index) { TRY(stage(E, E->d_inj_pos, inj_pos_index, 4 * N)); p.inj_pos = E->d_inj_pos; }
  if (inj_dataset) { TRY(stage(E, E->d_inj_ds, inj_dataset, 4 * N)); p.inj_ds = E->d_inj_ds; }
  HIPCHK(hipMemsetAsync(E->term_base, 0, 2 * sizeof(int32_t), E->stream));
  E->term_slot = 0;
  p.term_count = E->term_base;
  p.term_count_next = E->term_base + 1;
  HIPCHK(gte::launch_reset(p, E->vec, E->cfg.nontemporal_obs, E->coop, E->stage, E->blocks,
                           E->threads, E->stream));
  TRY(append_log(E));
  if (E->affinity_period > 0) {  // new start rows: re-sort the processing order
    HIPCHK(gte::launch_affinity_rebuild(E->p, E->d_bins, E->n_bins_per_ds, E->d_slot_of_rank,
                                        E->d_perm, E->stream));
    E->p.perm = E->d_perm;
    E->steps_since_rebuild = 0;
  }
  // host staging buffers may be reused by the caller right away: pageable copies above
  // are complete on return, but keep the contract simple and explicit
  HIPCHK(hipStreamSynchronize(E->stream));
  E->was_reset = true;
  return GTE_OK;
}

int gte_set_autoreset_injection(gte_env* E, int32_t n, const int32_t* inj_idx,
                                const int32_t* inj_pos_index, const int32_t* inj_dataset) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (n < 0) return fail(GTE_ERR_INVALID, "n_episodes must be >= 0");
  for (int d = 0; d < E->p.D; ++d)
    if (E->h_ds[d].T <= 0) return fail(GTE_ERR_STATE, "upload every dataset before queueing draws");
  TRY(check_injection(E, (size_t)E->p.N * (size_t)n, inj_idx, inj_pos_index, inj_dataset));
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(hipStreamSynchronize(E->stream));
  Params& p = E->p;
  const size_t count = (size_t)p.N * (size_t)n;
  // the queues of an earlier call are released here (the stream is idle: nothing reads them)
  auto put = [&](int32_t** slot, const int32_t* src, const int32_t** param) -> int {
    *param = nullptr;
    if (*slot) { (void)hipFree(*slot); *slot = nullptr; }
    if (!src || n == 0) return GTE_OK;
    void* dev = nullptr;
    HIPCHK(hipMalloc(&dev, sizeof(int32_t) * count));
    *slot = (int32_t*)dev;  // owned by the env from here on (freed above or in gte_destroy)
    HIPCHK(hipMemcpy(dev, src, sizeof(int32_t) * count, hipMemcpyHostToDevice));
    *param = *slot;
    return GTE_OK;
  };
  TRY(put(&E->d_q_idx, inj_idx, &p.q_idx));
  TRY(put(&E->d_q_pos, inj_pos_index, &p.q_pos));
  TRY(put(&E->d_q_ds, inj_dataset, &p.q_ds));
  p.q_n = n;
  HIPCHK(gte::launch_rewind_queue(p.rec, p.N, E->stream));
  HIPCHK(hipStreamSynchronize(E->stream));
  HIPCHK(hipDeviceSynchronize());
  return GTE_OK;
}

int gte_step(gte_env* E, const int32_t* actions, int32_t actions_on_device) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_step before gte_reset");
  if (!actions) return fail(GTE_ERR_INVALID, "actions is NULL");
  // Stream capture (hipStreamBeginCapture by whoever owns the stream — e.g. torch.cuda.graph around
  // policy + step): everything a step enqueues is capturable — kernels and one memset of the
  // re-sort — provided nothing comes from pageable host memory, and the host-side bookkeeping a
  // replay cannot repeat stays consistent: the two-slot terminal counter alternates per launch, so
  // a graph must hold an EVEN number of steps and be replayed from the slot it was captured at
  // (gte_get_outputs().term_slot; StepGraph in step_graph.py checks both); the trajectory log's
  // row index is host state, so logged envs cannot be captured.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(E->stream, &capture) != hipSuccess) { (void)hipGetLastError(); capture = hipStreamCaptureStatusNone; }
  if (capture == hipStreamCaptureStatusActive) {
    if (!actions_on_device)
      return fail(GTE_ERR_STATE, "gte_step on a capturing stream needs device-resident actions "
                                 "(a copy from pageable host memory cannot be captured)");
    if (E->cfg.log_steps > 0)
      return fail(GTE_ERR_STATE, "gte_step on a capturing stream: the trajectory log's row index is "
                                 "host state a replay would not advance (log_steps must be 0)");
  }
  if (E->affinity_period > 0 && ++E->steps_since_rebuild >= E->affinity_period) {
    // envs drift one row per step and ~1/duration of them jump at a reset: re-sort now and
    // then (3 tiny launches, stream-ordered between two steps)
    HIPCHK(gte::launch_affinity_rebuild(E->p, E->d_bins, E->n_bins_per_ds, E->d_slot_of_rank,
                                        E->d_perm, E->stream));
    E->steps_since_rebuild = 0;
  }
  Params p = E->p;
  if (actions_on_device) {
    p.actions = actions;
  } else {
    TRY(stage(E, E->d_actions, actions, sizeof(int32_t) * (size_t)p.N));
    p.actions = E->d_actions;
  }
  // two-slot terminal counter: this launch adds to one slot (cleared by the previous
  // launch or by gte_reset) and clears the other, so no memset sits between steps
  E->term_slot ^= 1;
  p.term_count = E->term_base + E->term_slot;
  p.term_count_next = E->term_base + (E->term_slot ^ 1);
  // With a trajectory log the step kernel writes the row itself (shared-TU instantiation): the lane
  // that stepped the env puts its 80-byte record into LDS and the copy waves write it out, five
  // lanes per env.  At the config-3 shape, us per step: 38.5 against 43.2 with the separate
  // gte_log_kernel launch (and 37.5 without a log; profiles/r03_log_ab.log).  (Rounds 1-2 kept the
  // log as twelve [L, N] columns: twelve scattered stores per env from the stepping lane, which
  // beyond 16 384 envs lost to the separate launch.)
  // kernel_variant bit 1024 keeps the separate launch (A/B), 2048 = the default now.
  const bool fused_log = E->cfg.log_steps > 0 && !(E->cfg.kernel_variant & 1024);
  if (fused_log) {
    p.log = E->log;
    p.log_row_base = (E->log_rows % E->cfg.log_steps) * (int64_t)p.N;
  }
  // (hot_tu_covers: the isolated instantiations have no terminal records and no trajectory row)
  const bool hot = E->vec == 4 && E->coop && E->stage == 1 && !(E->cfg.kernel_variant & 64) &&
                   gte::hot_tu_covers(p);
  if (hot && E->cfg.nontemporal_obs == 2)
    HIPCHK(gte::launch_step_hot(p, E->blocks, E->threads, gte::lds_bytes(p, E->stage), E->stream));
  else if (hot && E->cfg.nontemporal_obs == 1)
    HIPCHK(gte::launch_step_hot_nt(p, E->blocks, E->threads, gte::lds_bytes(p, E->stage), E->stream));
  else
    HIPCHK(gte::launch_step(p, E->vec, E->cfg.nontemporal_obs, E->coop, E->stage, E->blocks,
                            E->threads, E->stream));
  if (fused_log) E->log_rows += 1;
  else TRY(append_log(E));
  return GTE_OK;
}

static bool cfg_is_auto_epw(const gte_env* E) { return E->cfg.envs_per_wave == 0; }

int gte_rollout(gte_env* E, const int32_t* actions, int32_t n_steps, const gte_rollout_bufs* b) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_rollout before gte_reset");
  if (!actions) return fail(GTE_ERR_INVALID, "actions is NULL");
  if (n_steps < 1) return fail(GTE_ERR_INVALID, "n_steps must be >= 1");
  static const gte_rollout_bufs none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!b) b = &none;
  const size_t N = (size_t)E->p.N;
  const size_t V = (size_t)E->p.W * (size_t)E->p.Fobs;
  if (b->obs && ((uintptr_t)b->obs & 15)) return fail(GTE_ERR_INVALID, "obs must be 16-byte aligned");
  HIPCHK(hipSetDevice(E->cfg.device));
  // kernel_variant 128 = never fused (A/B and tests of the per-launch path)
  const bool fused = E->vec == 4 && E->coop && E->stage == 1 && !E->cfg.final_obs && !E->p.persist &&
                     E->cfg.log_steps == 0 && !(E->cfg.kernel_variant & 128);
  // per-step observation rows are written once and not read back by the kernels: a stream
  struct RestoreStorePolicy {  // whatever path leaves this function, the env's policy returns
    int32_t& slot;
    int32_t value;
    ~RestoreStorePolicy() { slot = value; }
  } restore_store_policy{E->cfg.nontemporal_obs, E->cfg.nontemporal_obs};
  if (b->obs && E->store_auto) E->cfg.nontemporal_obs = 1;
  // one step as its own launch, writing row k of every per-step buffer (what the unfused path
  // does for every step, and the backtest path for its last one)
  const Params keep = E->p;
  auto step_row = [&](int32_t k) -> int {
    if (b->obs) E->p.obs = b->obs + (size_t)k * N * V;
    if (b->reward) E->p.reward = b->reward + (size_t)k * N;
    if (b->reward64) E->p.reward64 = b->reward64 + (size_t)k * N;
    if (b->terminated) E->p.terminated = b->terminated + (size_t)k * N;
    if (b->truncated) E->p.truncated = b->truncated + (size_t)k * N;
    int rc = gte_step(E, actions + (size_t)k * N, 1);
    if (rc == GTE_OK && b->valuation) {
      hipError_t e = gte::launch_extract_state(E->p.rec, E->p.N, E->soa, E->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync(b->valuation + (size_t)k * N, E->soa.pv, 8 * N, hipMemcpyDeviceToDevice,
                           E->stream);
      if (e != hipSuccess) rc = fail(GTE_ERR_HIP, "rollout: %s", hipGetErrorString(e));
    }
    E->p.obs = keep.obs; E->p.reward = keep.reward; E->p.reward64 = keep.reward64;
    E->p.terminated = keep.terminated; E->p.truncated = keep.truncated;
    return rc;
  };
  auto count_steps = [&](int32_t n) -> int {  // the processing order ages with every fused step too
    if (E->affinity_period > 0) {
      E->steps_since_rebuild += n;
      if (E->steps_since_rebuild >= E->affinity_period) {
        HIPCHK(gte::launch_affinity_rebuild(E->p, E->d_bins, E->n_bins_per_ds, E->d_slot_of_rank,
                                            E->d_perm, E->stream));
        E->steps_since_rebuild = 0;
      }
    }
    return GTE_OK;
  };
  if (!fused) {
    // same results, one launch per step
    for (int32_t k = 0; k < n_steps; ++k) TRY(step_row(k));
  } else if (!b->obs) {
    // Backtest: no observation is kept but the last one.  n_steps - 1 steps of pure state machine
    // from registers (gte_rollout_state_kernel), then the last step as an ordinary launch, which
    // also produces the observation and the terminal list.
    if (n_steps > 1) {
      TRY(count_steps(n_steps - 1));
      gte::RolloutArgs r = {actions, n_steps - 1, nullptr, b->reward, b->reward64, b->terminated,
                            b->truncated, b->valuation, 0, 0, nullptr};
      // identity order: with no window to gather, the L2-affinity order would only scatter the
      // per-env loads and stores (actions, rewards, flags) that are coalesced in env order
      Params ps = E->p;
      ps.perm = nullptr;
      // envs per wavefront: full waves once every SIMD has one (65 536 envs: 5.8 us per step
      // with 64, 6.0 with 32, 9.2 with 16 — throughput of the scattered record / ring stores);
      // small batches are a latency chain and two half-filled waves overlap better (4 096 envs:
      // 3.5 us with 32, 3.9 with 64) — profiles/r02_state_epw.log
      const int sepw = (E->p.N >= 64 * 1024) ? 64 : 32;
      const hipError_t le = gte::launch_rollout_state(ps, r, n_steps - 1, sepw, E->stream);
      if (le != hipSuccess) return fail(GTE_ERR_HIP, "rollout launch: %s", hipGetErrorString(le));
    }
    TRY(step_row(n_steps - 1));
  } else {
    TRY(count_steps(n_steps));
    Params p = E->p;
    // Window-resident kernel (gte_rollout.hip): each env's W-1 older rows stay in LDS for the
    // whole launch.  Geometry: E envs per workgroup such that (a) the newest rows fit the owner
    // threads, (b) as many envs as possible are resident per CU and (c) the workgroups fill a
    // whole number of rounds (each workgroup runs all K steps, so a part-filled last round costs
    // a full one): minimise rounds x max(phase A chain, the CU's observation bytes per step).
    const int nt = E->cfg.nontemporal_obs;
    if (E->resident_epb[nt] == 0) {
      E->resident_epb[nt] = -1;
      hipDeviceProp_t prop;
      const int64_t FV = p.Fobs / 4;
      if (p.W >= 2 && !(E->cfg.kernel_variant & 256) &&
          hipGetDeviceProperties(&prop, E->cfg.device) == hipSuccess) {
        double best = 0.0;
        const char* force = getenv("GTE_RESIDENT_EPB");  // tuning: envs per group, no search
        for (int e = 64; e >= 1; --e) {
          if (force && atoi(force) != e) continue;
          if ((int64_t)e * FV > 2 * 192) continue;  // RES_NEW * RES_OWNERS newest-row vectors
          if (gte::resident_lds_bytes(p, e) > (size_t)160 * 1024) continue;
          const int per_cu = gte::resident_blocks_per_cu(p, e, nt);
          if (per_cu <= 0) continue;
          const int64_t slots = (int64_t)per_cu * prop.multiProcessorCount;
          const int64_t wgs = ((int64_t)p.N + e - 1) / e;
          // the launch is a work queue over the groups: "rounds" is a real number, at least one
          const double rounds = wgs > slots ? (double)wgs / (double)slots : 1.0;
          const double live = (double)(wgs < slots ? (wgs + prop.multiProcessorCount - 1) / prop.multiProcessorCount
                                                   : per_cu);  // workgroups sharing a CU
          const double us_bw = live * e * (double)p.W * p.Fobs * 4.0 / 22.0e3;  // ~5.6 TB/s over 256 CUs
          // a workgroup's barriers and its state wave's latency are hidden by the OTHER workgroups
          // of its CU: prefer four of them
          // (measured at config 3, profiles/r02_resident_epb.log: 4 per CU 27.7 us per step, 3: 28.9,
          // 2: 30.3, 1: 47.2)
          static const double kAlone[5] = {1.7, 1.7, 1.10, 1.05, 1.0};
          const double alone = kAlone[per_cu < 4 ? per_cu : 4];
          const double cost = rounds * (us_bw > 3.0 ? us_bw : 3.0) * alone;
          if (getenv("GTE_DEBUG_GEOMETRY"))
            fprintf(stderr, "[gte] resident rollout, %2d envs/workgroup: LDS %6zu B, %d workgroups/CU, "
                            "%lld groups, %.2f round(s), cost %.1f\n", e, gte::resident_lds_bytes(p, e),
                    per_cu, (long long)wgs, rounds, cost);
          if (best == 0.0 || cost < best * 0.999) {
            best = cost;
            E->resident_epb[nt] = e;
            E->resident_slots[nt] = (int)slots;
          }
        }
      }
    }
    E->term_slot ^= 1;
    p.term_count = E->term_base + E->term_slot;
    p.term_count_next = E->term_base + (E->term_slot ^ 1);
    if (E->resident_epb[nt] > 0) {
      // identity processing order: the L2-affinity order exists for the table reads of the
      // per-step gather; here one row per env and step is read, and consecutive envs make each
      // workgroup's observation stores one contiguous run
      p.perm = nullptr;
      if (!E->d_group_counter) {
        TRY(dev_alloc(E, &E->d_group_counter, 4));
        HIPCHK(hipDeviceSynchronize());  // (the zero-fill ran on the null stream)
      }
      HIPCHK(hipMemsetAsync(E->d_group_counter, 0, sizeof(int32_t), E->stream));
      const int epb = E->resident_epb[nt];
      const int n_groups = (p.N + epb - 1) / epb;
      gte::RolloutArgs r = {actions, n_steps, b->obs, b->reward, b->reward64, b->terminated,
                            b->truncated, b->valuation, epb, n_groups, E->d_group_counter};
      const int blocks = n_groups < E->resident_slots[nt] ? n_groups : E->resident_slots[nt];
      const hipError_t le = gte::launch_rollout_resident(p, r, nt, blocks, E->stream);
      if (le != hipSuccess) return fail(GTE_ERR_HIP, "rollout launch: %s", hipGetErrorString(le));
    } else {
      if (E->rollout_epw == 0) {
        // The gather-per-step rollout kernel is a long-running loop: a workgroup that is not
        // resident from the start runs all K steps after the others have finished.  It needs more
        // registers than the step kernel, so it gets its own workgroup size, the smallest that
        // keeps every workgroup resident.
        E->rollout_epw = p.epw;
        hipDeviceProp_t prop;
        if (cfg_is_auto_epw(E) && hipGetDeviceProperties(&prop, E->cfg.device) == hipSuccess) {
          for (int e = 1; e <= 16; ++e) {
            Params q = p;
            q.epw = e;
            if ((int64_t)e * p.W * p.Fobs / 4 < 64) continue;
            const int per_cu = gte::rollout_blocks_per_cu(q, E->cfg.nontemporal_obs);
            const int64_t wgs = ((int64_t)p.N + 4 * e - 1) / (4 * e);
            if (per_cu > 0 && wgs <= (int64_t)per_cu * prop.multiProcessorCount) { E->rollout_epw = e; break; }
            if (e == 16) E->rollout_epw = 16;  // more envs than one round holds: biggest workgroups
          }
        }
      }
      p.epw = E->rollout_epw;
      const int r_blocks = (int)((((int64_t)p.N + p.epw - 1) / p.epw + 3) / 4);
      gte::RolloutArgs r = {actions, n_steps, b->obs, b->reward, b->reward64, b->terminated,
                            b->truncated, b->valuation, 0, 0, nullptr};
      const hipError_t le = gte::launch_rollout(p, r, E->cfg.nontemporal_obs, r_blocks, E->threads,
                                                E->stream);
      if (le != hipSuccess) return fail(GTE_ERR_HIP, "rollout launch: %s", hipGetErrorString(le));
    }
  }
  // the env's own return buffers describe the last step
  const size_t last = (size_t)(n_steps - 1) * N;
  if (b->reward) HIPCHK(hipMemcpyAsync(E->p.reward, b->reward + last, 4 * N, hipMemcpyDeviceToDevice, E->stream));
  if (b->reward64) HIPCHK(hipMemcpyAsync(E->p.reward64, b->reward64 + last, 8 * N, hipMemcpyDeviceToDevice, E->stream));
  if (b->terminated) HIPCHK(hipMemcpyAsync(E->p.terminated, b->terminated + last, N, hipMemcpyDeviceToDevice, E->stream));
  if (b->truncated) HIPCHK(hipMemcpyAsync(E->p.truncated, b->truncated + last, N, hipMemcpyDeviceToDevice, E->stream));
  return GTE_OK;
}

int gte_add_limit_orders(gte_env* E, const int32_t* pos_index, const double* limit,
                         const uint8_t* persistent) {
  if (!E || !pos_index || !limit) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_add_limit_orders before gte_reset");
  Params& p = E->p;
  for (int d = 0; d < p.D; ++d)
    if (!E->h_ds[d].high || !E->h_ds[d].low)
      return fail(GTE_ERR_STATE, "limit orders need high/low columns in dataset %d", d);
  const size_t N = (size_t)p.N;
  for (size_t i = 0; i < N; ++i)
    if (pos_index[i] >= p.P) return fail(GTE_ERR_INVALID, "pos_index[%zu] out of range", i);
  HIPCHK(hipSetDevice(E->cfg.device));
  if (!p.lo_pos) {  // first use: allocate the order tables (the counts live in EnvRec)
    int32_t* lo_pos = nullptr;
    TRY(dev_alloc(E, &lo_pos, N * p.P));
    TRY(dev_alloc(E, &p.lo_limit, N * p.P));
    TRY(dev_alloc(E, &p.lo_persist, N * p.P));
    TRY(dev_alloc(E, &E->d_lo_limit_in, N));
    TRY(dev_alloc(E, &E->d_lo_persist_in, N));
    HIPCHK(hipDeviceSynchronize());
    p.lo_pos = lo_pos;
  }
  TRY(stage(E, E->d_inj_pos, pos_index, 4 * N));
  TRY(stage(E, E->d_lo_limit_in, limit, 8 * N));
  if (persistent) TRY(stage(E, E->d_lo_persist_in, persistent, N));
  HIPCHK(gte::launch_add_orders(p, E->d_inj_pos, E->d_lo_limit_in,
                                persistent ? E->d_lo_persist_in : nullptr, E->stream));
  HIPCHK(hipStreamSynchronize(E->stream));  // host arrays and staging buffers are free again
  return GTE_OK;
}

int gte_get_log(gte_env* E, gte_log_view* out) {
  if (!E || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  if (E->cfg.log_steps <= 0) return fail(GTE_ERR_STATE, "created with log_steps = 0");
  gte::LogRow* r = E->log.rows;  // the columns as strided views of the [L, N] rows
  out->idx = &r->idx; out->step = &r->step; out->position_index = &r->pos;
  out->dataset_index = &r->dsi; out->portfolio_valuation = &r->pv;
  out->real_position = &r->realpos; out->reward = &r->reward; out->flags = &r->flags;
  out->rows = E->log_rows; out->L = E->cfg.log_steps; out->N = E->p.N;
  out->asset = &r->asset; out->fiat = &r->fiat;
  out->interest_asset = &r->ia; out->interest_fiat = &r->ifi;
  out->env_stride = (int64_t)sizeof(gte::LogRow);
  out->row_stride = (int64_t)sizeof(gte::LogRow) * E->p.N;
  return GTE_OK;
}

// rows first .. first+n-1 (mod L) of ONE env, oldest first: at most two strided 2-D copies per array
static int pull_log_column(gte_env* E, int32_t env_id, int64_t first, int32_t n, void* host,
                           size_t offset, size_t elem) {
  if (!host) return GTE_OK;
  const int N = E->p.N, L = E->cfg.log_steps;
  const size_t pitch = sizeof(gte::LogRow) * (size_t)N;  // the same env, one row later
  const char* col = (const char*)E->log.rows + offset + sizeof(gte::LogRow) * (size_t)env_id;
  const int64_t run1 = (first + n <= L) ? n : (L - first);
  HIPCHK(hipMemcpy2D(host, elem, col + (size_t)first * pitch, pitch, elem, (size_t)run1,
                     hipMemcpyDeviceToHost));
  if (run1 < n)
    HIPCHK(hipMemcpy2D((char*)host + run1 * elem, elem, col, pitch, elem, (size_t)(n - run1),
                       hipMemcpyDeviceToHost));
  return GTE_OK;
}

static int log_window(gte_env* E, int32_t env_id, int32_t* n, int32_t* n_out, int64_t* first) {
  if (!E || !n_out) return fail(GTE_ERR_INVALID, "NULL argument");
  if (E->cfg.log_steps <= 0) return fail(GTE_ERR_STATE, "created with log_steps = 0");
  const int N = E->p.N, L = E->cfg.log_steps;
  if (env_id < 0 || env_id >= N) return fail(GTE_ERR_INVALID, "env_id out of range");
  const int64_t have = E->log_rows < L ? E->log_rows : L;
  if (*n > have) *n = (int32_t)have;
  if (*n < 0) *n = 0;
  *n_out = *n;
  *first = *n ? (E->log_rows - *n) % L : 0;
  if (*n) {
    HIPCHK(hipSetDevice(E->cfg.device));
    HIPCHK(hipStreamSynchronize(E->stream));
  }
  return GTE_OK;
}

int gte_read_log(gte_env* E, int32_t env_id, int32_t n, int32_t* idx, int32_t* step,
                 int32_t* position_index, int32_t* dataset_index, double* portfolio_valuation,
                 double* real_position, double* reward, uint8_t* flags, int32_t* n_out) {
  int64_t first = 0;
  TRY(log_window(E, env_id, &n, n_out, &first));
  if (n == 0) return GTE_OK;
  TRY(pull_log_column(E, env_id, first, n, idx, offsetof(gte::LogRow, idx), 4));
  TRY(pull_log_column(E, env_id, first, n, step, offsetof(gte::LogRow, step), 4));
  TRY(pull_log_column(E, env_id, first, n, position_index, offsetof(gte::LogRow, pos), 4));
  TRY(pull_log_column(E, env_id, first, n, dataset_index, offsetof(gte::LogRow, dsi), 4));
  TRY(pull_log_column(E, env_id, first, n, portfolio_valuation, offsetof(gte::LogRow, pv), 8));
  TRY(pull_log_column(E, env_id, first, n, real_position, offsetof(gte::LogRow, realpos), 8));
  TRY(pull_log_column(E, env_id, first, n, reward, offsetof(gte::LogRow, reward), 8));
  TRY(pull_log_column(E, env_id, first, n, flags, offsetof(gte::LogRow, flags), 1));
  return GTE_OK;
}

int gte_read_log_portfolio(gte_env* E, int32_t env_id, int32_t n, double* asset, double* fiat,
                           double* interest_asset, double* interest_fiat, int32_t* n_out) {
  int64_t first = 0;
  TRY(log_window(E, env_id, &n, n_out, &first));
  if (n == 0) return GTE_OK;
  TRY(pull_log_column(E, env_id, first, n, asset, offsetof(gte::LogRow, asset), 8));
  TRY(pull_log_column(E, env_id, first, n, fiat, offsetof(gte::LogRow, fiat), 8));
  TRY(pull_log_column(E, env_id, first, n, interest_asset, offsetof(gte::LogRow, ia), 8));
  TRY(pull_log_column(E, env_id, first, n, interest_fiat, offsetof(gte::LogRow, ifi), 8));
  return GTE_OK;
}

int gte_read_log_envs(gte_env* E, const int32_t* env_ids, int32_t n_ids, int32_t max_rows,
                      int32_t finished, gte_log_batch* out) {
  if (!E || !env_ids || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  if (E->cfg.log_steps <= 0) return fail(GTE_ERR_STATE, "created with log_steps = 0");
  if (n_ids < 0) return fail(GTE_ERR_INVALID, "n_ids must be >= 0");
  if (finished && !E->p.final_rec)
    return fail(GTE_ERR_STATE, "finished episodes need autoreset = same-step with final_obs");
  const int N = E->p.N, L = E->cfg.log_steps;
  if (max_rows <= 0 || max_rows > L) max_rows = L;
  for (int32_t i = 0; i < n_ids; ++i)
    if (env_ids[i] < 0 || env_ids[i] >= N) return fail(GTE_ERR_INVALID, "env_ids[%d] = %d out of range", i, env_ids[i]);
  memset(out, 0, sizeof *out);
  out->n_ids = n_ids; out->max_rows = max_rows;
  if (n_ids == 0) return GTE_OK;
  HIPCHK(hipSetDevice(E->cfg.device));
  // [ids i32 n | n_rows i32 n | pad -> 16] [7 f64 columns] [4 i32 columns] [u8 column]
  const size_t cells = (size_t)n_ids * (size_t)max_rows;
  const size_t head = (((size_t)n_ids * 8) + 15) & ~(size_t)15;
  const size_t need = head + cells * (7 * 8 + 4 * 4 + 1);
  if (need > E->h_logpack_bytes) {  // pinned and mapped: the kernel writes host memory directly
    if (E->h_logpack) HIPCHK(hipHostFree(E->h_logpack));
    E->h_logpack = nullptr; E->h_logpack_bytes = 0;
    HIPCHK(hipHostMalloc(&E->h_logpack, need + need / 2, hipHostMallocMapped));
    E->h_logpack_bytes = need + need / 2;
  }
  char* b = (char*)E->h_logpack;
  int32_t* ids = (int32_t*)b;
  memcpy(ids, env_ids, sizeof(int32_t) * (size_t)n_ids);
  gte::LogPack o;
  o.n_rows = ids + n_ids;
  double* f = (double*)(b + head);
  o.pv = f; o.realpos = f + cells; o.reward = f + 2 * cells; o.asset = f + 3 * cells;
  o.fiat = f + 4 * cells; o.ia = f + 5 * cells; o.ifi = f + 6 * cells;
  int32_t* w = (int32_t*)(f + 7 * cells);
  o.idx = w; o.step = w + cells; o.pos = w + 2 * cells; o.dsi = w + 3 * cells;
  o.flags = (uint8_t*)(w + 4 * cells);
  HIPCHK(gte::launch_pack_log(E->log, N, L, (long long)E->log_rows, ids, n_ids, max_rows, finished ? 1 : 0,
                              E->p.final_rec, E->p.reward64, o, E->stream));
  HIPCHK(hipStreamSynchronize(E->stream));
  out->n_rows = o.n_rows;
  out->idx = o.idx; out->step = o.step; out->position_index = o.pos; out->dataset_index = o.dsi;
  out->portfolio_valuation = o.pv; out->real_position = o.realpos; out->reward = o.reward;
  out->asset = o.asset; out->fiat = o.fiat; out->interest_asset = o.ia; out->interest_fiat = o.ifi;
  out->flags = o.flags;
  return GTE_OK;
}

int gte_set_log_reward(gte_env* E, const double* reward_device) {
  if (!E || !reward_device) return fail(GTE_ERR_INVALID, "NULL argument");
  if (E->cfg.log_steps <= 0) return fail(GTE_ERR_STATE, "created with log_steps = 0");
  if (E->log_rows <= 0) return fail(GTE_ERR_STATE, "the log is empty");
  const int64_t row = (E->log_rows - 1) % E->cfg.log_steps;
  HIPCHK(hipMemcpy2DAsync(&E->log.rows[row * (int64_t)E->p.N].reward, sizeof(gte::LogRow), reward_device,
                          sizeof(double), sizeof(double), (size_t)E->p.N, hipMemcpyDeviceToDevice, E->stream));
  return GTE_OK;
}

int gte_apply_reward(gte_env* E, const double* reward_device, int32_t terminal_view) {
  if (!E || !reward_device) return fail(GTE_ERR_INVALID, "NULL argument");
  if (E->cfg.log_steps <= 0) return fail(GTE_ERR_STATE, "created with log_steps = 0");
  if (E->log_rows <= 0) return fail(GTE_ERR_STATE, "the log is empty");
  HIPCHK(hipSetDevice(E->cfg.device));
  const int64_t row = (E->log_rows - 1) % E->cfg.log_steps;
  HIPCHK(gte::launch_apply_reward(E->p, reward_device, E->log.rows + row * (int64_t)E->p.N,
                                  terminal_view ? 1 : 0, E->stream));
  return GTE_OK;
}

int gte_get_outputs(gte_env* E, gte_outputs* out) {
  if (!E || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(E->cfg.device));
  TRY(ensure_owned_obs(E));
  const Params& p = E->p;
  out->obs = p.obs; out->reward = p.reward; out->reward64 = p.reward64;
  out->terminated = p.terminated; out->truncated = p.truncated;
  out->term_count = E->term_base; out->term_ids = p.term_ids;
  out->obs_elems_per_env = (int64_t)p.W * p.Fobs;
  out->term_slot = E->term_slot;
  out->reserved0 = 0;
  out->final_obs = p.final_obs;
  return GTE_OK;
}

int gte_bind_outputs(gte_env* E, const gte_outputs* b) {
  if (!E || !b) return fail(GTE_ERR_INVALID, "NULL argument");
  HIPCHK(hipStreamSynchronize(E->stream));
  Params& p = E->p;
  if (b->obs && ((uintptr_t)b->obs & 15)) return fail(GTE_ERR_INVALID, "obs must be 16-byte aligned");
  p.obs = b->obs;  // NULL: back to the library's own buffers (allocated on first need)
  if (E->cfg.final_obs) p.final_obs = b->final_obs;
  if (E->was_reset) TRY(ensure_owned_obs(E));
  p.reward = b->reward ? b->reward : E->owned.reward;
  p.reward64 = b->reward64 ? b->reward64 : E->owned.reward64;
  p.terminated = b->terminated ? b->terminated : E->owned.terminated;
  p.truncated = b->truncated ? b->truncated : E->owned.truncated;
  E->term_base = b->term_count ? b->term_count : E->owned.term_count;  // i32 [2]
  HIPCHK(hipMemset(E->term_base, 0, 2 * sizeof(int32_t)));
  HIPCHK(hipDeviceSynchronize());
  E->term_slot = 0;
  p.term_ids = b->term_ids ? b->term_ids : E->owned.term_ids;
  return GTE_OK;
}

static_assert(sizeof(gte_env_snapshot) == 96, "gte_env_snapshot layout");

static int read_envs_impl(gte_env* E, int32_t first, int32_t count, int32_t want_obs,
                          const gte_env_snapshot** out, const float** obs, bool hands_out_views) {
  if (!E || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_read_envs before gte_reset");
  if (first < 0 || count < 1 || (int64_t)first + count > E->p.N)
    return fail(GTE_ERR_INVALID, "envs %d..%lld out of range", first, (long long)first + count - 1);
  if (want_obs && !obs) return fail(GTE_ERR_INVALID, "obs is NULL");
  HIPCHK(hipSetDevice(E->cfg.device));
  const size_t elems = (size_t)E->p.W * (size_t)E->p.Fobs;
  const size_t head = sizeof(gte_env_snapshot) * (size_t)count;  // 96 B each: 16-byte aligned
  const size_t need = head + (want_obs ? sizeof(float) * elems * (size_t)count : 0);
  // Callers may hold views into the staging buffer: once it exists at full size it is never
  // reallocated, so size it for the whole batch the first time a buffer is made.
  const size_t full = (sizeof(gte_env_snapshot) + sizeof(float) * elems) * (size_t)E->p.N;
  if (hands_out_views) E->view_reads = true;
  const size_t want = E->view_reads ? full : need;
  if (want > E->h_snap_bytes) {  // pinned and mapped: the kernel writes host memory directly
    if (E->h_snap) HIPCHK(hipHostFree(E->h_snap));
    E->h_snap = nullptr; E->h_snap_bytes = 0;
    HIPCHK(hipHostMalloc(&E->h_snap, want, hipHostMallocMapped));
    E->h_snap_bytes = want;
  }
  float* h_obs = (float*)((char*)E->h_snap + head);
  HIPCHK(gte::launch_snapshot(E->p.rec, E->p.reward64, E->p.terminated, E->p.truncated, E->p.obs,
                              (int64_t)elems, first, count, E->h_snap, want_obs ? h_obs : nullptr,
                              E->stream));
  HIPCHK(hipStreamSynchronize(E->stream));
  *out = (const gte_env_snapshot*)E->h_snap;
  if (obs) *obs = want_obs ? h_obs : nullptr;
  return GTE_OK;
}

int gte_read_envs_view(gte_env* E, int32_t first, int32_t count, int32_t want_obs,
                       const gte_env_snapshot** out, const float** obs) {
  return read_envs_impl(E, first, count, want_obs, out, obs, true);
}

int gte_read_envs(gte_env* E, int32_t first, int32_t count, gte_env_snapshot* out, float* obs) {
  if (!out) return fail(GTE_ERR_INVALID, "NULL argument");
  const gte_env_snapshot* snaps = nullptr;
  const float* h_obs = nullptr;
  TRY(read_envs_impl(E, first, count, obs ? 1 : 0, &snaps, &h_obs, false));
  memcpy(out, snaps, sizeof(gte_env_snapshot) * (size_t)count);
  if (obs) memcpy(obs, h_obs, sizeof(float) * (size_t)E->p.W * (size_t)E->p.Fobs * (size_t)count);
  return GTE_OK;
}

int gte_read_env(gte_env* E, int32_t e, gte_env_snapshot* out, float* obs) {
  return gte_read_envs(E, e, 1, out, obs);
}

int gte_bind_returns(gte_env* E, float* reward, uint8_t* terminated, uint8_t* truncated) {
  if (!E || !reward || !terminated || !truncated) return fail(GTE_ERR_INVALID, "NULL argument");
  // Params travel by value with every launch: later launches see the new pointers,
  // launches already enqueued keep the old ones.
  E->p.reward = reward;
  E->p.terminated = terminated;
  E->p.truncated = truncated;
  return GTE_OK;
}

int gte_get_state(gte_env* E, gte_state_view* out) {
  if (!E || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  // the state lives in 128-byte records; snapshot it into struct-of-arrays mirrors
  // (stream-ordered: the views reflect every launch enqueued before this call)
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(gte::launch_extract_state(E->p.rec, E->p.N, E->soa, E->stream));
  const gte::StateSoA& o = E->soa;
  out->idx = o.idx; out->step = o.step; out->position_index = o.pos;
  out->dataset_index = o.dsi; out->start_idx = o.start; out->episode = o.episode;
  out->needs_reset = o.needs_reset; out->asset = o.asset; out->fiat = o.fiat;
  out->interest_asset = o.ia; out->interest_fiat = o.ifi;
  out->portfolio_valuation = o.pv; out->real_position = o.realpos;
  return GTE_OK;
}

int gte_set_dynamic_features(gte_env* E, const float* values_device, uint32_t mask) {
  if (!E || !values_device) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_set_dynamic_features before gte_reset");
  if (E->p.nd <= 0 || (mask >> E->p.nd) != 0u)
    return fail(GTE_ERR_INVALID, "mask 0x%x names features beyond n_dyn = %d", mask, E->p.nd);
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(gte::launch_set_dynamic(E->p, values_device, mask, E->stream));
  return GTE_OK;
}

int gte_set_dynamic_columns(gte_env* E, const void* const* columns_device, const int32_t* is_f64) {
  if (!E || !columns_device || !is_f64) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->was_reset) return fail(GTE_ERR_STATE, "gte_set_dynamic_columns before gte_reset");
  if (E->p.nd <= 0) return fail(GTE_ERR_INVALID, "the env has no dynamic features");
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(gte::launch_set_dynamic_columns(E->p, columns_device, is_f64, E->stream));
  return GTE_OK;
}

int gte_get_final_state(gte_env* E, gte_state_view* out) {
  if (!E || !out) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->p.final_rec) return fail(GTE_ERR_STATE, "created without final_obs");
  HIPCHK(hipSetDevice(E->cfg.device));
  HIPCHK(gte::launch_extract_state(E->p.final_rec, E->p.N, E->fsoa, E->stream));
  const gte::StateSoA& o = E->fsoa;
  out->idx = o.idx; out->step = o.step; out->position_index = o.pos;
  out->dataset_index = o.dsi; out->start_idx = o.start; out->episode = o.episode;
  out->needs_reset = o.needs_reset; out->asset = o.asset; out->fiat = o.fiat;
  out->interest_asset = o.ia; out->interest_fiat = o.ifi;
  out->portfolio_valuation = o.pv; out->real_position = o.realpos;
  return GTE_OK;
}

// ---------------------------------------------------------------------------
// multi-GPU: the return all-gather over RCCL (gte_comm.hip)

int gte_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return fail(GTE_ERR_INVALID, "id_out is NULL");
  if (const char* why = gte::rccl_load()) return fail(GTE_ERR_STATE, "RCCL unavailable: %s", why);
  const int r = gte::rccl_unique_id(id_out);
  if (r != 0) return fail(GTE_ERR_HIP, "ncclGetUniqueId: %s", gte::rccl_error(r));
  return GTE_OK;
}

int gte_comm_init(gte_env* E, const uint8_t* id, int32_t rank, int32_t world) {
  if (!E || !id) return fail(GTE_ERR_INVALID, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(GTE_ERR_INVALID, "rank %d of %d", rank, world);
  if (E->comm) return fail(GTE_ERR_STATE, "the env already has a communicator");
  if (const char* why = gte::rccl_load()) return fail(GTE_ERR_STATE, "RCCL unavailable: %s", why);
  HIPCHK(hipSetDevice(E->cfg.device));
  const int r = gte::rccl_comm_init(&E->comm, id, rank, world);
  if (r != 0) { E->comm = nullptr; return fail(GTE_ERR_HIP, "ncclCommInitRank: %s", gte::rccl_error(r)); }
  E->comm_rank = rank;
  E->comm_world = world;
  // anything failing from here on must not leave a half-made communicator behind (a retry would
  // be refused with "already has a communicator"): undo everything, keep the error message
  auto finish = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&E->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&E->comm_ready, hipEventDisableTiming));
    for (auto& ev : E->comm_done) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (E->gathered_returns) {  // a communicator of another size was here before
      for (void*& q : E->allocs) if (q == (void*)E->gathered_returns) q = nullptr;
      (void)hipFree(E->gathered_returns);
      E->gathered_returns = nullptr;
    }
    TRY(dev_alloc(E, &E->gathered_returns, (size_t)world * 6 * (size_t)E->p.N));
    HIPCHK(hipDeviceSynchronize());
    return GTE_OK;
  };
  const int rc = finish();
  if (rc != GTE_OK) {
    const std::string keep = g_err;
    (void)gte_comm_destroy(E);
    g_err = keep;
  }
  return rc;
}

int gte_allgather(gte_env* E, const void* src_device, void* dst_device, uint64_t bytes_per_rank,
                  int32_t mode) {
  if (!E || !src_device || !dst_device) return fail(GTE_ERR_INVALID, "NULL argument");
  if (!E->comm) return fail(GTE_ERR_STATE, "gte_allgather before gte_comm_init");
  if (mode != 0 && mode != 1) return fail(GTE_ERR_INVALID, "mode must be 0 (env stream) or 1 (overlapped)");
  hipStream_t s = E->stream;
  if (mode == 1) {  // behind everything enqueued so far, beside everything enqueued later
    HIPCHK(hipEventRecord(E->comm_ready, E->stream));
    HIPCHK(hipStreamWaitEvent(E->comm_stream, E->comm_ready, 0));
    s = E->comm_stream;
  }
  const int r = gte::rccl_allgather_bytes(E->comm, src_device, dst_device, (size_t)bytes_per_rank, s);
  if (r != 0) return fail(GTE_ERR_HIP, "ncclAllGather: %s", gte::rccl_error(r));
  if (mode == 1) {
    HIPCHK(hipEventRecord(E->comm_done[E->comm_seq & 3], E->comm_stream));
    E->comm_seq += 1;
  }
  return GTE_OK;
}

int gte_allgather_returns(gte_env* E, void* dst_device, int32_t mode, const void** gathered) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (!E->comm) return fail(GTE_ERR_STATE, "gte_allgather_returns before gte_comm_init");
  const Params& p = E->p;
  const size_t N = (size_t)p.N;
  if ((const uint8_t*)p.terminated != (const uint8_t*)p.reward + 4 * N || p.truncated != p.terminated + N)
    return fail(GTE_ERR_STATE, "the bound return buffers are not one packed [reward f32 | terminated u8 "
                               "| truncated u8] block of 6N bytes");
  void* dst = dst_device ? dst_device : (void*)E->gathered_returns;
  TRY(gte_allgather(E, p.reward, dst, 6 * N, mode));
  if (gathered) *gathered = dst;
  return GTE_OK;
}

int gte_allgather_obs(gte_env* E, float* dst_device, int32_t mode) {
  if (!E || !dst_device) return fail(GTE_ERR_INVALID, "NULL argument");
  const Params& p = E->p;
  if (!p.obs) return fail(GTE_ERR_STATE, "gte_allgather_obs before gte_reset (no observation exists yet)");
  return gte_allgather(E, p.obs, dst_device, sizeof(float) * (size_t)p.N * p.W * p.Fobs, mode);
}

int gte_comm_wait(gte_env* E, int32_t back) {
  if (!E || !E->comm) return fail(GTE_ERR_STATE, "no communicator");
  if (back < 0 || back > 3) return fail(GTE_ERR_INVALID, "back must be 0..3");
  const int64_t k = E->comm_seq - 1 - back;  // the overlapped gather to wait for
  if (k < 0) return GTE_OK;                  // not issued yet: nothing to wait for
  HIPCHK(hipStreamWaitEvent(E->stream, E->comm_done[k & 3], 0));
  return GTE_OK;
}

int gte_comm_synchronize(gte_env* E) {
  if (!E || !E->comm) return fail(GTE_ERR_STATE, "no communicator");
  HIPCHK(hipStreamSynchronize(E->comm_stream));
  return GTE_OK;
}

int gte_comm_destroy(gte_env* E) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (!E->comm) return GTE_OK;
  (void)hipStreamSynchronize(E->stream);
  if (E->comm_stream) (void)hipStreamSynchronize(E->comm_stream);
  const int r = gte::rccl_comm_destroy(E->comm);
  E->comm = nullptr;
  if (E->comm_ready) (void)hipEventDestroy(E->comm_ready);
  for (auto& ev : E->comm_done) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
  if (E->comm_stream) (void)hipStreamDestroy(E->comm_stream);
  E->comm_ready = nullptr;
  E->comm_seq = 0;
  E->comm_stream = nullptr;
  if (r != 0) return fail(GTE_ERR_HIP, "ncclCommDestroy: %s", gte::rccl_error(r));
  return GTE_OK;
}

int gte_set_stream(gte_env* E, void* hip_stream) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  HIPCHK(hipStreamSynchronize(E->stream));
  E->stream = (hipStream_t)hip_stream;  // NULL = the null stream, used as such
  return GTE_OK;
}

int gte_use_own_stream(gte_env* E) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  HIPCHK(hipStreamSynchronize(E->stream));
  E->stream = E->own_stream;
  return GTE_OK;
}

int gte_synchronize(gte_env* E) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  HIPCHK(hipStreamSynchronize(E->stream));
  return GTE_OK;
}

int gte_timer_start(gte_env* E) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  HIPCHK(hipEventRecord(E->ev0, E->stream));
  return GTE_OK;
}

int gte_timer_stop(gte_env* E, float* elapsed_ms) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (!elapsed_ms) {  // mark only: record the end now (asynchronous), read it with a later call
    HIPCHK(hipEventRecord(E->ev1, E->stream));
    E->timer_marked = true;
    return GTE_OK;
  }
  if (!E->timer_marked) HIPCHK(hipEventRecord(E->ev1, E->stream));
  E->timer_marked = false;
  HIPCHK(hipEventSynchronize(E->ev1));
  HIPCHK(hipEventElapsedTime(elapsed_ms, E->ev0, E->ev1));
  return GTE_OK;
}

int gte_read_obs(gte_env* E, int32_t first_env, int32_t n, float* host_dst) {
  if (!E || !host_dst) return fail(GTE_ERR_INVALID, "NULL argument");
  const Params& p = E->p;
  if (first_env < 0 || n < 0 || (int64_t)first_env + n > p.N)
    return fail(GTE_ERR_INVALID, "env range [%d, %d) out of [0, %d)", first_env, first_env + n, p.N);
  const size_t per = (size_t)p.W * p.Fobs;
  if (!p.obs) return fail(GTE_ERR_STATE, "gte_read_obs before gte_reset (no observation exists yet)");
  HIPCHK(hipStreamSynchronize(E->stream));
  HIPCHK(hipMemcpy(host_dst, p.obs + per * first_env, sizeof(float) * per * n, hipMemcpyDeviceToHost));
  return GTE_OK;
}

int gte_copy_to_host(gte_env* E, const void* device_src, void* host_dst, uint64_t bytes) {
  if (!E || !device_src || !host_dst) return fail(GTE_ERR_INVALID, "NULL argument");
  HIPCHK(hipStreamSynchronize(E->stream));
  HIPCHK(hipMemcpy(host_dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost));
  return GTE_OK;
}

int gte_get_launch_info(gte_env* E, int32_t* envs_per_wave, int32_t* threads_per_block,
                        int32_t* n_blocks, int32_t* vector_bytes) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  if (envs_per_wave) *envs_per_wave = E->p.epw;
  if (threads_per_block) *threads_per_block = E->threads;
  if (n_blocks) *n_blocks = E->blocks;
  if (vector_bytes)
    *vector_bytes = E->vec * 4 + 1000 * ((E->coop ? 1 : 0) + 2 * E->stage +
                                         16 * E->cfg.nontemporal_obs + 64 * (E->hot_per_cu & 15));
  return GTE_OK;
}

#ifdef GTE_STAMPS
// diagnostic build only: device buffer of u64 [n_blocks, 8] for the kernel's time stamps
int gte_debug_set_stamps(gte_env* E, void* device_buf) {
  if (!E) return fail(GTE_ERR_INVALID, "env is NULL");
  E->p.inj_ds = (const int32_t*)device_buf;
  return GTE_OK;
}
#endif

void gte_destroy(gte_env* E) {
  if (!E) return;
  (void)hipSetDevice(E->cfg.device);
  (void)gte_comm_destroy(E);
  (void)hipStreamSynchronize(E->stream);
  if (E->own_stream) (void)hipStreamSynchronize(E->own_stream);
  for (void* ptr : E->allocs) if (ptr) (void)hipFree(ptr);
  for (auto& v : E->ds_allocs)
    for (void* ptr : v)
      if (ptr) (void)hipFree(ptr);
  for (int32_t* q : {E->d_q_idx, E->d_q_pos, E->d_q_ds})
    if (q) (void)hipFree(q);
  if (E->h_snap) (void)hipHostFree(E->h_snap);
  if (E->h_logpack) (void)hipHostFree(E->h_logpack);
  if (E->ev0) (void)hipEventDestroy(E->ev0);
  if (E->ev1) (void)hipEventDestroy(E->ev1);
  if (E->own_stream) (void)hipStreamDestroy(E->own_stream);
  delete E;
}

}  // extern "C"
