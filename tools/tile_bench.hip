// tile_bench.hip — would a window gather fed from an LDS TILE beat one fed by L2-hit loads?
// With the L2-affinity order the 64 envs of a workgroup sit within ~100 table rows of one another,
// so the union of their 20-row windows is ~130 rows (17 KB) instead of 64 x 20 rows (164 KB): stage
// that union once per workgroup in LDS (coalesced), then assemble every env's window from LDS — the
// vector-memory pipe then only carries the observation stores (store-only floor 23-24 us for 168 MB,
// profiles/r01_store_bench.log) instead of loads + stores (30-35 us, same log; the product's gather
// runs 30.9 us).  Same geometry as the step kernel: 1 024 workgroups x 4 waves x 16 envs, 1 KiB per
// wave instruction, sc1 stores, 4 vectors in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -w tools/tile_bench.hip -o /tmp/tb && /tmp/tb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 20, FV = 8, VPE = W * FV, EPW = 16, TILE_ROWS = 160;

__device__ inline void st_sc1(f4* dst, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory"); }

// MODE 0: loads from the table (L2 hits) + stores — today's gather; 1: tile staged in LDS, windows from
// LDS; 2: store-only (registers)
template <int MODE>
__global__ __launch_bounds__(256) void k(f4* __restrict__ obs, const f4* __restrict__ table,
                                         const int* __restrict__ first_row, const int* __restrict__ env_of_slot, int n_env) {
  extern __shared__ f4 tile[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wg_first = blockIdx.x * 4 * EPW;
  int base = 0;
  if (MODE == 1) {
    // the tile starts at the smallest first row of the workgroup's envs (sorted order: no outliers here)
    int f = (lane < 4 * EPW && wg_first + lane < n_env) ? first_row[wg_first + lane] : 0x7fffffff;  // lane, not thread: every wave reads all 64
    for (int o = 32; o; o >>= 1) f = min(f, __shfl_xor(f, o));
    base = f;  // (every wave computes it from the same 64 values)
    for (int k = threadIdx.x; k < TILE_ROWS * FV; k += 256) tile[k] = table[(size_t)base * FV + k];
    __syncthreads();
  }
  const int s_first = wib * EPW;
  const int total = EPW * VPE;
  for (int k0 = 0; k0 < total; k0 += 256) {
    f4 v[4]; int el[4], j[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u * 64 + lane;
      el[u] = kk / VPE; j[u] = kk - el[u] * VPE;
      const int fr = first_row[wg_first + s_first + el[u]];
      if (MODE == 0) v[u] = table[(size_t)fr * FV + j[u]];
      else if (MODE == 1) {
        const int off = (fr - base) * FV + j[u];
        v[u] = (off + 0 < TILE_ROWS * FV) ? tile[off] : table[(size_t)fr * FV + j[u]];
      } else v[u] = (f4){(float)kk, 1.f, 2.f, 3.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int env = env_of_slot[wg_first + s_first + el[u]];
      st_sc1(obs + (size_t)env * VPE + j[u], v[u]);
    }
  }
}

template <int MODE>
float run(f4* obs, const f4* table, const int* fr, const int* eos, int n_env, size_t smem, int iters) {
  const int blocks = n_env / (4 * EPW);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), smem, 0, obs, table, fr, eos, n_env);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), smem, 0, obs, table, fr, eos, n_env);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const int T = 100000, n_env = 65536;
  f4 *obs, *table; int *fr, *eos;
  hipMalloc(&obs, (size_t)n_env * VPE * 16); hipMalloc(&table, (size_t)(T + TILE_ROWS) * FV * 16);
  hipMalloc(&fr, n_env * 4); hipMalloc(&eos, n_env * 4);
  hipMemset(table, 0, (size_t)(T + TILE_ROWS) * FV * 16);
  std::mt19937 g(1);
  // sorted first rows (the affinity order), slots XCD-major: rank r -> workgroup (r/64 % 8-interleaved)
  std::vector<int> rows(n_env), h_fr(n_env), h_eos(n_env);
  for (int i = 0; i < n_env; ++i) rows[i] = g() % (T - W);
  std::sort(rows.begin(), rows.end());
  const int n_wg = n_env / 64;
  int r = 0;
  std::vector<int> perm(n_env);
  for (int i = 0; i < n_env; ++i) perm[i] = i;
  std::shuffle(perm.begin(), perm.end(), g);  // env ids are unrelated to table position
  for (int x = 0; x < 8; ++x)
    for (int b = x; b < n_wg; b += 8)
      for (int s = 0; s < 64; ++s, ++r) { h_fr[b * 64 + s] = rows[r]; h_eos[b * 64 + s] = perm[r]; }
  hipMemcpy(fr, h_fr.data(), n_env * 4, hipMemcpyHostToDevice);
  hipMemcpy(eos, h_eos.data(), n_env * 4, hipMemcpyHostToDevice);
  const size_t smem = TILE_ROWS * FV * 16;
  // the LDS of the product's workgroup (12.8 KB) rides along so that residency matches: 4 workgroups per CU either way
  printf("65 536 envs x 2 560 B = 168 MB per launch, sorted windows, 1 024 workgroups of 64 envs, sc1 stores, us per launch:\n");
  for (int rep = 0; rep < 3; ++rep)
    printf("  loads from L2 + stores %6.2f   LDS tile (%d rows, %zu B) + stores %6.2f   stores only %6.2f\n",
           run<0>(obs, table, fr, eos, n_env, 12800, 200), TILE_ROWS, smem, run<1>(obs, table, fr, eos, n_env, smem + 12800, 200),
           run<2>(obs, table, fr, eos, n_env, 12800, 200));
  return 0;
}
