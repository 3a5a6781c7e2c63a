// ring_bench.hip — VERDICT r02 item 7, decided by measurement: would observations kept as a per-env
// RING (a closed-loop step writes only the newest row, 128 B per env, plus a head index; the
// chronological [N, W, F_obs] tensor the reference returns is produced on demand) beat today's
// gather, which re-emits all W rows every step (TradingEnv._get_obs, environments.py:156-160)?
//
// Two kernels with the access patterns such a mode would have, nothing else:
//   k_row          the observation part of a ring step: one table row per env (random rows, like
//                  episodes out of phase), dynamic columns patched, stored into ring[env][head]
//                  (128 B per env; 8 lanes x 16 B per env, 8 envs per wave instruction)
//   k_materialize  ring -> chronological obs: a permuted copy, vector j of env e comes from ring
//                  row (head[e] + 1 + j / FV) mod W; 1 KiB per wave instruction like the step
//                  kernel's gather (sc1 / nt / plain stores)
// Reference points measured elsewhere (profiles/): today's step 39.4 us at 65 536 envs (gather 30.9 of
// them, phase A + dispatch ~7.5-10), 21.8 us at 32 768 (config 4's share), 35.5-37.5 us at config 5,
// 142 us at 262 144; the step kernel WITHOUT its gather 13.2 us (r02_ablation_phase_a.log).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ring_bench.hip -o /tmp/rb && /tmp/rb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int W = 20, FV = 8, VPE = W * FV;  // 20 rows x 32 floats = 160 float4 per env

template <int F>
__device__ inline void st(f4* dst, f4 v) {
  if (F == 0) *dst = v;
  else if (F == 1) __builtin_nontemporal_store(v, dst);
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void k_row(f4* __restrict__ ring, const f4* __restrict__ table,
                                             const int* __restrict__ row, int* __restrict__ head, int n_env) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int e = t >> 3, j = t & 7;
  if (e >= n_env) return;
  int h = head[e] + 1;
  if (h >= W) h = 0;
  f4 v = table[(size_t)row[e] * FV + j];
  if (j == FV - 1) { v[2] = 1.0f; v[3] = 0.5f; }  // the two dynamic columns of the new row
  ring[(size_t)e * VPE + h * FV + j] = v;
  if (j == 0) head[e] = h;
}

template <int NT>
__global__ __launch_bounds__(256) void k_materialize(f4* __restrict__ obs, const f4* __restrict__ ring,
                                                     const int* __restrict__ head, int n_env, int epw) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int first = wave * epw;
  if (first >= n_env) return;
  const int n = min(epw, n_env - first);
  const int total = n * VPE;
  for (int k0 = 0; k0 < total; k0 += 256) {
    f4 v[4];
    int kk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * 64 + lane;
      kk[u] = k < total ? k : -1;
      if (kk[u] >= 0) {
        const int el = k / VPE, j = k - el * VPE;
        const int e = first + el;
        int r = head[e] + 1 + j / FV;
        if (r >= W) r -= W;
        v[u] = ring[(size_t)e * VPE + r * FV + (j % FV)];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (kk[u] >= 0) st<NT>(obs + (size_t)first * VPE + kk[u], v[u]);
  }
}

template <typename L>
float time_us(L launch, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1e3f / iters;
}

int main() {
  const int T = 100000;
  for (int n_env : {32768, 65536, 262144}) {
    f4 *ring, *obs, *table; int *row, *head;
    hipMalloc(&ring, (size_t)n_env * VPE * 16); hipMalloc(&obs, (size_t)n_env * VPE * 16);
    hipMalloc(&table, (size_t)T * FV * 16); hipMalloc(&row, n_env * 4); hipMalloc(&head, n_env * 4);
    hipMemset(ring, 0, (size_t)n_env * VPE * 16); hipMemset(table, 0, (size_t)T * FV * 16);
    std::mt19937 g(1);
    std::vector<int> r(n_env), h(n_env);
    for (int i = 0; i < n_env; ++i) { r[i] = g() % (T - 1); h[i] = g() % W; }
    hipMemcpy(row, r.data(), n_env * 4, hipMemcpyHostToDevice);
    hipMemcpy(head, h.data(), n_env * 4, hipMemcpyHostToDevice);
    const int rb = (n_env * 8 + 255) / 256;
    const float t_row = time_us([&] { hipLaunchKernelGGL(k_row, dim3(rb), dim3(256), 0, 0, ring, table, row, head, n_env); }, 200);
    printf("%7d envs (%4.0f MB of observations): ring-row write %6.2f us", n_env, n_env * 2560.0 / 1e6, t_row);
    for (int epw : {4, 16}) {
      const int waves = (n_env + epw - 1) / epw, mb = (waves + 3) / 4;
      const float p = time_us([&] { hipLaunchKernelGGL((k_materialize<0>), dim3(mb), dim3(256), 0, 0, obs, ring, head, n_env, epw); }, 100);
      const float nt = time_us([&] { hipLaunchKernelGGL((k_materialize<1>), dim3(mb), dim3(256), 0, 0, obs, ring, head, n_env, epw); }, 100);
      const float s1 = time_us([&] { hipLaunchKernelGGL((k_materialize<2>), dim3(mb), dim3(256), 0, 0, obs, ring, head, n_env, epw); }, 100);
      // the pair as a consumer would run it: newest row, then the chronological tensor
      const float both = time_us([&] {
        hipLaunchKernelGGL(k_row, dim3(rb), dim3(256), 0, 0, ring, table, row, head, n_env);
        hipLaunchKernelGGL((k_materialize<2>), dim3(mb), dim3(256), 0, 0, obs, ring, head, n_env, epw); }, 100);
      printf(" | materialise, %2d envs/wave: plain %6.2f  nt %6.2f  sc1 %6.2f  row+sc1 %6.2f us", epw, p, nt, s1, both);
    }
    printf("\n");
    hipFree(ring); hipFree(obs); hipFree(table); hipFree(row); hipFree(head);
  }
  return 0;
}
