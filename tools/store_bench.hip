// store_bench.hip — what does the observation store pattern cost by itself?
// 65 536 "envs" x 2 560 B = 168 MB written per launch, as 1 KiB wave instructions.
//   order:  sequential env order, or a random permutation of the envs (the L2-affinity
//           order scatters the 2.5 KiB chunks);  stores: plain or non-temporal;
//   epw:    envs per wave (16 = the step kernel's geometry).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int VPE = 160;  // float4 vectors per env (20 x 32 floats)

// store flavours: 0 plain, 1 nt (builtin), 2 sc1, 3 sc0 sc1, 4 nt sc1  (inline asm for 2..4)
template <int F>
__device__ inline void st(f4* dst, f4 v) {
  if (F == 0) *dst = v;
  else if (F == 1) __builtin_nontemporal_store(v, dst);
  else if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
  else if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v) : "memory");
}

template <int NT, bool WITH_LOAD>
__global__ __launch_bounds__(256) void k_store(f4* __restrict__ obs, const int* __restrict__ perm,
                                               const f4* __restrict__ table, int n_env, int epw) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int first = wave * epw;
  if (first >= n_env) return;
  const int total = epw * VPE;
  for (int k0 = 0; k0 < total; k0 += 256) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * 64 + lane;
      if (k < total) {
        const int el = k / VPE, j = k - el * VPE;
        const int env = perm[first + el];
        f4 v = {(float)k, 1.f, 2.f, 3.f};
        if (WITH_LOAD) v = table[(size_t)(env % 5000) * VPE + j];  // L2-resident source
        f4* dst = obs + (size_t)env * VPE + j;
        st<NT>(dst, v);
      }
    }
  }
}

template <int NT, bool WL>
float run(f4* obs, const int* perm, const f4* table, int n_env, int epw, int iters) {
  const int waves = (n_env + epw - 1) / epw, blocks = (waves + 3) / 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<NT, WL>), dim3(blocks), dim3(256), 0, 0, obs, perm, table, n_env, epw);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_store<NT, WL>), dim3(blocks), dim3(256), 0, 0, obs, perm, table, n_env, epw);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const int N = 65536, iters = 200;
  f4 *obs, *table; int *p_seq, *p_rnd;
  hipMalloc(&obs, (size_t)N * VPE * 16); hipMalloc(&table, (size_t)5000 * VPE * 16);
  hipMemset(table, 0, (size_t)5000 * VPE * 16);
  hipMalloc(&p_seq, N * 4); hipMalloc(&p_rnd, N * 4);
  std::vector<int> h(N); std::iota(h.begin(), h.end(), 0);
  hipMemcpy(p_seq, h.data(), N * 4, hipMemcpyHostToDevice);
  std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
  hipMemcpy(p_rnd, h.data(), N * 4, hipMemcpyHostToDevice);
  const double mb = (double)N * VPE * 16 / 1e6;
  for (int epw : {16, 4})
    for (int rnd = 0; rnd < 2; ++rnd) {
      const int* p = rnd ? p_rnd : p_seq;
      printf("epw %2d  %-10s  store-only: plain %5.1f  nt %5.1f  sc1 %5.1f  sc0sc1 %5.1f  sc1nt %5.1f us | load+store: plain %5.1f  nt %5.1f  sc1 %5.1f  sc0sc1 %5.1f  sc1nt %5.1f us  (%.0f MB)\n",
             epw, rnd ? "permuted" : "sequential",
             run<0, false>(obs, p, table, N, epw, iters), run<1, false>(obs, p, table, N, epw, iters),
             run<2, false>(obs, p, table, N, epw, iters), run<3, false>(obs, p, table, N, epw, iters),
             run<4, false>(obs, p, table, N, epw, iters),
             run<0, true>(obs, p, table, N, epw, iters), run<1, true>(obs, p, table, N, epw, iters),
             run<2, true>(obs, p, table, N, epw, iters), run<3, true>(obs, p, table, N, epw, iters),
             run<4, true>(obs, p, table, N, epw, iters), mb);
    }
  return 0;
}
