// stream_bench.hip — the floor of a rollout that keeps every step's observations: K rows of
// 65 536 x 2 560 B = 168 MB each, every row written once and never read back (a pure HBM write
// stream, unlike store_bench.hip whose single 168 MB buffer stays in the Infinity Cache).
//   order:  sequential env order, or a random permutation (the L2-affinity order scatters the
//           2.5 KiB chunks inside a row);  stores: plain / nt / sc1;  one launch per row.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o /tmp/stb && /tmp/stb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int VPE = 160;

template <int F>
__device__ inline void st(f4* dst, f4 v) {
  if (F == 0) *dst = v;
  else if (F == 1) __builtin_nontemporal_store(v, dst);
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

template <int NT>
__global__ __launch_bounds__(256) void k_store(f4* __restrict__ obs, const int* __restrict__ perm,
                                               int n_env, int epw) {
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
        st<NT>(obs + (size_t)env * VPE + j, v);
      }
    }
  }
}

template <int NT>
float run(f4* obs, int rows, const int* perm, int n_env, int epw, int iters) {
  const int waves = (n_env + epw - 1) / epw, blocks = (waves + 3) / 4;
  const size_t row = (size_t)n_env * VPE;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < rows; ++i) hipLaunchKernelGGL((k_store<NT>), dim3(blocks), dim3(256), 0, 0, obs + row * (i % rows), perm, n_env, epw);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_store<NT>), dim3(blocks), dim3(256), 0, 0, obs + row * (i % rows), perm, n_env, epw);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const int N = 65536, iters = 256;
  int *p_seq, *p_rnd;
  hipMalloc(&p_seq, N * 4); hipMalloc(&p_rnd, N * 4);
  std::vector<int> h(N); std::iota(h.begin(), h.end(), 0);
  hipMemcpy(p_seq, h.data(), N * 4, hipMemcpyHostToDevice);
  std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
  hipMemcpy(p_rnd, h.data(), N * 4, hipMemcpyHostToDevice);
  const double mb = (double)N * VPE * 16 / 1e6;
  for (int rows : {1, 32}) {
    f4* obs; hipMalloc(&obs, (size_t)rows * N * VPE * 16);
    for (int epw : {16, 4})
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int* p = rnd ? p_rnd : p_seq;
        const float t0 = run<0>(obs, rows, p, N, epw, iters), t1 = run<1>(obs, rows, p, N, epw, iters),
                    t2 = run<2>(obs, rows, p, N, epw, iters);
        printf("rows %2d (%5.0f MB)  epw %2d  %-10s  store-only: plain %5.1f us (%.2f TB/s)  nt %5.1f us (%.2f TB/s)  sc1 %5.1f us (%.2f TB/s)\n",
               rows, rows * mb, epw, rnd ? "permuted" : "sequential", t0, mb / t0 / 1e6, t1, mb / t1 / 1e6, t2, mb / t2 / 1e6);
      }
    hipFree(obs);
  }
  return 0;
}
