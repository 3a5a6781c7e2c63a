// gte_hot.hip — the ONE instantiation the headline shape runs (classic step kernel, 16-byte
// vectors, sc1 stores, cooperative phase A, raw rings staged in LDS), compiled alone in its
// own translation unit.  hipcc's code generation for a kernel depends on what is compiled
// next to it (builds of the same source differed by +-5 us per step); an isolated TU makes
// the hot kernel's code independent of every other kernel in the library.
#define GTE_HOT_ONLY 1
#include "gte_kernels.hip"

namespace gte {

hipError_t launch_step_hot(const Params& p, int blocks, int threads, size_t smem,
                           hipStream_t stream) {
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  auto magic = [](uint32_t d) { return ((1ull << 40) + d - 1) / d; };
  const uint64_t vm = magic(V / 4), fm = magic((uint32_t)p.Fobs / 4),
                 wm = magic((uint32_t)(p.W * (p.nd ? p.nd : 1)));
  hipLaunchKernelGGL((gte_kernel<MODE_STEP, 4, 2, true, STAGE_RAW>), dim3(blocks), dim3(threads), smem,
                     stream, p, vm, fm, wm);
  return hipGetLastError();
}

}  // namespace gte
