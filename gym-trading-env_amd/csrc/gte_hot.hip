// gte_hot.hip — the ONE instantiation the headline shape runs (classic step kernel, 16-byte
// vectors, cooperative phase A, raw rings staged in LDS), compiled alone in its own
// translation unit.  hipcc's code generation for a kernel depends on what is compiled next to
// it (builds of the same source differed by +-5 us per step); an isolated TU makes the hot
// kernel's code independent of every other kernel in the library.
//
// Compiled twice: as is (sc1 observation stores: the observation buffer stays in the
// Infinity Cache, batches up to ~190 MB of observations) and through gte_hot_nt.hip
// (non-temporal stores: streaming, for bigger batches).
#define GTE_HOT_ONLY 1
#include "gte_kernels.hip"

#ifndef GTE_HOT_NT
#define GTE_HOT_NT 2
#define GTE_HOT_NAME(x) x
#endif

namespace gte {

hipError_t GTE_HOT_NAME(launch_step_hot)(const Params& p, int blocks, int threads, size_t smem,
                                         hipStream_t stream) {
  if (!hot_tu_covers(p)) return hipErrorInvalidValue;  // features compiled out of this TU (gte_device.h)
  const uint32_t V = (uint32_t)(p.W * p.Fobs);
  auto magic = [](uint32_t d) { return ((1ull << 40) + d - 1) / d; };
  const uint64_t vm = magic(V / 4), fm = magic((uint32_t)p.Fobs / 4),
                 wm = magic((uint32_t)(p.W * (p.nd ? p.nd : 1)));
  hipLaunchKernelGGL((gte_kernel<MODE_STEP, 4, GTE_HOT_NT, true, STAGE_RAW>), dim3(blocks),
                     dim3(threads), smem, stream, p, vm, fm, wm);
  return hipGetLastError();
}

// Workgroups of this kernel one CU holds at once (registers, LDS): the launch geometry sizes
// the workgroups so that all of them are resident together (gte_api.hip, choose_epw).
int GTE_HOT_NAME(hot_blocks_per_cu)(size_t smem) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &n, gte_kernel<MODE_STEP, 4, GTE_HOT_NT, true, STAGE_RAW>, 64 * GTE_WAVES, smem) != hipSuccess)
    return 0;
  return n;
}

}  // namespace gte
