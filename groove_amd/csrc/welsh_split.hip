// welsh_split.hip — the role-split Welsh kernel of mid-size banks (welsh_split.h), its own translation unit.
#define GROOVE_WELSH_CLASS_TU 1
#define GROOVE_WELSH_SPLIT_TU 1
#include "kernels.h"
#include "welsh_split.h"
namespace groove {
void launch_welsh_split(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused) {
  if (fused) hipLaunchKernelGGL(welsh_render_split_kernel<true>, dim3(a.n_wgs), dim3(kSplitThreads), 0, st, a, wg_base);
  else hipLaunchKernelGGL(welsh_render_split_kernel<false>, dim3(a.n_wgs), dim3(kSplitThreads), 0, st, a, wg_base);
}
} // namespace groove
