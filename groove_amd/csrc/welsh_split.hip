// welsh_split.hip — the role-split Welsh kernels of mid-size banks (welsh_split.h); compiled twice, -DGROOVE_WELSH_SPLIT_TU=3
// (front | tangent | back) and =2 (front + tangent | back), each translation unit with its own class-specialised fronts.
#define GROOVE_WELSH_CLASS_TU 1
#ifndef GROOVE_WELSH_SPLIT_TU
#error "compile with -DGROOVE_WELSH_SPLIT_TU=<3 | 2>"
#endif
#include "kernels.h"
#include "welsh_split.h"
namespace groove {
#if GROOVE_WELSH_SPLIT_TU == 3
void launch_welsh_split(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused) {
  if (fused) hipLaunchKernelGGL((welsh_render_split_kernel<true, 3>), dim3(a.n_wgs), dim3(3 * kSplitLanes), 0, st, a, wg_base);
  else hipLaunchKernelGGL((welsh_render_split_kernel<false, 3>), dim3(a.n_wgs), dim3(3 * kSplitLanes), 0, st, a, wg_base);
}
#else
void launch_welsh_split2(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused) {
  if (fused) hipLaunchKernelGGL((welsh_render_split_kernel<true, 2>), dim3(a.n_wgs), dim3(2 * kSplitLanes), 0, st, a, wg_base);
  else hipLaunchKernelGGL((welsh_render_split_kernel<false, 2>), dim3(a.n_wgs), dim3(2 * kSplitLanes), 0, st, a, wg_base);
}
#endif
} // namespace groove
