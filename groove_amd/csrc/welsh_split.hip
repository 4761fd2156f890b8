// welsh_split.hip — the role-split Welsh kernels of mid-size banks (welsh_split.h); compiled three times, -DGROOVE_WELSH_SPLIT_TU=3
// (front | tangent | back), =2 (front + tangent | back) and =4 (ctl | osc | tangent + quotients | back), each translation unit with its own class-specialised fronts.
#define GROOVE_WELSH_CLASS_TU 1
#ifndef GROOVE_WELSH_SPLIT_TU
#error "compile with -DGROOVE_WELSH_SPLIT_TU=<4 | 3 | 2>"
#endif
#include "kernels.h"
#include "welsh_split.h"
namespace groove {
#if GROOVE_WELSH_SPLIT_TU == 3
void launch_welsh_split(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_split_kernel<true, 3>, dim3(a.n_wgs), dim3(3 * kSplitLanes), st, done, a, wg_base);
  else launch_bound(welsh_render_split_kernel<false, 3>, dim3(a.n_wgs), dim3(3 * kSplitLanes), st, done, a, wg_base);
}
#elif GROOVE_WELSH_SPLIT_TU == 4
void launch_welsh_split4(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_split4_kernel<true>, dim3(a.n_wgs), dim3(4 * kSplitLanes), st, done, a, wg_base);
  else launch_bound(welsh_render_split4_kernel<false>, dim3(a.n_wgs), dim3(4 * kSplitLanes), st, done, a, wg_base);
}
#else
void launch_welsh_split2(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_split_kernel<true, 2>, dim3(a.n_wgs), dim3(2 * kSplitLanes), st, done, a, wg_base);
  else launch_bound(welsh_render_split_kernel<false, 2>, dim3(a.n_wgs), dim3(2 * kSplitLanes), st, done, a, wg_base);
}
#endif
} // namespace groove
#ifdef GROOVE_SPLIT_PROBE
#define GROOVE_PROBE_NAME2(n) groove_debug_split_probe_read##n
#define GROOVE_PROBE_NAME(n) GROOVE_PROBE_NAME2(n)
extern "C" int GROOVE_PROBE_NAME(GROOVE_WELSH_SPLIT_TU)(unsigned long long out[12], int reset) { // measurement build only
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(groove::g_split_probe), sizeof(groove::g_split_probe)) != hipSuccess) return 3;
  if (reset) { unsigned long long z[12] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(groove::g_split_probe), z, sizeof(z)) != hipSuccess) return 4; }
  return 0;
}
#endif
