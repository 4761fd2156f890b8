// welsh_class.hip — the fused, class-specialised uniform Welsh kernel of ONE base kind (compiled
// with -DGROOVE_BASE_KIND=0..5, so the block bodies build in parallel), or (-DGROOVE_BASE_KIND=9)
// the all-kinds kernel of small banks (=8: its block-writing form), or (=10 / 11) the mix kernel of big banks.  See
// kernels.h, "Workgroup KINDS".
#define GROOVE_WELSH_CLASS_TU 1
#if defined(GROOVE_BASE_KIND) && (GROOVE_BASE_KIND == 9 || GROOVE_BASE_KIND == 8)
#define GROOVE_WELSH_ANY_TU 1
#endif
#if defined(GROOVE_BASE_KIND) && (GROOVE_BASE_KIND == 10 || GROOVE_BASE_KIND == 11)
#define GROOVE_WELSH_MIX_TU 1
#endif
#include "kernels.h"
#ifndef GROOVE_BASE_KIND
#error "compile with -DGROOVE_BASE_KIND=<0..5, 8, 9, 10, 11>"
#endif
namespace groove {
#if GROOVE_BASE_KIND == 0
void launch_welsh_uniform_specialised_0(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F32, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F32, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 1
void launch_welsh_uniform_specialised_1(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F32, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F32, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 2
void launch_welsh_uniform_specialised_2(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F64_SMOOTH, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F64_SMOOTH, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 3
void launch_welsh_uniform_specialised_3(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F64_SMOOTH, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F64_SMOOTH, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 4
void launch_welsh_uniform_specialised_4(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F64, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F64, false, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 5
void launch_welsh_uniform_specialised_5(const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (fused) launch_bound(welsh_render_uniform_kernel<true, LFO_F64, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
  else launch_bound(welsh_render_uniform_kernel<false, LFO_F64, true, true>, dim3(a.n_wgs), dim3(kThreads), st, done, a);
}
#elif GROOVE_BASE_KIND == 9
void launch_welsh_uniform_any(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done) {
  launch_bound(welsh_render_uniform_any_kernel<true>, dim3(a.n_wgs), dim3(kThreads), st, done, a, wg_base);
}
#elif GROOVE_BASE_KIND == 8
void launch_welsh_uniform_any_unfused(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done) {
  launch_bound(welsh_render_uniform_any_kernel<false>, dim3(a.n_wgs), dim3(kThreads), st, done, a, wg_base);
}
#elif GROOVE_BASE_KIND == 10
void launch_welsh_uniform_mix(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done) {
  launch_bound(welsh_render_uniform_mix_kernel<true>, dim3(a.n_wgs), dim3(kThreads), st, done, a, wg_base);
}
#elif GROOVE_BASE_KIND == 11
void launch_welsh_uniform_mix_unfused(const UniformArgs& a, const uint8_t* wg_base, hipStream_t st, hipEvent_t done) {
  launch_bound(welsh_render_uniform_mix_kernel<false>, dim3(a.n_wgs), dim3(kThreads), st, done, a, wg_base);
}
#else
#error "GROOVE_BASE_KIND out of range"
#endif
static_assert(wg_base_kind_of(LFO_F32, false) == 0 && wg_base_kind_of(LFO_F32, true) == 1 &&
              wg_base_kind_of(LFO_F64_SMOOTH, false) == 2 && wg_base_kind_of(LFO_F64_SMOOTH, true) == 3, "base kind numbering");
} // namespace groove
