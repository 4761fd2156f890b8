// welsh_tp.hip — the time-parallel Welsh kernel (welsh_tp.h: one wavefront per voice, lanes = time), its own
// translation unit so that it builds beside the class-specialised ones.
#define GROOVE_WELSH_CLASS_TU 1
#include "kernels.h"
#include "welsh_tp.h"
#include <cstdlib>
#include <hip/hip_ext.h>
namespace groove {
// `done` (optional): an event that completes with this kernel, bound to the dispatch's own completion signal
// (hipExtLaunchKernelGGL) instead of recorded behind it (a barrier packet of its own)
template <class K>
static void tp_launch(K kernel, dim3 grid, dim3 blk, unsigned lds, hipStream_t st, hipEvent_t done, const TpArgs& a) {
  if (done) hipExtLaunchKernelGGL(kernel, grid, blk, lds, st, nullptr, done, 0, a);
  else hipLaunchKernelGGL(kernel, grid, blk, lds, st, a);
}
template <int VPW>
static void launch_welsh_tp_vpw(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  const dim3 grid(welsh_tp_grid(a.n, VPW)), blk(kTpThreads);
  if (a.full_coef) { // (rare: the resonance routing)
    if (fused) tp_launch(welsh_tp_kernel<true, false, true, VPW>, grid, blk, 0, st, done, a);
    else if (a.bq_coef) tp_launch(welsh_tp_kernel<false, true, true, VPW>, grid, blk, 0, st, done, a);
    else tp_launch(welsh_tp_kernel<false, false, true, VPW>, grid, blk, 0, st, done, a);
  }
  else if (fused) tp_launch(welsh_tp_kernel<true, false, false, VPW>, grid, blk, 0, st, done, a);
  else if (a.bq_coef) {
    tp_launch(welsh_tp_kernel<false, true, false, VPW>, grid, blk, 0, st, done, a);
  }
  else tp_launch(welsh_tp_kernel<false, false, false, VPW>, grid, blk, 0, st, done, a);
}
void launch_welsh_tp(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (a.vpw == 2) launch_welsh_tp_vpw<2>(a, st, fused, done);
  else launch_welsh_tp_vpw<1>(a, st, fused, done);
}
template <int VPW>
static void launch_fm_tp_vpw(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  const dim3 grid(welsh_tp_workgroups(a.n, VPW)), blk(kTpThreads);
  if (fused) tp_launch(fm_tp_kernel<true, VPW>, grid, blk, 0, st, done, a);
  else tp_launch(fm_tp_kernel<false, VPW>, grid, blk, 0, st, done, a);
}
void launch_fm_tp(const TpArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  if (a.vpw == 4) launch_fm_tp_vpw<4>(a, st, fused, done);
  else if (a.vpw == 2) launch_fm_tp_vpw<2>(a, st, fused, done);
  else launch_fm_tp_vpw<1>(a, st, fused, done);
}
void launch_sampler_tp(const TpArgs& a, const float* bank, const InlineEvents& ie, hipStream_t st, bool fused, hipEvent_t done) {
  const uint32_t vpw = a.vpw > 1 ? a.vpw : sampler_tp_vpw(a.n); // (a.vpw: the deferred form's choice)
  const dim3 grid(sampler_tp_workgroups(a.n, vpw)), blk(kSamplerTpThreads);
  if (fused) launch_bound(sampler_tp_kernel<true>, grid, blk, st, done, a, bank, ie, vpw);
  else launch_bound(sampler_tp_kernel<false>, grid, blk, st, done, a, bank, ie, vpw);
}
void launch_tp_mixed(const TpMixedArgs& m, const InlineEvents& ie, uint32_t grid, hipStream_t st, hipEvent_t done) {
  const dim3 g(grid), blk(kTpThreads);
  const bool w2 = m.welsh.n_wg && m.welsh.vpw == 2, f4 = m.fm.n_wg && m.fm.vpw == 4;
  if (w2 && f4) launch_bound(tp_mixed_kernel<2, 4>, g, blk, st, done, m, ie);
  else if (w2) launch_bound(tp_mixed_kernel<2, 1>, g, blk, st, done, m, ie);
  else if (f4) launch_bound(tp_mixed_kernel<1, 4>, g, blk, st, done, m, ie);
  else launch_bound(tp_mixed_kernel<1, 1>, g, blk, st, done, m, ie);
}
} // namespace groove
#ifdef GROOVE_TP_PROBE
extern "C" int groove_debug_tp_probe_read(unsigned long long out[16], int reset) { // measurement build only (tools/tp_probe.py)
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(groove::g_tp_probe), sizeof(groove::g_tp_probe)) != hipSuccess) return 3;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(groove::g_tp_probe), z, sizeof(z)) != hipSuccess) return 4; }
  return 0;
}
#endif
