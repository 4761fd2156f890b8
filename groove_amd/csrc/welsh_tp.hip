// welsh_tp.hip — the time-parallel Welsh kernel (welsh_tp.h: one wavefront per voice, lanes = time), its own
// translation unit so that it builds beside the class-specialised ones.
#define GROOVE_WELSH_CLASS_TU 1
#include "kernels.h"
#include "welsh_tp.h"
#include <cstdlib>
namespace groove {
void launch_welsh_tp(const TpArgs& a, hipStream_t st, bool fused) {
  const dim3 grid(welsh_tp_grid(a.n)), blk(kTpThreads);
  if (fused) hipLaunchKernelGGL(welsh_tp_kernel<true>, grid, blk, 0, st, a);
  else if (a.bq_coef) {
    // (experiment knob, round 3: unused dynamic LDS caps how many of this kernel's 163-VGPR wavefronts a CU takes, leaving
    // registers for the effect kernels that run beside it on the ctx stream — docs/STREAMS.md item 11)
    static const unsigned pad = [] { const char* e = std::getenv("GROOVE_TP_PAD_LDS"); return e ? (unsigned)std::strtoul(e, nullptr, 10) : 0u; }();
    hipLaunchKernelGGL((welsh_tp_kernel<false, true>), grid, blk, pad, st, a);
  }
  else hipLaunchKernelGGL(welsh_tp_kernel<false>, grid, blk, 0, st, a);
}
void launch_fm_tp(const TpArgs& a, hipStream_t st, bool fused) {
  const dim3 grid(welsh_tp_workgroups(a.n)), blk(kTpThreads);
  if (fused) hipLaunchKernelGGL(fm_tp_kernel<true>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(fm_tp_kernel<false>, grid, blk, 0, st, a);
}
void launch_sampler_tp(const TpArgs& a, const float* bank, const InlineEvents& ie, hipStream_t st, bool fused) {
  const dim3 grid(sampler_tp_workgroups(a.n)), blk(kSamplerTpThreads);
  const uint32_t vpw = sampler_tp_vpw(a.n);
  if (fused) hipLaunchKernelGGL(sampler_tp_kernel<true>, grid, blk, 0, st, a, bank, ie, vpw);
  else hipLaunchKernelGGL(sampler_tp_kernel<false>, grid, blk, 0, st, a, bank, ie, vpw);
}
} // namespace groove
