// diag.h — every diagnostic of the render path in ONE place.  Nothing here changes what a kernel computes.
//
// Always compiled in (the product build):
//   DiagCounters        one small device record per context.  `zero_segments` counts how often the wave minimum of
//                       frames-to-next-boundary over a wave's ACTIVE lanes came back 0 (kernels.h run_frames_segmented,
//                       welsh_split.h).  welsh_segment_begin promises >= 1 for any consistent envelope record, so the
//                       count must stay 0 for the life of a context: groove_debug_info reports it, the GPU tests and
//                       bench.py assert it.  Cost: one scalar compare + branch per segment.  `fast_table_misses`: the same kind of
//                       assertion for the FAST copies of the block bodies (kernels.h): must stay 0, reported and asserted alike.
//
// Diagnostic builds (make EXTRA=-D...; never shipped, never set by bench.py or the tests' product library):
//   GROOVE_DIAG_SHADOW_IN_MIN   round 3's behaviour: the shadow lanes of padding waves take part in the wave minimum.
//                               Reproduces the stall of DESIGN.md section 7 as counted events instead of an endless
//                               loop, and records the offending lanes (tools/zero_segment_hunt.py).
//   GROOVE_HEARTBEAT            workgroups started / finished by the per-kind kernels, counted in coherent host memory
//                               the host can read while the device is stuck (wait_deadline prints them).
//   GROOVE_TP_PROBE, GROOVE_SPLIT_PROBE   cycle probes of the time-parallel and role-split kernels (welsh_tp.h,
//                               welsh_split.h; tools/tp_probe.py, tools/split_probe.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace groove {

constexpr uint32_t kDiagRecords = 16;
struct DiagRecord {
  uint32_t wg, wave, lane, active, count, frame; // workgroup of the list, virtual wave, lane, lane < count, the wave's voice count, frame at which the segment began
  uint32_t amp_state, amp_n, amp_N, fil_state, fil_n, fil_N;
};
struct DiagCounters {
  uint32_t zero_segments;     // product: wave minimum over ACTIVE lanes == 0 (must stay 0)
  uint32_t shadow_zero_lanes; // GROOVE_DIAG_SHADOW_IN_MIN: lanes whose own frames-to-boundary was 0
  uint32_t shadow_zero_waves; // GROOVE_DIAG_SHADOW_IN_MIN: wave minima that came back 0 because of them
  uint32_t records;           // DiagRecord slots claimed (the first kDiagRecords are kept)
  uint32_t fast_table_misses; // product: a wave that took the FAST copy of its body (kernels.h welsh_wave_tables_up) found a look-ahead table down at a segment start (must stay 0)
  uint32_t fast_waves;        // waves that took the FAST copy of their body, counted only while bit 2 of the look-ahead word is set (groove_set_look_ahead(7): the tests ask whether the path they mean to test ran)
  DiagRecord rec[kDiagRecords];
};

#if defined(__HIPCC__)
// The guard's counter: called by ONE lane of a wave whose minimum was 0.
__device__ __forceinline__ void diag_count_zero_segment(uint32_t* diag) {
  if (diag) atomicAdd(diag + 0, 1u);
}
__device__ __forceinline__ void diag_count_fast_wave(uint32_t* diag) {
  if (diag) atomicAdd(diag + offsetof(DiagCounters, fast_waves) / 4, 1u);
}
// ... and the FAST bodies' assertion: called by one lane of a wave whose tables were promised for the block and are not up.
__device__ __forceinline__ void diag_count_fast_table_miss(uint32_t* diag) {
  if (diag) atomicAdd(diag + offsetof(DiagCounters, fast_table_misses) / 4, 1u);
}
#endif

} // namespace groove
