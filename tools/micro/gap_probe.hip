// gap_probe.hip — what sits between two consecutive kernels of a HIP stream on MI355X (round 3).
// A kernel that spins for a fixed number of clock ticks is launched N times in several stream / event patterns; the time per
// launch minus the spin is the inter-kernel cost of the pattern.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gap_probe tools/micro/gap_probe.hip && /tmp/gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, float* out) { // realtime counter: 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 400, grid = 1024;             // 1,024 workgroups of 256: every CU busy
  const unsigned long long T = 5000;          // 50 us
  float* d; CK(hipMalloc(&d, 4));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t a, b, c; CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, 0)); CK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, 0));
  std::vector<hipEvent_t> ev(N + 8), ev2(N + 8);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  for (auto& e : ev2) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  hipEvent_t old; CK(hipEventCreateWithFlags(&old, hipEventDisableTiming | hipEventDisableSystemFence));
  CK(hipEventRecord(old, a)); CK(hipDeviceSynchronize());
  auto run = [&](const char* name, auto body) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      for (int i = 0; i < N; ++i) if (body(i)) return 1;
      CK(hipDeviceSynchronize());
      const double per = (now() - t0) / N * 1e6;
      if (rep) printf("%-78s %7.2f us per launch = spin 50 + %6.2f\n", name, per, per - 50.0);
    }
    return 0;
  };
  if (run("one stream, back to back", [&](int) -> int { hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d); return 0; })) return 1;
  if (run("one stream, hipEventRecord after every kernel", [&](int i) -> int { hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d); return hipEventRecord(ev[i], b) != hipSuccess; })) return 1;
  if (run("one stream, the event bound to the dispatch (hipExtLaunchKernelGGL)", [&](int i) -> int { hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, nullptr, ev[i], 0, T, d); return 0; })) return 1;
  if (run("one stream, hipStreamWaitEvent on a long-completed event of another stream before every kernel", [&](int) -> int { if (hipStreamWaitEvent(b, old, 0) != hipSuccess) return 1; hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d); return 0; })) return 1;
  if (run("two streams alternating, each kernel waits for the other stream's previous one (record + wait)", [&](int i) -> int {
        hipStream_t s = (i & 1) ? c : b;
        if (i && hipStreamWaitEvent(s, ev[i - 1], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, s, T, d);
        return hipEventRecord(ev[i], s) != hipSuccess; })) return 1;
  if (run("render-ahead shape: side stream long kernel, ctx stream waits for it, short kernel, record; side waits 2 back", [&](int i) -> int {
        if (i >= 2 && hipStreamWaitEvent(b, ev2[i - 2], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d);
        if (hipEventRecord(ev[i], b) != hipSuccess) return 1;
        if (hipStreamWaitEvent(a, ev[i], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 400ull, d);
        return hipEventRecord(ev2[i], a) != hipSuccess; })) return 1;
  if (run("the same, the side kernels half the chip (512 workgroups): two can overlap", [&](int i) -> int {
        if (i >= 2 && hipStreamWaitEvent(b, ev2[i - 2], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(grid / 2), dim3(256), 0, b, T, d);
        if (hipEventRecord(ev[i], b) != hipSuccess) return 1;
        if (hipStreamWaitEvent(a, ev[i], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 400ull, d);
        return hipEventRecord(ev2[i], a) != hipSuccess; })) return 1;
  if (run("render-ahead shape, HOST-paced: the host waits for the event 3 back (hipEventSynchronize), no device wait on the side stream", [&](int i) -> int {
        if (i >= 3 && hipEventSynchronize(ev2[i - 3]) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d);
        if (hipEventRecord(ev[i], b) != hipSuccess) return 1;
        if (hipStreamWaitEvent(a, ev[i], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 400ull, d);
        return hipEventRecord(ev2[i], a) != hipSuccess; })) return 1;
  if (run("render-ahead shape, host-paced by hipEventQuery polling 3 back", [&](int i) -> int {
        if (i >= 3) while (hipEventQuery(ev2[i - 3]) == hipErrorNotReady) {}
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, b, T, d);
        if (hipEventRecord(ev[i], b) != hipSuccess) return 1;
        if (hipStreamWaitEvent(a, ev[i], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, 400ull, d);
        return hipEventRecord(ev2[i], a) != hipSuccess; })) return 1;
  if (run("ctx-bound shape: ctx stream = wait(side kernel i) + 40 us kernel; side stream 30 us kernels back to back", [&](int i) -> int {
        hipLaunchKernelGGL(spin, dim3(grid / 2), dim3(256), 0, b, 3000ull, d);
        if (hipEventRecord(ev[i], b) != hipSuccess) return 1;
        if (hipStreamWaitEvent(a, ev[i], 0) != hipSuccess) return 1;
        hipLaunchKernelGGL(spin, dim3(grid / 2), dim3(256), 0, a, 4000ull, d);
        return 0; })) return 1;
  printf("done\n");
  return 0;
}
