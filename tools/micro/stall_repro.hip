// stall_repro.hip — library-free reproducer attempt for the per-process crawl of DESIGN.md section 7.
//
// What the million-voice path does, reduced to its stream structure: one HIGH-priority stream that carries short
// "reduce" kernels, four NORMAL-priority streams that each carry one long kernel per block (kernels that use scratch,
// like the round-2 class bodies did), optionally three LOW-priority streams, and the pipelined event pattern of
// render_mix_pipelined (groove_hip.hip): per block every kind stream waits for the reduce of two blocks ago, runs its
// kernel and records an event; the high-priority stream waits for the four events, runs the reduce, records its own.
//
//   stall_repro [blocks=200] [kind_streams=4] [placeholder=0] [scratch=1] [low_streams=3] [flat_priorities=0]
//
// Prints one line: total ms, ms per block, the slowest block's ms.  A healthy process takes ~0.35 ms per block; the
// crawl is 10-50 s per block.  tools/micro/stall_repro.sh runs it in N fresh processes under a timeout.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

template <bool SCRATCH>
__global__ __launch_bounds__(256) void long_kernel(float* __restrict__ state, float* __restrict__ rows, uint32_t n, uint32_t iters) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  float priv[48]; // indexed dynamically below: lives in scratch when SCRATCH
  float x = state[v];
#pragma unroll
  for (int i = 0; i < 48; ++i) priv[i] = x + (float)i;
  double y1 = x, y2 = 0.0;
  for (uint32_t f = 0; f < iters; ++f) {
    const double in = (double)priv[SCRATCH ? ((f * 7u + (uint32_t)(x * 3.0f)) % 48u) : (f % 48u) & 0u];
    const double y = 0.1 * in + 1.6 * y1 - 0.64 * y2; // an IIR step, f64 like the real kernels
    y2 = y1; y1 = y;
    if (SCRATCH) priv[(f * 5u) % 48u] = (float)y;
    x = (float)y * 0.5f + x * 0.5f;
  }
  state[v] = x * 1e-3f + 0.5f;
  if (threadIdx.x == 0) rows[blockIdx.x] = (float)y1;
}
__global__ void reduce_kernel(const float* __restrict__ rows, uint32_t n, float* __restrict__ bus) {
  float a = 0.0f;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) a += rows[i];
  atomicAdd(bus, a);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? std::atoi(argv[1]) : 200;
  const int kind_streams = argc > 2 ? std::atoi(argv[2]) : 4;
  const int placeholder = argc > 3 ? std::atoi(argv[3]) : 0;
  const int scratch = argc > 4 ? std::atoi(argv[4]) : 1;
  const int low_streams = argc > 5 ? std::atoi(argv[5]) : 3;
  const int flat = argc > 6 ? std::atoi(argv[6]) : 0;
  CHECK(hipSetDevice(0));
  int lo = 0, hi = 0;
  CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  if (flat) lo = hi = 0;
  hipStream_t ctx, kind[4], low[3], ph = nullptr;
  CHECK(hipStreamCreateWithPriority(&ctx, hipStreamNonBlocking, hi));
  for (int k = 0; k < 4; ++k) {
    if (k < kind_streams) CHECK(hipStreamCreateWithPriority(&kind[k], hipStreamNonBlocking, 0));
    else kind[k] = kind[k - kind_streams];
    if (k == 2 && kind_streams == 3 && placeholder) CHECK(hipStreamCreateWithPriority(&ph, hipStreamNonBlocking, 0));
  }
  for (int k = 0; k < low_streams; ++k) CHECK(hipStreamCreateWithPriority(&low[k], hipStreamNonBlocking, lo));
  const uint32_t per_kind[4] = {250000, 380000, 190000, 180000};
  float *state[4], *rows[4][2], *bus;
  CHECK(hipMalloc(&bus, 4096));
  CHECK(hipMemsetAsync(bus, 0, 4096, ctx));
  for (int k = 0; k < 4; ++k) {
    CHECK(hipMalloc(&state[k], per_kind[k] * 4));
    CHECK(hipMemsetAsync(state[k], 0, per_kind[k] * 4, ctx));
    for (int s = 0; s < 2; ++s) CHECK(hipMalloc(&rows[k][s], (per_kind[k] / 256 + 1) * 4));
  }
  const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
  hipEvent_t ev_fork, ev_render[4][2], ev_reduce[2];
  CHECK(hipEventCreateWithFlags(&ev_fork, flags));
  for (int s = 0; s < 2; ++s) {
    CHECK(hipEventCreateWithFlags(&ev_reduce[s], flags));
    for (int k = 0; k < 4; ++k) CHECK(hipEventCreateWithFlags(&ev_render[k][s], flags));
  }
  CHECK(hipStreamSynchronize(ctx));
  CHECK(hipEventRecord(ev_fork, ctx));
  for (int k = 0; k < 4; ++k) CHECK(hipStreamWaitEvent(kind[k], ev_fork, 0));
  std::vector<hipEvent_t> t(blocks + 1);
  for (auto& e : t) CHECK(hipEventCreate(&e));
  const auto w0 = std::chrono::steady_clock::now();
  CHECK(hipEventRecord(t[0], ctx));
  for (int b = 0; b < blocks; ++b) {
    const int slot = b & 1;
    for (int k = 3; k >= 0; --k) {
      if (b >= 2) CHECK(hipStreamWaitEvent(kind[k], ev_reduce[slot], 0));
      const uint32_t n = per_kind[k];
      if (scratch) hipLaunchKernelGGL(long_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, kind[k], state[k], rows[k][slot], n, 256u);
      else hipLaunchKernelGGL(long_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, kind[k], state[k], rows[k][slot], n, 256u);
      CHECK(hipEventRecord(ev_render[k][slot], kind[k]));
      CHECK(hipStreamWaitEvent(ctx, ev_render[k][slot], 0));
    }
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), 0, ctx, rows[k][slot], per_kind[k] / 256, bus);
    CHECK(hipEventRecord(ev_reduce[slot], ctx));
    CHECK(hipEventRecord(t[b + 1], ctx));
    if (low_streams && (b % 8) == 0) // a little traffic on the low-priority streams, as a mixed project has
      for (int k = 0; k < low_streams; ++k) hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(64), 0, low[k], rows[0][slot], 16u, bus + 8 + k);
  }
  CHECK(hipStreamSynchronize(ctx));
  for (int k = 0; k < low_streams; ++k) CHECK(hipStreamSynchronize(low[k]));
  const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  float worst = 0.0f, total = 0.0f;
  for (int b = 0; b < blocks; ++b) {
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, t[b], t[b + 1]));
    worst = ms > worst ? ms : worst;
    total += ms;
  }
  std::printf("stall_repro blocks=%d kind_streams=%d placeholder=%d scratch=%d low=%d flat=%d: wall %.1f ms, %.3f ms/block, worst block %.3f ms%s\n",
              blocks, kind_streams, placeholder, scratch, low_streams, flat, wall, total / blocks, worst, worst > 100.0f ? "  <-- STALL" : "");
  return 0;
}
