// write_bw.hip — what does it cost to WRITE a 2 GB voice block on MI355X?  (round 3, for the materialised million-voice form)
//   A  streaming stores, 16 bytes per lane, consecutive (the ceiling)
//   B  the render kernels' pattern: wavefront w stores 256 bytes (64 lanes x 4) at row f, rows n * 4 bytes apart, frame after frame
//   C  the same with 1 KB per (workgroup, row): four adjacent wavefronts' pieces issued together by one wavefront (16 bytes per lane)
// prints GB/s for each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
__global__ __launch_bounds__(256) void stream16(float4* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void rows256(float* __restrict__ p, uint32_t n, uint32_t frames) { // one voice per lane, frames in sequence
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n) return;
  for (uint32_t ch = 0; ch < 2; ++ch)
    for (uint32_t f = 0; f < frames; ++f) p[((size_t)ch * frames + f) * n + v] = (float)f;
}
__global__ __launch_bounds__(256) void rows1k(float* __restrict__ p, uint32_t n, uint32_t frames) { // wave w of the workgroup stores rows f = w, w + 4, ... whole (1 KB)
  const uint32_t q = threadIdx.x & 63u, w = threadIdx.x >> 6, v0 = blockIdx.x * 256;
  if (v0 + 256 > n) return;
  for (uint32_t ch = 0; ch < 2; ++ch)
    for (uint32_t f = w; f < frames; f += 4)
      *reinterpret_cast<float4*>(p + ((size_t)ch * frames + f) * n + v0 + 4 * q) = make_float4((float)f, 0.f, 0.f, 0.f);
}
int main() {
  const uint32_t n = 1000192, frames = 256; // 3,907 workgroups of 256 voices
  const size_t bytes = (size_t)2 * frames * n * 4;
  float* p;
  CHECK(hipMalloc(&p, bytes));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  auto time = [&](const char* name, auto launch) {
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    std::printf("%-44s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
  };
  time("A streaming 16-byte stores", [&] { hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, (float4*)p, bytes / 16); });
  time("B 256 bytes per wavefront per row (render kernels)", [&] { hipLaunchKernelGGL(rows256, dim3((n + 255) / 256), dim3(256), 0, 0, p, n, frames); });
  time("C 1 KB per wavefront per row", [&] { hipLaunchKernelGGL(rows1k, dim3(n / 256), dim3(256), 0, 0, p, n, frames); });
  time("hipMemsetAsync", [&] { CHECK(hipMemsetAsync(p, 0, bytes, 0)); });
  return 0;
}
