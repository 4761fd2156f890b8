// ticket_probe.hip (round 5) — fx_run_allpass_kernel's scheduling skeleton without the audio: 2,048 workgroups, eight per-XCD
// ticket queues of 256 frames, every item publishes a flag and waits for the flags of frames f - 75, f - 150, f - 220, f - 225,
// f - 295 (bounded).  Alone, and beside a long kernel on another stream.  Prints timeouts and how the items were spread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Sync { unsigned ticket[8]; unsigned done; unsigned timeouts; unsigned processed[8]; unsigned max_spins; };
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u; }
__global__ void spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32); }
__global__ __launch_bounds__(256) void chain(unsigned* flags, Sync* sync, unsigned epoch, unsigned frames, float* sink) {
  __shared__ unsigned s_ticket;
  const unsigned xcc = xcc_id(), X = gridDim.x;
  const unsigned nx = xcc < X ? (X - xcc + 7u) / 8u : 0u, items = nx * frames;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = items ? atomicAdd(&sync->ticket[xcc], 1u) : 0xFFFFFFFFu;
    __syncthreads();
    const unsigned t = s_ticket;
    __syncthreads();
    if (t >= items) break;
    const unsigned f = t / nx, x = xcc + 8u * (t % nx);
    float acc = (float)threadIdx.x;
    for (int k = 0; k < 400; ++k) acc = acc * 1.0001f + 0.5f; // a little work
    sink[(x * frames + f) * 256 + threadIdx.x] = acc;
    unsigned* row = flags + x * frames;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(row + f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(&sync->processed[xcc], 1u); }
    if (threadIdx.x < 64u) {
      const unsigned j = threadIdx.x >> 3, i = threadIdx.x & 7u, back = j * 75u + i * 220u;
      if (back != 0 && back <= f && j <= f / 75u && i <= (f - j * 75u) / 220u) {
        unsigned spins = 0;
        while (__hip_atomic_load(row + (f - back), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 16)) { atomicAdd(&sync->timeouts, 1u); break; }
        }
        atomicMax(&sync->max_spins, spins);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
  }
  if (threadIdx.x == 0 && atomicAdd(&sync->done, 1u) == gridDim.x * gridDim.y - 1u) {
    for (int k = 0; k < 8; ++k) sync->ticket[k] = 0u;
    sync->done = 0u;
    __threadfence();
  }
}
int main() {
  const unsigned X = 8, frames = 256;
  unsigned* flags; Sync* sync; float* sink;
  (void)hipMalloc(&flags, X * frames * 4); (void)hipMalloc(&sync, sizeof(Sync)); (void)hipMalloc(&sink, (size_t)X * frames * 256 * 4);
  (void)hipMemset(flags, 0, X * frames * 4); (void)hipMemset(sync, 0, sizeof(Sync));
  hipStream_t s1, s2; (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned epoch = 0;
  for (int beside = 0; beside < 2; ++beside) {
    for (int rep = 0; rep < 6; ++rep) {
      if (beside) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s2, (unsigned long long)khz / 20ull); // 50 us of every CU, repeatedly
      (void)hipEventRecord(e0, s1);
      hipLaunchKernelGGL(chain, dim3(X, frames), dim3(256), 0, s1, flags, sync, ++epoch, frames, sink);
      (void)hipEventRecord(e1, s1);
      if (hipStreamSynchronize(s1) != hipSuccess) { printf("sync failed\n"); return 1; }
      (void)hipStreamSynchronize(s2);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      Sync h; (void)hipMemcpy(&h, sync, sizeof(h), hipMemcpyDeviceToHost);
      printf("{\"beside_another_kernel\": %d, \"ms\": %.4f, \"timeouts\": %u, \"max_polls\": %u, \"items_by_xcd\": [%u, %u, %u, %u, %u, %u, %u, %u], \"tickets_left\": [%u, %u], \"done\": %u}\n", beside, ms, h.timeouts, h.max_spins,
             h.processed[0], h.processed[1], h.processed[2], h.processed[3], h.processed[4], h.processed[5], h.processed[6], h.processed[7], h.ticket[0], h.ticket[7], h.done);
      (void)hipMemset(&sync->timeouts, 0, 4 * 10);
    }
  }
  return 0;
}
