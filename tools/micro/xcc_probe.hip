// xcc_probe.hip (round 5) — what fx_run_allpass_kernel's cross-workgroup hand-over rests on, measured:
//   1. HW_REG_XCC_ID by block index: is workgroup i on XCD i mod 8 (lone kernel; beside another stream's kernel)?
//   2. a producer workgroup stores a row (plain stores) and then a flag; a consumer workgroup ON THE SAME XCD polls the flag
//      and reads the row: which (store, load) scopes make the flag visible, how many polls it takes, is the row right?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/xcc_probe tools/micro/xcc_probe.hip && tools/micro/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u; }
__global__ void where(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
__global__ void spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32); }
// MODE 0: flag store / load relaxed at agent scope;  1: workgroup scope;  2: plain volatile;  3: agent-scope release / acquire
template <int MODE>
__global__ __launch_bounds__(256) void handover(float* rows, unsigned* flags, unsigned* result, unsigned epoch, unsigned n_pairs) {
  // blocks 2p (producer) and 2p + 16 ... : pair p = (block p, block p + 8 * k): same XCD if the mapping is i mod 8
  const unsigned b = blockIdx.x, half = gridDim.x / 2;
  const bool producer = b < half;
  const unsigned p = producer ? b : b - half;
  float* row = rows + (size_t)p * 1024;
  unsigned* flag = flags + p;
  if (producer) {
    unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8); // let the consumer start polling first
    for (int k = threadIdx.x; k < 1024; k += 256) row[k] = (float)(epoch * 1000u + k);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) *(volatile unsigned*)flag = epoch;
      else __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      result[4 * p + 3] = xcc_id();
    }
  } else {
    __shared__ unsigned s_spins;
    if (threadIdx.x == 0) {
      unsigned spins = 0, v = 0;
      for (; spins < 200000; ++spins) {
        if (MODE == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) v = *(volatile unsigned*)flag;
        else v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == epoch) break;
        __builtin_amdgcn_s_sleep(4);
      }
      s_spins = spins;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    unsigned bad = 0;
    for (int k = threadIdx.x; k < 1024; k += 256) bad += row[k] != (float)(epoch * 1000u + k);
    bad = __syncthreads_count(bad != 0);
    if (threadIdx.x == 0) { result[4 * p + 0] = s_spins; result[4 * p + 1] = bad; result[4 * p + 2] = xcc_id(); }
  }
  (void)n_pairs;
}
template <int MODE> void run_mode(const char* name, float* rows, unsigned* flags, unsigned* result, unsigned pairs) {
  std::vector<unsigned> h(4 * pairs);
  unsigned seen = 0, late = 0, bad = 0, cross = 0, max_spins = 0;
  for (unsigned epoch = 1; epoch <= 20; ++epoch) {
    hipLaunchKernelGGL(handover<MODE>, dim3(2 * pairs), dim3(256), 0, 0, rows, flags, result, epoch, pairs);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    (void)hipMemcpy(h.data(), result, h.size() * 4, hipMemcpyDeviceToHost);
    for (unsigned p = 0; p < pairs; ++p) {
      if (h[4 * p] < 200000) ++seen; else ++late;
      bad += h[4 * p + 1] != 0; cross += h[4 * p + 2] != h[4 * p + 3];
      if (h[4 * p] < 200000 && h[4 * p] > max_spins) max_spins = h[4 * p];
    }
  }
  printf("{\"mode\": \"%s\", \"pairs\": %u, \"flag_seen\": %u, \"flag_never_seen\": %u, \"rows_wrong\": %u, \"pairs_on_different_xcds\": %u, \"max_polls\": %u}\n", name, 20 * pairs, seen, late, bad, cross, max_spins);
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 4096 * 4);
  std::vector<unsigned> h(4096);
  hipLaunchKernelGGL(where, dim3(2048), dim3(256), 0, 0, d); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
  unsigned ok = 0; for (unsigned i = 0; i < 2048; ++i) ok += h[i] == (i & 7u);
  printf("{\"xcc_of_block_is_i_mod_8\": \"%u of 2048\", \"first_16\": [", ok); for (int i = 0; i < 16; ++i) printf("%u%s", h[i], i < 15 ? ", " : "]}\n");
  hipStream_t s2; (void)hipStreamCreate(&s2);
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipLaunchKernelGGL(spin, dim3(700), dim3(256), 0, s2, (unsigned long long)khz * 3ull); // ~3 ms on another stream
  hipLaunchKernelGGL(where, dim3(2048), dim3(256), 0, 0, d); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
  ok = 0; for (unsigned i = 0; i < 2048; ++i) ok += h[i] == (i & 7u);
  printf("{\"beside_another_kernel\": \"%u of 2048\"}\n", ok);
  const unsigned pairs = 64; // producers: blocks 0..63, consumers: blocks 64..127: pair p = (p, p + 64): same i mod 8
  float* rows; unsigned *flags, *result;
  (void)hipMalloc(&rows, (size_t)pairs * 1024 * 4); (void)hipMalloc(&flags, pairs * 4); (void)hipMalloc(&result, 4 * pairs * 4);
  (void)hipMemset(flags, 0, pairs * 4);
  run_mode<0>("relaxed, agent scope", rows, flags, result, pairs);
  (void)hipMemset(flags, 0, pairs * 4);
  run_mode<1>("relaxed, workgroup scope", rows, flags, result, pairs);
  (void)hipMemset(flags, 0, pairs * 4);
  run_mode<2>("volatile", rows, flags, result, pairs);
  (void)hipMemset(flags, 0, pairs * 4);
  run_mode<3>("release / acquire, agent scope", rows, flags, result, pairs);
  return 0;
}
