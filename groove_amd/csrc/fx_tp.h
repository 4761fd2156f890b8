// fx_tp.h — time-parallel forms of the IIR effect kernels (BiQuad 12 dB family, 24 dB low-pass as an effect).
//
// fx_biquad_kernel / fx_lp24_kernel (kernels.h) give a (channel, lane) pair to one thread that walks the block's
// frames serially: with 2 x 4,096 lane-channels (config #3) that is 128 wavefronts on 1,024 SIMDs and 33 us of pure
// dependent-chain latency.  The recurrences are linear with coefficients that do not change inside a block, so —
// exactly as welsh_tp.h does for a voice's filter — ONE WAVEFRONT takes one lane-channel, its 64 lanes take 4 frames
// each, build the affine map of their frames, a log-step scan hands every lane its start state, and a second pass
// produces the outputs.  Same f64 arithmetic per frame as the serial kernels (pass B performs their operations in
// their order from a start state that agrees to f64 rounding).  Requires blocks of up to 256 frames.
#pragma once
#include "kernels.h"
#include "welsh_tp.h"

namespace groove {

// (BqAffine and bq_tp_wave — one lane-channel's block on one wavefront — live in welsh_tp.h: the time-parallel Welsh kernel
// applies the same scan to its own output when a BiQuad is fused behind it.)
// Workgroup = 8 adjacent lane-channels of one channel x the block's frames (one per wave).  The block is planar [frame][lane], so a
// lane-channel's frames are 4 n bytes apart: read per wavefront they would cost one 128-byte line per sample.  The
// workgroup therefore moves its [frames][32] tile through LDS with whole-line accesses (two 128-byte rows per wave
// instruction), transposed to tile[lane-channel][frame] (+1 float per row: conflict-free both ways), and its waves take
// the lane-channels in turn from there (one ds_read_b128 = a lane's four frames).
// Two tile widths: 8 lane-channels (one per wave) for small banks, where the number of workgroups is what matters, and
// 32 (four per wave, one after the other) for banks of thousands of lanes, where a tile row is then a whole 128-byte
// line instead of a quarter of one (the 8-wide tile moves every line of the block four times).
constexpr uint32_t kFxTile = 8, kFxTileWide = 32, kFxTpThreads = 512, kFxTileRow = kTpMaxFrames + 4; // rows stay 16-byte aligned
template <uint32_t TILE> struct FxTile { float t[TILE][kFxTileRow]; };
template <uint32_t TILE>
__device__ __forceinline__ void fx_tile_load(FxTile<TILE>& tile, const float* __restrict__ data, uint32_t n, uint32_t frames, uint32_t lane0, uint32_t lanes) {
  const uint32_t c = threadIdx.x % TILE;
  for (uint32_t f = threadIdx.x / TILE; f < frames; f += kFxTpThreads / TILE)
    if (c < lanes) tile.t[c][f] = data[(size_t)f * n + lane0 + c];
}
template <uint32_t TILE>
__device__ __forceinline__ void fx_tile_store(const FxTile<TILE>& tile, float* __restrict__ data, uint32_t n, uint32_t frames, uint32_t lane0, uint32_t lanes) {
  const uint32_t c = threadIdx.x % TILE;
  for (uint32_t f = threadIdx.x / TILE; f < frames; f += kFxTpThreads / TILE)
    if (c < lanes) data[(size_t)f * n + lane0 + c] = tile.t[c][f];
}
// grid: (ceil(n / TILE), 2 channels).  coef: [5][n] f64 (b0 b1 b2 a1 a2); st: [4][2n] f64 (x1 x2 y1 y2); data: the planar block, in place.
template <uint32_t TILE>
__global__ __launch_bounds__(kFxTpThreads) void fx_biquad_tp_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  __shared__ FxTile<TILE> tile;
  __shared__ double s_coef[5][TILE], s_st[4][TILE]; // the tile's coefficients and state: one coalesced load each, not one round trip per lane-channel
  __shared__ float s_wet[TILE];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, ch = blockIdx.y;
  const uint32_t lane0 = blockIdx.x * TILE, lanes = min(TILE, n - lane0);
  float* __restrict__ base = data + ch * ch_stride;
  const size_t tn = 2 * (size_t)n;
  if (threadIdx.x < lanes) {
    const uint32_t c = threadIdx.x, ln = lane0 + c;
#pragma unroll
    for (int k = 0; k < 5; ++k) s_coef[k][c] = coef[(size_t)k * n + ln];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_st[k][c] = st[(size_t)k * tn + ch * n + ln];
    s_wet[c] = wet[ln];
  }
  fx_tile_load(tile, base, n, frames, lane0, lanes);
  __syncthreads();
  const uint32_t n0 = lane * kTpChunk;
  const uint32_t cnt = n0 < frames ? (frames - n0 < kTpChunk ? frames - n0 : kTpChunk) : 0u;
  for (uint32_t c = wave; c < lanes; c += kFxTpThreads / 64) { // this wave's lane-channels, one after the other
    const double b0 = s_coef[0][c], b1 = s_coef[1][c], b2 = s_coef[2][c], a1 = s_coef[3][c], a2 = s_coef[4][c];
    const double sx1 = s_st[0][c], sx2 = s_st[1][c], sy1 = s_st[2][c], sy2 = s_st[3][c];
    const float wm = s_wet[c];
    float xf[kTpChunk];
    {
      const float4 q = *reinterpret_cast<const float4*>(&tile.t[c][n0]);
      xf[0] = q.x; xf[1] = q.y; xf[2] = q.z; xf[3] = q.w;
#pragma unroll
      for (uint32_t j = 0; j < kTpChunk; ++j) if (j >= cnt) xf[j] = 0.0f;
    }
    float o[kTpChunk];
    double ns[4];
    const BiquadCoefD cf{b0, b1, b2, a1, a2};
    const bool holds_end = bq_tp_wave(xf, cnt, lane, frames, cf, sx1, sx2, sy1, sy2, wm, o, ns);
    if (cnt) *reinterpret_cast<float4*>(&tile.t[c][n0]) = make_float4(o[0], o[1], o[2], o[3]); // (frames past the block: never stored)
    if (holds_end) { s_st[0][c] = ns[0]; s_st[1][c] = ns[1]; s_st[2][c] = ns[2]; s_st[3][c] = ns[3]; }
  }
  __syncthreads();
  if (threadIdx.x < lanes && frames) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st[(size_t)k * tn + ch * n + lane0 + threadIdx.x] = s_st[k][threadIdx.x];
  }
  fx_tile_store(tile, base, n, frames, lane0, lanes);
}
// coef: [6][n] f64 (b0 a1 a2 per section); st: [4][2n] f64.
template <uint32_t TILE>
__global__ __launch_bounds__(kFxTpThreads) void fx_lp24_tp_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  __shared__ FxTile<TILE> tile;
  __shared__ double s_coef[6][TILE], s_st[4][TILE];
  __shared__ float s_wet[TILE];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, ch = blockIdx.y;
  const uint32_t lane0 = blockIdx.x * TILE, lanes = min(TILE, n - lane0);
  float* __restrict__ base = data + ch * ch_stride;
  const size_t tn = 2 * (size_t)n;
  if (threadIdx.x < lanes) {
    const uint32_t c = threadIdx.x, ln = lane0 + c;
#pragma unroll
    for (int k = 0; k < 6; ++k) s_coef[k][c] = coef[(size_t)k * n + ln];
#pragma unroll
    for (int k = 0; k < 4; ++k) s_st[k][c] = st[(size_t)k * tn + ch * n + ln];
    s_wet[c] = wet[ln];
  }
  fx_tile_load(tile, base, n, frames, lane0, lanes);
  __syncthreads();
  const uint32_t n0 = lane * kTpChunk;
  const uint32_t cnt = n0 < frames ? (frames - n0 < kTpChunk ? frames - n0 : kTpChunk) : 0u;
  for (uint32_t c = wave; c < lanes; c += kFxTpThreads / 64) {
    const Lp24CoefD cf{s_coef[0][c], s_coef[1][c], s_coef[2][c], s_coef[3][c], s_coef[4][c], s_coef[5][c]};
    const double s_init[4] = {s_st[0][c], s_st[1][c], s_st[2][c], s_st[3][c]};
    const float wm = s_wet[c];
    float xf[kTpChunk];
    {
      const float4 q = *reinterpret_cast<const float4*>(&tile.t[c][n0]);
      xf[0] = q.x; xf[1] = q.y; xf[2] = q.z; xf[3] = q.w;
    }
    Lp24Affine mine;
    lp24_affine_identity(mine);
#pragma unroll
    for (uint32_t j = 0; j < kTpChunk; ++j) if (j < cnt) lp24_affine_push(mine, cf, (double)xf[j]);
    Lp24Affine incl = mine;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
      Lp24Affine other;
      tp_shfl_affine(incl, (int)lane - d, other);
      if ((int)lane >= d) lp24_affine_compose(incl, other);
    }
    double s_end[4], sv[4];
    lp24_affine_mul(incl, s_init, s_end, true);
#pragma unroll
    for (int i = 0; i < 4; ++i) { sv[i] = tp_shfl(s_end[i], (int)lane - 1); if (lane == 0) sv[i] = s_init[i]; }
    float o[kTpChunk];
#pragma unroll
    for (uint32_t j = 0; j < kTpChunk; ++j) {
      o[j] = 0.0f;
      if (j < cnt) {
        o[j] = (float)lp24_step_v(sv, cf, (double)xf[j]);
        if (wm < 1.0f) o[j] = fmaf(o[j], wm, xf[j] * (1.0f - wm));
      }
    }
    if (cnt) *reinterpret_cast<float4*>(&tile.t[c][n0]) = make_float4(o[0], o[1], o[2], o[3]);
    if (frames && lane == (frames - 1) / kTpChunk) { s_st[0][c] = s_end[0]; s_st[1][c] = s_end[1]; s_st[2][c] = s_end[2]; s_st[3][c] = s_end[3]; }
  }
  __syncthreads();
  if (threadIdx.x < lanes && frames) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st[(size_t)k * tn + ch * n + lane0 + threadIdx.x] = s_st[k][threadIdx.x];
  }
  fx_tile_store(tile, base, n, frames, lane0, lanes);
}

} // namespace groove
