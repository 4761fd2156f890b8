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

// y-recurrence of the Direct Form 1 biquad as an affine map of (y1, y2): y = w - a1 y1 - a2 y2, w = b0 x + b1 x1 + b2 x2
struct BqAffine { double m00, m01, m10, m11, z0, z1; }; // (y1, y2)' = M (y1, y2) + z
__device__ __forceinline__ void bq_affine_identity(BqAffine& m) { m.m00 = 1.0; m.m01 = 0.0; m.m10 = 0.0; m.m11 = 1.0; m.z0 = 0.0; m.z1 = 0.0; }
__device__ __forceinline__ void bq_affine_push(BqAffine& m, double a1, double a2, double w) {
  // one more frame: (y1, y2) -> (w - a1 y1 - a2 y2, y1)
  const double n00 = -a1 * m.m00 - a2 * m.m10, n01 = -a1 * m.m01 - a2 * m.m11, nz0 = (w - a1 * m.z0) - a2 * m.z1;
  m.m10 = m.m00; m.m11 = m.m01; m.z1 = m.z0;
  m.m00 = n00; m.m01 = n01; m.z0 = nz0;
}
__device__ __forceinline__ void bq_affine_compose(BqAffine& later, const BqAffine& e) { // later <- later o e
  BqAffine r;
  r.m00 = later.m00 * e.m00 + later.m01 * e.m10; r.m01 = later.m00 * e.m01 + later.m01 * e.m11;
  r.m10 = later.m10 * e.m00 + later.m11 * e.m10; r.m11 = later.m10 * e.m01 + later.m11 * e.m11;
  r.z0 = later.m00 * e.z0 + later.m01 * e.z1 + later.z0;
  r.z1 = later.m10 * e.z0 + later.m11 * e.z1 + later.z1;
  later = r;
}

constexpr int kFxTpWaves = 4; // lane-channels per workgroup
// coef: [5][n] f64 (b0 b1 b2 a1 a2); st: [4][2n] f64 (x1 x2 y1 y2); data: the planar block, in place.
__global__ __launch_bounds__(kFxTpWaves * 64) void fx_biquad_tp_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kFxTpWaves + (threadIdx.x >> 6)));
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, ln = t % n;
  const size_t tn = 2 * (size_t)n;
  const double b0 = coef[ln], b1 = coef[(size_t)n + ln], b2 = coef[(size_t)2 * n + ln], a1 = coef[(size_t)3 * n + ln], a2 = coef[(size_t)4 * n + ln];
  const double sx1 = st[t], sx2 = st[tn + t], sy1 = st[2 * tn + t], sy2 = st[3 * tn + t];
  const float wm = wet[ln];
  float* __restrict__ ptr = data + ch * ch_stride + ln;
  const uint32_t n0 = lane * kTpChunk;
  const uint32_t cnt = n0 < frames ? (frames - n0 < kTpChunk ? frames - n0 : kTpChunk) : 0u;
  float xf[kTpChunk];
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) xf[j] = j < cnt ? ptr[(size_t)(n0 + j) * n] : 0.0f;
  // the two inputs before this lane's first frame: the previous lane's last two, or the state
  const double last1 = (double)xf[kTpChunk - 1], last2 = (double)xf[kTpChunk - 2];
  double px1 = tp_shfl(last1, (int)lane - 1), px2 = tp_shfl(last2, (int)lane - 1);
  if (lane == 0) { px1 = sx1; px2 = sx2; }
  double w[kTpChunk];
  BqAffine mine;
  bq_affine_identity(mine);
  {
    double x1 = px1, x2 = px2;
#pragma unroll
    for (uint32_t j = 0; j < kTpChunk; ++j) {
      const double x = (double)xf[j];
      w[j] = b0 * x + b1 * x1 + b2 * x2;
      if (j < cnt) bq_affine_push(mine, a1, a2, w[j]);
      x2 = x1; x1 = x;
    }
  }
  BqAffine incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    BqAffine o;
    o.m00 = tp_shfl(incl.m00, (int)lane - d); o.m01 = tp_shfl(incl.m01, (int)lane - d);
    o.m10 = tp_shfl(incl.m10, (int)lane - d); o.m11 = tp_shfl(incl.m11, (int)lane - d);
    o.z0 = tp_shfl(incl.z0, (int)lane - d); o.z1 = tp_shfl(incl.z1, (int)lane - d);
    if ((int)lane >= d) bq_affine_compose(incl, o);
  }
  const double e1 = incl.m00 * sy1 + incl.m01 * sy2 + incl.z0, e2 = incl.m10 * sy1 + incl.m11 * sy2 + incl.z1; // (y1, y2) after this lane
  double y1 = tp_shfl(e1, (int)lane - 1), y2 = tp_shfl(e2, (int)lane - 1);
  if (lane == 0) { y1 = sy1; y2 = sy2; }
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) {
    if (j < cnt) {
      const double y = w[j] - a1 * y1 - a2 * y2;
      y2 = y1; y1 = y;
      float o = (float)y;
      if (wm < 1.0f) o = fmaf(o, wm, xf[j] * (1.0f - wm));
      ptr[(size_t)(n0 + j) * n] = o;
    }
  }
  const uint32_t last = (frames - 1) / kTpChunk;
  if (lane == last) { // x1, x2 = the block's last two inputs; y1, y2 = its last two outputs
    const uint32_t c = cnt; // >= 1 in this lane
    const double nx1 = (double)xf[c - 1], nx2 = c >= 2 ? (double)xf[c - 2] : px1;
    st[t] = nx1; st[tn + t] = nx2; st[2 * tn + t] = y1; st[3 * tn + t] = y2;
  }
}
// coef: [6][n] f64 (b0 a1 a2 per section); st: [4][2n] f64.
__global__ __launch_bounds__(kFxTpWaves * 64) void fx_lp24_tp_kernel(
    float* __restrict__ data, uint32_t n, uint32_t frames, size_t ch_stride,
    const double* __restrict__ coef, double* __restrict__ st, const float* __restrict__ wet) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kFxTpWaves + (threadIdx.x >> 6)));
  if (t >= 2 * n) return;
  const uint32_t ch = t / n, ln = t % n;
  const size_t tn = 2 * (size_t)n;
  const Lp24CoefD c{coef[ln], coef[(size_t)n + ln], coef[(size_t)2 * n + ln], coef[(size_t)3 * n + ln], coef[(size_t)4 * n + ln], coef[(size_t)5 * n + ln]};
  const double s_init[4] = {st[t], st[tn + t], st[2 * tn + t], st[3 * tn + t]};
  const float wm = wet[ln];
  float* __restrict__ ptr = data + ch * ch_stride + ln;
  const uint32_t n0 = lane * kTpChunk;
  const uint32_t cnt = n0 < frames ? (frames - n0 < kTpChunk ? frames - n0 : kTpChunk) : 0u;
  float xf[kTpChunk];
  Lp24Affine mine;
  lp24_affine_identity(mine);
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) {
    xf[j] = j < cnt ? ptr[(size_t)(n0 + j) * n] : 0.0f;
    if (j < cnt) lp24_affine_push(mine, c, (double)xf[j]);
  }
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
#pragma unroll
  for (uint32_t j = 0; j < kTpChunk; ++j) {
    if (j < cnt) {
      float o = (float)lp24_step_v(sv, c, (double)xf[j]);
      if (wm < 1.0f) o = fmaf(o, wm, xf[j] * (1.0f - wm));
      ptr[(size_t)(n0 + j) * n] = o;
    }
  }
  if (lane == (frames - 1) / kTpChunk) { st[t] = s_end[0]; st[tn + t] = s_end[1]; st[2 * tn + t] = s_end[2]; st[3 * tn + t] = s_end[3]; }
}

} // namespace groove
