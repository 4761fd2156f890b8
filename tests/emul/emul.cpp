// emul.cpp — host build (g++) of the SAME per-lane DSP text the HIP kernels run
// (groove_amd/csrc/dsp_core.h + derive.h), looped over voices on the CPU.
//
// DEVELOPMENT / TEST HARNESS ONLY.  It exists so the fp32/f64 arithmetic choices of the
// device code can be checked against the f64 oracle in the GPU-less CI tier.  It is not
// shipped, not linked into libgroove_hip.so, and the product never calls it.
#include "../../groove_amd/csrc/derive.h"
#include <vector>
#include <cstring>
using namespace groove;

struct EmulBank {
  int kind; uint32_t n; double sr; int generic_lfo = 0; int segmented = 1;
  std::vector<WelshParams> wp; std::vector<WelshState> ws; std::vector<WelshCold> wc;
  std::vector<FmParams> fp; std::vector<FmState> fs; std::vector<double> ratio;
  std::vector<SamplerParams> sp; std::vector<SamplerState> ss; std::vector<float> pcm;
};

template <bool FIRST, bool RETUNE>
static void welsh_emul_frame2(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, int mode, float& L, float& R) {
  if (mode == LFO_F32) welsh_frame<FIRST, RETUNE, LFO_F32>(p, s, rc, sc, L, R);
  else if (mode == LFO_F64) welsh_frame<FIRST, RETUNE, LFO_F64>(p, s, rc, sc, L, R);
  else welsh_frame<FIRST, RETUNE, LFO_F64_SMOOTH>(p, s, rc, sc, L, R);
}
static void welsh_emul_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, bool first, bool retune,
                             int mode, float& L, float& R) {
  if (first) { if (retune) welsh_emul_frame2<true, true>(p, s, rc, sc, mode, L, R); else welsh_emul_frame2<true, false>(p, s, rc, sc, mode, L, R); }
  else { if (retune) welsh_emul_frame2<false, true>(p, s, rc, sc, mode, L, R); else welsh_emul_frame2<false, false>(p, s, rc, sc, mode, L, R); }
}

template <bool RETUNE>
static void welsh_emul_segment_frame2(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, int mode, float& L, float& R) {
  if (mode == LFO_F32) welsh_frame<false, RETUNE, LFO_F32, OSC_ANY, OSC_ANY, OSC_ANY, true>(p, s, rc, sc, L, R);
  else if (mode == LFO_F64) welsh_frame<false, RETUNE, LFO_F64, OSC_ANY, OSC_ANY, OSC_ANY, true>(p, s, rc, sc, L, R);
  else welsh_frame<false, RETUNE, LFO_F64_SMOOTH, OSC_ANY, OSC_ANY, OSC_ANY, true>(p, s, rc, sc, L, R);
}
static void welsh_emul_segment_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, bool retune, int mode,
                                     float& L, float& R) {
  if (retune) welsh_emul_segment_frame2<true>(p, s, rc, sc, mode, L, R); else welsh_emul_segment_frame2<false>(p, s, rc, sc, mode, L, R);
}

extern "C" {
// segmented != 0 (default): frames after the first run in boundary-free segments, as in the uniform kernels
void emul_set_segmented(void* h, int on);
// generic_lfo != 0: evaluate the f64 LFO exactly on every frame (the per-lane kernel's choice) instead of the recurrences
void emul_set_generic_lfo(void* h, int on);
void* emul_welsh_create(const groove_welsh_params* p, uint32_t n, uint32_t sr) {
  EmulBank* b = new EmulBank(); b->kind = 0; b->n = n; b->sr = sr;
  b->wp.resize(n); b->ws.assign(n, initial_welsh_state()); b->wc.resize(n);
  for (uint32_t v = 0; v < n; ++v) b->wp[v] = derive_welsh(p[v], sr, b->wc[v]);
  return b;
}
void* emul_fm_create(const groove_fm_params* p, uint32_t n, uint32_t sr) {
  EmulBank* b = new EmulBank(); b->kind = 1; b->n = n; b->sr = sr;
  b->fp.resize(n); b->fs.assign(n, initial_fm_state()); b->ratio.resize(n);
  for (uint32_t v = 0; v < n; ++v) { b->fp[v] = derive_fm(p[v], sr); b->ratio[v] = p[v].ratio; }
  return b;
}
void* emul_sampler_create(const float* pcm, uint64_t frames, const groove_sample_desc* d, uint32_t nd,
                          const groove_sampler_params* p, uint32_t n, uint32_t sr) {
  EmulBank* b = new EmulBank(); b->kind = 2; b->n = n; b->sr = sr;
  b->pcm.assign(pcm, pcm + frames); b->sp.resize(n); b->ss.assign(n, SamplerState{0, 0, 0, 0});
  for (uint32_t v = 0; v < n; ++v) {
    const groove_sample_desc& sd = d[p[v].sample_index < nd ? p[v].sample_index : 0];
    b->sp[v] = SamplerParams{(uint32_t)sd.offset, sd.length, (double)sd.root_hz, p[v].gain, p[v].one_shot};
  }
  return b;
}
void emul_bank_destroy(void* h) { delete (EmulBank*)h; }
void emul_set_generic_lfo(void* h, int on) { ((EmulBank*)h)->generic_lfo = on; }
void emul_set_segmented(void* h, int on) { ((EmulBank*)h)->segmented = on; }
void emul_bank_note_events(void* h, const groove_note_event* ev, uint32_t n_ev) {
  EmulBank* b = (EmulBank*)h;
  for (uint32_t i = 0; i < n_ev; ++i) {
    uint32_t lo = ev[i].voice, hi = ev[i].voice + 1;
    if (ev[i].voice == GROOVE_ALL_VOICES) { lo = 0; hi = b->n; }
    for (uint32_t v = lo; v < hi && v < b->n; ++v) {
      if (b->kind == 0) welsh_note(b->wp[v], b->ws[v], b->wc[v].tune1, b->wc[v].tune2, b->wc[v].fixed1, b->wc[v].fixed2, b->sr, ev[i].key, ev[i].on != 0);
      else if (b->kind == 1) fm_note(b->fp[v], b->fs[v], b->ratio[v], b->sr, ev[i].key, ev[i].on != 0);
      else sampler_note(b->sp[v], b->ss[v], ev[i].key, ev[i].on != 0);
    }
  }
}
// out[2][frames][n] fp32
void emul_bank_render(void* h, uint32_t frames, float* out) {
  EmulBank* b = (EmulBank*)h;
  const uint32_t n = b->n;
  RenderConsts rc{(float)(3.14159265358979323846 / b->sr), (float)(0.49 * b->sr)};
  for (uint32_t v = 0; v < n; ++v) {
    WelshScratch sc{};
    bool retunes = false;
    int mode = LFO_F64;
    if (b->kind == 0) {
      sc = welsh_scratch_init(b->wp[v], rc); retunes = welsh_retunes(b->wp[v]);
      mode = b->generic_lfo ? (welsh_lfo_mode(b->wp[v]) == LFO_F32 ? LFO_F32 : LFO_F64) : welsh_lfo_mode(b->wp[v]);
    }
    float sampler_buf[16] = {};
    uint32_t seg_left = 0; // segmented form (uniform kernels): frames left before the next boundary check
    bool seg_live = false;
    for (uint32_t f = 0; f < frames; ++f) {
      float L, R;
      if (b->kind == 0 && b->segmented && f > 0) { // mirrors run_frames_segmented with a one-lane wave
        if (seg_left == 0) seg_left = welsh_segment_begin(b->wp[v], b->ws[v], seg_live);
        L = R = 0.0f;
        if (seg_live) welsh_emul_segment_frame(b->wp[v], b->ws[v], rc, sc, retunes, mode, L, R);
        else welsh_segment_idle_frame(b->ws[v]);
        --seg_left;
      } else if (b->kind == 0) { // mirrors the kernels' checked form: frame 0 peeled, RETUNE and the LFO mode chosen per voice
        welsh_emul_frame(b->wp[v], b->ws[v], rc, sc, f == 0, retunes, mode, L, R);
      } else if (b->kind == 1) {
        if (f == 0) fm_frame<true>(b->fp[v], b->fs[v], L, R); else fm_frame<false>(b->fp[v], b->fs[v], L, R);
      } else { // mirrors the kernel: chunks of 16 frames (sampler_chunk)
        if ((f & 15u) == 0) sampler_chunk<16>(b->sp[v], b->ss[v], b->pcm.data(), frames - f < 16 ? frames - f : 16, sampler_buf);
        L = R = sampler_buf[f & 15u];
      }
      out[(size_t)f * n + v] = L;
      out[((size_t)frames + f) * n + v] = R;
    }
  }
}
float emul_bitcrush(float x, uint32_t bits) { return bitcrush(x, bits); }
// both forms of the 24 dB coefficient computation (dsp_core.h): out[0..5] two-step, out[6..11] fused
void emul_lp24_coef_both(double ripple, float fc, float sr, double* out) {
  const Lp24Consts c = derive_lp24_consts(ripple);
  const Lp24CoefD a = lp24_widen(lp24_coef_from_fc(c, fc, 3.14159265358979323846f / sr, 0.49f * sr));
  const Lp24CoefD b = lp24_coefd_from_fc(c, fc, 3.14159265358979323846f / sr, 0.49f * sr);
  std::memcpy(out, &a, sizeof a);
  std::memcpy(out + 6, &b, sizeof b);
}
}
