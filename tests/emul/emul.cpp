// emul.cpp — host build (g++) of the SAME per-lane DSP text the HIP kernels run
// (groove_amd/csrc/dsp_core.h + derive.h), looped over voices on the CPU.
//
// DEVELOPMENT / TEST HARNESS ONLY.  It exists so the fp32/f64 arithmetic choices of the
// device code can be checked against the f64 oracle in the GPU-less CI tier.  It is not
// shipped, not linked into libgroove_hip.so, and the product never calls it.
#define GROOVE_EMUL_F32_FILTER_HOOK 1
namespace groove { int groove_emul_f32_filter = 0; } // 0: the product's f64 recurrence; 1 / 2: the fp32 forms of dsp_core.h's experiment hook
#include "../../groove_amd/csrc/derive.h"
#include "../../groove_amd/csrc/welsh_tp.h"
#include <vector>
#include <cstring>
using namespace groove;

struct EmulBank {
  int kind; uint32_t n; double sr; int generic_lfo = 0; int segmented = 1; int time_parallel = 0; int role_split = 0; int f32_kind = 0; int lfo_look_ahead = 0;
  uint32_t min_seg = 0xFFFFFFFFu; // the smallest value welsh_segment_begin has returned for any voice of this bank (must stay >= 1)
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

// the per-kind kernels' fp32-filter copy of the block (kernels.h welsh_block<..., F32OK>, WF_FILTER_F32 patches): frame 0 checked,
// then hoisted segments, the filter's recurrence in fp32
template <bool FIRST, bool RETUNE>
static void welsh_emul_f32_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, int mode, float& L, float& R) {
  if (mode == LFO_F32) welsh_frame<FIRST, RETUNE, LFO_F32, OSC_ANY, OSC_ANY, OSC_ANY, !FIRST, false, !FIRST, true>(p, s, rc, sc, L, R);
  else welsh_frame<FIRST, RETUNE, LFO_F64_SMOOTH, OSC_ANY, OSC_ANY, OSC_ANY, !FIRST, false, !FIRST, true>(p, s, rc, sc, L, R);
}
template <bool RETUNE>
static void welsh_emul_segment_frame2(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, int mode, float& L, float& R) {
  // the uniform kernels' form: boundary-free segment, envelope counters hoisted (welsh_frame<..., HOIST>)
  if (mode == LFO_F32) welsh_frame<false, RETUNE, LFO_F32, OSC_ANY, OSC_ANY, OSC_ANY, true, false, true>(p, s, rc, sc, L, R);
  else if (mode == LFO_F64) welsh_frame<false, RETUNE, LFO_F64, OSC_ANY, OSC_ANY, OSC_ANY, true, false, true>(p, s, rc, sc, L, R);
  else welsh_frame<false, RETUNE, LFO_F64_SMOOTH, OSC_ANY, OSC_ANY, OSC_ANY, true, false, true>(p, s, rc, sc, L, R);
}
// ... and a frame of a segment whose LFO comes from the wave's look-ahead table (kernels.h "LFO look-ahead"; a one-lane wave always
// agrees with itself): `mod` evaluated exactly at this frame's phase, handed in
template <bool RETUNE, bool F32FILT>
static void welsh_emul_ltab_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, uint64_t phase, float& L, float& R) {
  welsh_frame<false, RETUNE, LFO_F64_SMOOTH, OSC_ANY, OSC_ANY, OSC_ANY, true, false, true, F32FILT>(p, s, rc, sc, L, R, 0u, 1u, welsh_lfo_mod_exact<OSC_ANY>(p, phase));
}
static void welsh_emul_segment_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, bool retune, int mode,
                                     float& L, float& R) {
  if (retune) welsh_emul_segment_frame2<true>(p, s, rc, sc, mode, L, R); else welsh_emul_segment_frame2<false>(p, s, rc, sc, mode, L, R);
}

// The role-split kernel's frame (csrc/welsh_split.h), role by role with what travels between the roles through LDS on the
// device: role A = welsh_frame_front -> {sum (NaN: silent), gain} and the cutoff percent (NaN: no retune); role B = the tangent of
// the cutoff, negated above SR/4 (NaN: coefficients stand); role C = coefficients from the tangent, the filter step, the gains.
// `coef` is role C's running coefficient set (welsh_scratch_init's at the start of a block).
template <bool FIRST, bool RETUNE, int MODE, bool SEG>
static void welsh_emul_split_frame3(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, Lp24CoefD& coef, float& L, float& R) {
  const float kNan = __builtin_nanf("");
  float sum = 0.0f, g = 0.0f, pct = 0.0f, lfo = 0.0f;
  bool retune = false;
  const bool ok = welsh_frame_front<FIRST, RETUNE, MODE, OSC_ANY, OSC_ANY, OSC_ANY, SEG, false, SEG>(p, s, sc, sum, g, pct, retune, lfo);
  const float ac_sum = ok ? sum : kNan, ab_pct = (RETUNE && ok && retune) ? pct : kNan;     // role A's LDS records
  float bc_t = kNan;                                                                          // role B
  if (RETUNE) { bool hi; const float tj = lp24_t_from_pct(ab_pct, rc, hi); bc_t = (ab_pct == ab_pct) ? (hi ? -tj : tj) : kNan; }
  L = 0.0f; R = 0.0f;                                                                         // role C
  if (ac_sum == ac_sum) {
    if (RETUNE && bc_t == bc_t) coef = lp24_coefd_from_t(p.fc, fabsf(bc_t), bc_t < 0.0f, (p.flags & WF_COEF_WIDE) != 0);
    welsh_frame_back<false>(p, s.filt, coef, ac_sum, g, L, R);
  }
}
// The four-role form: the front in two halves (welsh_frame_ctl: envelopes and LFO -> gain, percent, `mod`; welsh_frame_osc:
// the oscillators), role B also takes the fp32 quotients of the coefficients (lp24_coefq_from_t), role C widens them.
template <bool FIRST, bool RETUNE, int MODE, bool SEG>
static void welsh_emul_split_frame4(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, Lp24CoefD& coef, float& L, float& R) {
  const float kNan = __builtin_nanf("");
  float g = 0.0f, pct = 0.0f;
  double mod = 0.0;
  bool retune = false, first = false;
  const bool ok = welsh_frame_ctl<FIRST, RETUNE, MODE, OSC_ANY, SEG, SEG>(p, s, sc, g, pct, retune, mod, first);          // role A1
  const float ab_pct = (RETUNE && ok && retune) ? pct : kNan;
  const float ac_sum = ok ? welsh_frame_osc<MODE, OSC_ANY, OSC_ANY, false>(p, s, MODE == LFO_F64_SMOOTH, mod, first) : kNan; // role A2
  Lp24CoefQ q{kNan, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                                                                          // role B
  bool hi = false;
  if (RETUNE && ab_pct == ab_pct) { const float t = lp24_t_from_pct(ab_pct, rc, hi); q = lp24_coefq_from_t(p.fc, t, hi, (p.flags & WF_COEF_WIDE) != 0); if (hi) q.qa = -q.qa; }
  L = 0.0f; R = 0.0f;                                                                                                        // role C
  if (ac_sum == ac_sum) {
    if (RETUNE && q.ba == q.ba) { const bool up = q.qa < 0.0f; q.qa = fabsf(q.qa); coef = lp24_coefd_from_q(q, up, (p.flags & WF_COEF_WIDE) != 0); }
    welsh_frame_back<false>(p, s.filt, coef, ac_sum, g, L, R);
  }
}
static int g_roles = 3;
template <bool FIRST, bool SEG>
static void welsh_emul_split_frame(const WelshParams& p, WelshState& s, const RenderConsts& rc, WelshScratch& sc, Lp24CoefD& coef, bool retune, int mode,
                                   float& L, float& R) {
  if (g_roles == 4) {
    if (mode == LFO_F32) { if (retune) welsh_emul_split_frame4<FIRST, true, LFO_F32, SEG>(p, s, rc, sc, coef, L, R); else welsh_emul_split_frame4<FIRST, false, LFO_F32, SEG>(p, s, rc, sc, coef, L, R); }
    else { if (retune) welsh_emul_split_frame4<FIRST, true, LFO_F64_SMOOTH, SEG>(p, s, rc, sc, coef, L, R); else welsh_emul_split_frame4<FIRST, false, LFO_F64_SMOOTH, SEG>(p, s, rc, sc, coef, L, R); }
    return;
  }
  if (mode == LFO_F32) { if (retune) welsh_emul_split_frame3<FIRST, true, LFO_F32, SEG>(p, s, rc, sc, coef, L, R); else welsh_emul_split_frame3<FIRST, false, LFO_F32, SEG>(p, s, rc, sc, coef, L, R); }
  else { if (retune) welsh_emul_split_frame3<FIRST, true, LFO_F64_SMOOTH, SEG>(p, s, rc, sc, coef, L, R); else welsh_emul_split_frame3<FIRST, false, LFO_F64_SMOOTH, SEG>(p, s, rc, sc, coef, L, R); }
}

// The time-parallel form (welsh_tp.h) with a loop over 64 "lanes" in place of the wavefront: the per-lane
// functions are the kernel's, the cross-lane steps (prefix sums, max-scan, affine-map scan) plain loops.
static void welsh_tp_render_voice(const WelshParams& p, WelshState& state, const RenderConsts& rc, uint32_t frames, float* outL, float* outR) {
  const WelshState s0 = state;
  const bool first0 = (s0.vflags & VF_FIRST) != 0;
  const uint32_t live_total = env_idle_at(s0.amp, p.amp, frames);
  const bool retunes = welsh_retunes(p), scans = welsh_tp_scans(p);
  std::vector<float> nz[3];
  OscState nz_end[3] = {s0.o1, s0.o2, s0.lfo};
  for (int o = 0; o < 3; ++o) {
    nz[o].assign(kTpMaxFrames, 0.0f);
    if (welsh_tp_noise(p, o)) for (uint32_t j = 0; j < live_total; ++j) nz[o][j] = noise_tick(nz_end[o]);
  }
  struct Lane {
    WelshState s; uint32_t n0, cnt; uint64_t ph1[kTpChunk], ph2[kTpChunk], inc1[kTpChunk], loc1[kTpChunk], loc2[kTpChunk], run1, run2;
    float x[kTpChunk], amp[kTpChunk]; Lp24CoefD coef[kTpChunk]; bool lives[kTpChunk]; Lp24Affine mine, incl; double s_end[4];
  };
  std::vector<Lane> L(kTpLanes);
  for (uint32_t lane = 0; lane < kTpLanes; ++lane) {
    Lane& l = L[lane];
    l.n0 = lane * kTpChunk;
    l.cnt = l.n0 < frames ? (frames - l.n0 < kTpChunk ? frames - l.n0 : kTpChunk) : 0u;
    l.s = s0;
    env_seek(l.s.amp, p.amp, l.n0 < frames ? l.n0 : 0u);
    env_seek(l.s.fil, p.fil, l.n0 < frames ? l.n0 : 0u);
    const uint32_t live_before = l.n0 < live_total ? l.n0 : live_total;
    const uint64_t adv = (uint64_t)(live_before - ((first0 && live_before >= 1u) ? 1u : 0u));
    l.s.lfo.phase = s0.lfo.phase + adv * p.lfo_inc;
    l.s.o1.phase = s0.o1.phase + adv * s0.o1_inc;
    l.s.o2.phase = s0.o2.phase + adv * s0.o2_inc;
    if (live_before >= 1u) l.s.vflags = 0;
    l.run1 = l.run2 = 0;
    for (uint32_t j = 0; j < kTpChunk; ++j) { l.ph1[j] = l.ph2[j] = l.inc1[j] = 0; }
    if (scans) {
      uint64_t lph = l.s.lfo.phase;
      for (uint32_t j = 0; j < kTpChunk; ++j) {
        const uint32_t f = l.n0 + j;
        const bool live = j < l.cnt && f < live_total, is_first = first0 && f == 0;
        uint64_t i1 = 0, i2 = 0;
        if (live) {
          if (!is_first) lph += p.lfo_inc;
          if (!is_first) welsh_tp_incs(p, s0, lph, nz[2][f < kTpMaxFrames ? f : 0], i1, i2);
        }
        l.inc1[j] = i1; l.run1 += i1; l.run2 += i2; l.loc1[j] = l.run1; l.loc2[j] = l.run2;
      }
    }
  }
  if (scans) {
    uint64_t base1 = 0, base2 = 0;
    int wprev = -1;
    std::vector<uint64_t> sum2(kTpMaxFrames, 0);
    for (uint32_t lane = 0; lane < kTpLanes; ++lane) { // prefix sums first (the gather below reads other lanes' entries)
      Lane& l = L[lane];
      for (uint32_t j = 0; j < kTpChunk; ++j) { l.ph1[j] = s0.o1.phase + base1 + l.loc1[j]; l.loc2[j] += base2; if (j < l.cnt) sum2[l.n0 + j] = l.loc2[j]; }
      base1 += l.run1; base2 += l.run2;
    }
    for (uint32_t lane = 0; lane < kTpLanes; ++lane) {
      Lane& l = L[lane];
      int w = wprev;
      for (uint32_t j = 0; j < kTpChunk; ++j) {
        if (l.ph1[j] < l.inc1[j]) w = (int)(l.n0 + j);
        if (p.flags & WF_SYNC) { if (j < l.cnt) l.ph2[j] = w >= 0 ? l.loc2[j] - sum2[w] : s0.o2.phase + l.loc2[j]; }
        else l.ph2[j] = s0.o2.phase + l.loc2[j];
      }
      wprev = w;
    }
  }
  for (uint32_t lane = 0; lane < kTpLanes; ++lane) {
    Lane& l = L[lane];
    Lp24CoefD cur = lp24_coefd_from_fc(p.fc, p.cutoff_hz, rc.pi_over_sr, rc.fc_max);
    float prev_pct = __builtin_nanf("");
    lp24_affine_identity(l.mine);
    for (uint32_t j = 0; j < kTpChunk; ++j) {
      const uint32_t f = l.n0 + j;
      l.x[j] = 0.0f; l.amp[j] = 0.0f; l.lives[j] = false;
      if (j < l.cnt) {
        env_tick(l.s.amp, p.amp);
        env_tick(l.s.fil, p.fil);
        if (f < live_total) {
          l.lives[j] = true;
          const bool is_first = first0 && f == 0;
          float t_unused = 0.0f; bool hi_unused = false;
          if (retunes) welsh_tp_frame<true>(p, l.s, rc, is_first, scans, l.ph1[j], l.ph2[j], nz[0][f], nz[1][f], nz[2][f], cur, prev_pct, l.x[j], l.amp[j], t_unused, hi_unused);
          else welsh_tp_frame<false>(p, l.s, rc, is_first, scans, l.ph1[j], l.ph2[j], nz[0][f], nz[1][f], nz[2][f], cur, prev_pct, l.x[j], l.amp[j], t_unused, hi_unused);
          l.s.vflags = 0;
          lp24_affine_push(l.mine, cur, (double)l.x[j]);
        } else if (scans) { l.s.o1.phase = l.ph1[j]; l.s.o2.phase = l.ph2[j]; }
      }
      l.coef[j] = cur;
    }
    l.incl = l.mine;
  }
  for (int d = 1; d < (int)kTpLanes; d <<= 1) { // Hillis-Steele, as the wavefront does it
    std::vector<Lp24Affine> prev(kTpLanes);
    for (uint32_t lane = 0; lane < kTpLanes; ++lane) prev[lane] = L[lane].incl;
    for (uint32_t lane = d; lane < kTpLanes; ++lane) lp24_affine_compose(L[lane].incl, prev[lane - d]);
  }
  const double s_init[4] = {s0.filt.s0, s0.filt.s1, s0.filt.s2, s0.filt.s3};
  for (uint32_t lane = 0; lane < kTpLanes; ++lane) lp24_affine_mul(L[lane].incl, s_init, L[lane].s_end, true);
  for (uint32_t lane = 0; lane < kTpLanes; ++lane) {
    Lane& l = L[lane];
    double st[4];
    for (int i = 0; i < 4; ++i) st[i] = lane ? L[lane - 1].s_end[i] : s_init[i];
    for (uint32_t j = 0; j < l.cnt; ++j) {
      float y = 0.0f;
      if (l.lives[j]) y = (float)lp24_step_v(st, l.coef[j], (double)l.x[j]);
      const float m = y * l.amp[j];
      outL[l.n0 + j] = m * p.gl; outR[l.n0 + j] = m * p.gr;
    }
  }
  if (frames) {
    Lane& l = L[(frames - 1) / kTpChunk];
    l.s.o1.x1 = nz_end[0].x1; l.s.o1.x2 = nz_end[0].x2; l.s.o2.x1 = nz_end[1].x1; l.s.o2.x2 = nz_end[1].x2;
    l.s.lfo.x1 = nz_end[2].x1; l.s.lfo.x2 = nz_end[2].x2;
    l.s.filt.s0 = l.s_end[0]; l.s.filt.s1 = l.s_end[1]; l.s.filt.s2 = l.s_end[2]; l.s.filt.s3 = l.s_end[3];
    state = l.s;
  }
}

extern "C" {
void emul_set_f32_filter(int form) { groove::groove_emul_f32_filter = form; }
// time_parallel != 0: Welsh voices render through the time-parallel form (blocks of up to 256 frames)
void emul_set_time_parallel(void* h, int on);
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
// f32_kind != 0: what the per-kind kernels of big banks do — patches the host measures safe (derive.h welsh_filter_f32_ok ->
// WF_FILTER_F32) run their filter's recurrence in fp32 (segmented form only).  Returns how many voices carry the flag.
uint32_t emul_set_f32_kind(void* h, int on) {
  EmulBank* b = (EmulBank*)h;
  b->f32_kind = on;
  uint32_t flagged = 0;
  for (uint32_t v = 0; v < b->n && b->kind == 0; ++v) {
    b->wp[v].flags &= ~WF_FILTER_F32;
    if (on && welsh_filter_f32_ok(b->wp[v], b->sr)) { b->wp[v].flags |= WF_FILTER_F32; ++flagged; }
  }
  return flagged;
}
// What kernel instantiation the library gives this patch (dsp_core.h welsh_base_kind / welsh_body_classes; derive.h flags):
// out = {base kind 0..5, LFO class, oscillator 1 class, oscillator 2 class, WF_FILTER_F32 promise, derived flags word}.
void emul_welsh_classify(const groove_welsh_params* p, uint32_t sr, uint32_t out[6]) {
  WelshCold c;
  WelshParams o = derive_welsh(*p, (double)sr, c);
  const int base = welsh_base_kind(o);
  int cl, c1, c2;
  welsh_body_classes(o, base, cl, c1, c2);
  out[0] = (uint32_t)base; out[1] = (uint32_t)cl; out[2] = (uint32_t)c1; out[3] = (uint32_t)c2;
  out[4] = welsh_filter_f32_ok(o, (double)sr) ? 1u : 0u; out[5] = o.flags;
}
double emul_filter_f32_error(const groove_welsh_params* p, uint32_t sr) { WelshCold c; return welsh_filter_f32_error(derive_welsh(*p, (double)sr, c), (double)sr); }
void emul_set_segmented(void* h, int on) { ((EmulBank*)h)->segmented = on; }
void emul_set_lfo_look_ahead(void* h, int on) { ((EmulBank*)h)->lfo_look_ahead = on; }
void emul_set_time_parallel(void* h, int on) { ((EmulBank*)h)->time_parallel = on; }
// role_split != 0: Welsh voices of the four class-specialised base kinds (no exact-f64 LFO) render role by role, as the
// role-split kernel does (welsh_split.h); must give the segmented form's bits
void emul_set_role_split(void* h, int on) { ((EmulBank*)h)->role_split = on; g_roles = on == 4 ? 4 : 3; } // (4: the four-role form)
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
  const RenderConsts rc = render_consts(b->sr);
  for (uint32_t v = 0; v < n; ++v) {
    if (b->kind == 0 && b->time_parallel && frames <= kTpMaxFrames) {
      std::vector<float> l(frames), r(frames);
      welsh_tp_render_voice(b->wp[v], b->ws[v], rc, frames, l.data(), r.data());
      for (uint32_t f = 0; f < frames; ++f) { out[(size_t)f * n + v] = l[f]; out[((size_t)frames + f) * n + v] = r[f]; }
      continue;
    }
    WelshScratch sc{};
    bool retunes = false;
    int mode = LFO_F64;
    if (b->kind == 0) {
      sc = welsh_scratch_init(b->wp[v], rc); retunes = welsh_retunes(b->wp[v]);
      mode = b->generic_lfo ? (welsh_lfo_mode(b->wp[v]) == LFO_F32 ? LFO_F32 : LFO_F64) : welsh_lfo_mode(b->wp[v]);
    }
    float sampler_buf[16] = {};
    uint32_t seg_left = 0, seg_len = 0; // segmented form (uniform kernels): frames left before the next boundary check
    bool seg_live = false;
    const bool f32 = b->kind == 0 && b->f32_kind && b->segmented && mode != LFO_F64 && (b->wp[v].flags & WF_FILTER_F32);
    if (f32) welsh_scratch_f32_begin(b->wp[v], b->ws[v], rc, sc);
    const bool split = b->kind == 0 && b->role_split && b->segmented && mode != LFO_F64; // (the exact-f64 kinds keep the all-kinds kernel)
    Lp24CoefD split_coef = sc.coef; // role C's coefficients: welsh_scratch_init's at the start of the block
    for (uint32_t f = 0; f < frames; ++f) {
      float L, R;
      if (split) { // mirrors welsh_split_front_impl's walk (frame 0 checked, then hoisted segments) with roles B and C in line
        if (f == 0) welsh_emul_split_frame<true, false>(b->wp[v], b->ws[v], rc, sc, split_coef, retunes, mode, L, R);
        else {
          if (seg_left == 0) {
            seg_left = welsh_segment_begin(b->wp[v], b->ws[v], seg_live);
            if (seg_left < b->min_seg) b->min_seg = seg_left;
            if (seg_left == 0) seg_left = 1; // the kernels' guard (kernels.h run_frames_segmented): never taken by a consistent record
            welsh_segment_start_hoisted(b->ws[v], sc);
            if (seg_left > frames - f) seg_left = frames - f;
            seg_len = seg_left;
          }
          L = R = 0.0f;
          if (seg_live) welsh_emul_split_frame<false, true>(b->wp[v], b->ws[v], rc, sc, split_coef, retunes, mode, L, R);
          if (--seg_left == 0) welsh_segment_end_hoisted<false>(b->wp[v], b->ws[v], seg_len, seg_live);
        }
      } else
      if (b->kind == 0 && b->segmented && f > 0) { // mirrors run_frames_segmented<HOISTED> with a one-lane wave
        if (seg_left == 0) {
          seg_left = welsh_segment_begin(b->wp[v], b->ws[v], seg_live);
          if (seg_left < b->min_seg) b->min_seg = seg_left;
          if (seg_left == 0) seg_left = 1; // the kernels' guard
          if (seg_left > frames - f) seg_left = frames - f;
          seg_len = seg_left;
          welsh_segment_start_hoisted(b->ws[v], sc);
        }
        L = R = 0.0f;
        const bool ltab = b->lfo_look_ahead && mode == LFO_F64_SMOOTH && seg_len >= 8; // (kernels.h CoefTab::kMinSegment)
        if (seg_live && ltab) {
          const uint64_t ph = b->ws[v].lfo.phase + (uint64_t)(seg_len - seg_left + 1) * b->wp[v].lfo_inc; // the phase stands for the segment
          if (f32) { if (retunes) welsh_emul_ltab_frame<true, true>(b->wp[v], b->ws[v], rc, sc, ph, L, R); else welsh_emul_ltab_frame<false, true>(b->wp[v], b->ws[v], rc, sc, ph, L, R); }
          else { if (retunes) welsh_emul_ltab_frame<true, false>(b->wp[v], b->ws[v], rc, sc, ph, L, R); else welsh_emul_ltab_frame<false, false>(b->wp[v], b->ws[v], rc, sc, ph, L, R); }
          if (--seg_left == 0) {
            welsh_segment_end_hoisted<false>(b->wp[v], b->ws[v], seg_len, seg_live);
            b->ws[v].lfo.phase += (uint64_t)seg_len * b->wp[v].lfo_inc;
            welsh_lfo_reseed_smooth<OSC_ANY>(b->wp[v], b->ws[v], sc);
          }
          out[(size_t)f * n + v] = L;
          out[((size_t)frames + f) * n + v] = R;
          continue;
        }
        if (seg_live && f32) { if (retunes) welsh_emul_f32_frame<false, true>(b->wp[v], b->ws[v], rc, sc, mode, L, R); else welsh_emul_f32_frame<false, false>(b->wp[v], b->ws[v], rc, sc, mode, L, R); }
        else if (seg_live) welsh_emul_segment_frame(b->wp[v], b->ws[v], rc, sc, retunes, mode, L, R);
        if (--seg_left == 0) welsh_segment_end_hoisted<false>(b->wp[v], b->ws[v], seg_len, seg_live);
      } else if (b->kind == 0 && f32) { // (frame 0 of the fp32-filter copy)
        if (retunes) welsh_emul_f32_frame<true, true>(b->wp[v], b->ws[v], rc, sc, mode, L, R); else welsh_emul_f32_frame<true, false>(b->wp[v], b->ws[v], rc, sc, mode, L, R);
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
    if (f32) welsh_scratch_f32_end(b->ws[v], sc);
  }
}
uint32_t emul_bank_min_segment(void* h) { return ((EmulBank*)h)->min_seg; }
// welsh_segment_begin on a voice whose two envelope records are given word by word (state, n, N): what a lane that loaded
// such a record — consistent, or TORN between two stores — contributes to its wave's minimum.
uint32_t emul_segment_begin_of(const groove_welsh_params* p, uint32_t sr, const uint32_t amp[3], const uint32_t fil[3]) {
  WelshCold cold;
  const WelshParams wp = derive_welsh(*p, (double)sr, cold);
  WelshState s = initial_welsh_state();
  s.amp.state = amp[0]; s.amp.n = amp[1]; s.amp.N = amp[2];
  s.fil.state = fil[0]; s.fil.n = fil[1]; s.fil.N = fil[2];
  bool live;
  return welsh_segment_begin(wp, s, live);
}
float emul_bitcrush(float x, uint32_t bits) { return bitcrush(x, bits); }
// the device's 24 dB coefficients (dsp_core.h lp24_coefd_from_fc) and the same in f64 throughout (derive.h lp24_coeffs_h): out[0..5], out[6..11]
void emul_lp24_coef_both(double ripple, float fc, float sr, double* out) {
  const Lp24Consts c = derive_lp24_consts(ripple);
  const Lp24CoefD a = lp24_coefd_from_fc(c, fc, (float)(3.14159265358979323846 / sr), (float)(0.49 * sr));
  std::memcpy(out, &a, sizeof a);
  lp24_coeffs_h((double)fc, ripple, (double)sr, out + 6);
}
}
