// groove_hip.hip — implementation of the C ABI in include/groove_hip.h.
//
// Host side of the MI355X render path: owns device memory, derives packed per-lane
// parameters (derive.h), queues block-granular note events, launches the kernels in
// kernels.h (voice kernels on per-kind / per-bank side streams, everything else on the ctx stream).
// Nothing here falls back to a CPU path: every entry point
// either launches HIP work or returns an error.
#include "../../include/groove_hip.h"
#include "kernels.h"
#include "welsh_tp.h"
#include "welsh_split.h"
#include "fx_tp.h"
#include <dlfcn.h>
#include <hip/hip_ext.h> // hipExtLaunchKernelGGL: an event bound to a kernel's own completion signal (launch_reduce, launch_welsh_tp)
#include <rccl/rccl.h> // types, enumerators and prototypes only: the library itself is dlopen'ed (rccl_open)
#include <string>
#include <map>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <memory>
#include <thread>
#include <new>

using namespace groove;

namespace {
thread_local std::string g_last_error;
}

// The lane order of a bank that keeps its voices in another order than the caller gave them (patch-major regrouping,
// welsh_upload_params): src_lane[caller's lane] = the library's lane.  Shared by the bank and by every block its renders
// filled, so that a block outlives a re-grouping (or the bank) with the order its content has.
struct LaneOrder {
  uint32_t n = 0;
  uint32_t* d_src_lane = nullptr;
  ~LaneOrder() { if (d_src_lane) (void)hipFree(d_src_lane); }
};

struct groove_block {
  groove_ctx* ctx;
  uint32_t n, cap;
  float* d;
  // LAZY LANE ORDER.  A render of a regrouped bank fills `d_alt` in the LIBRARY's lane order (coalesced rows) and tags the
  // block with that order; `d` — the pointer the caller can ask for — is only produced (one gather pass, block_normalise)
  // when something needs the caller's order: an effect (per-lane parameters), an element-wise accumulate, a download,
  // groove_block_device_ptr.  The mix bus never does (a sum over the lanes has no order), so render + mix of a regrouped
  // bank costs what it costs for a grouped one (round 2 gathered every block: 2.85 ms against 0.97 at 1,000,000 voices).
  float* d_alt = nullptr;
  std::shared_ptr<LaneOrder> order; // set: the block's current content is d_alt, in this order
  uint32_t order_frames = 0;
  // groove_bank_render_async: the producer kernels run on side streams; ev_ready[k] is recorded on
  // side stream k behind them, ready_mask says which are outstanding (block_acquire clears it).
  hipEvent_t ev_free = nullptr, ev_ready[16] = {};
  uint32_t ready_mask = 0;
  bool released = false; // groove_block_release: ev_free marks the end of the block's consumers so far
  bool free_marked = false; // ev_free already completes with the block's last consumer (groove_mix bound it to its last kernel): a release needs no record
  // The lane sums of the block, as the render that filled it left them: sums[row][ch][frame], `sum_rows` rows whose
  // column totals are the block's bus contribution (the fused path's partial rows).  groove_mix reduces these 8 MB
  // instead of reading 2 GB of voice block back; anything else that writes the block clears sums_valid.
  float* d_sums = nullptr;
  size_t sums_cap = 0;
  uint32_t sum_rows = 0, sum_frames = 0;
  bool sums_valid = false;
  // The all-pass stream (groove_set_fx_allpass_stream; fx_launch_run): a chain that ends in a reverb leaves the block's last
  // kernel — the reverb's two all-passes — on a side stream, beside the next block's fused run.  That kernel reads the comb sum
  // from the BLOCK's own staging buffer (no other block's kernels touch it) and writes the lane sums into one of two buffers the
  // block keeps for that stream alone (only all-pass kernels, all on that one stream, and flushes that order themselves against
  // it, ever touch them): nothing here needs a cross-stream wait in the steady state.
  float* d_stage = nullptr;
  uint32_t stage_cap = 0; // frames
  float* d_sums_ap[2] = {nullptr, nullptr};
  size_t sums_ap_cap[2] = {0, 0};
  int ap_flip = 0;
  bool sums_on_ap = false; // the valid lane sums are d_sums_ap[ap_flip], written on the all-pass stream
};

enum BankKind { BANK_WELSH = 0, BANK_FM = 1, BANK_SAMPLER = 2 };


// One distinct filter description of a Welsh patch: what derive.h welsh_filter_f32_error depends on (welsh_upload_params).
struct F32FilterKey {
  float c0, d1, c2, d3, hz, start, end, depth;
  uint32_t bits;
  bool operator<(const F32FilterKey& o) const { return std::memcmp(this, &o, sizeof(F32FilterKey)) < 0; }
};
struct groove_bank {
  groove_ctx* ctx;
  BankKind kind;
  std::map<F32FilterKey, bool> f32_memo; // WF_FILTER_F32 verdicts measured so far at f32_memo_sr (0.6 ms each: not again for every control change)
  double f32_memo_sr = 0.0;
  uint32_t n;
  uint32_t pw, sw; // param / state words per lane
  uint32_t* d_params = nullptr;
  uint32_t* d_state = nullptr;
  double* d_cold = nullptr; // welsh: [4][n]; fm: ratio [n]
  WaveDesc* d_waves = nullptr;   // welsh: virtual waves (runs of <= 64 voices sharing a patch)
  uint32_t n_vwaves = 0;         // 0: the bank runs on the per-lane kernel
  size_t vwaves_cap = 0;
  // welsh, fused path: block pipeline (render_mix_pipelined).  Two slots of partial rows / segment sums and
  // the events that order slot reuse: a base kind's render of block b+2 waits for the reduce of block b.
  float* d_pipe_part[2] = {nullptr, nullptr};
  float* d_pipe_seg[2] = {nullptr, nullptr};
  size_t pipe_part_cap[2] = {0, 0}, pipe_seg_cap[2] = {0, 0};
  hipEvent_t ev_render_done[kBaseKinds + 4][2] = {};
  hipEvent_t ev_reduce_done[2] = {nullptr, nullptr};
  bool reduce_recorded[2] = {false, false};
  int pipe_slot = 0;
  // groove_bank_render_mix_paced: the reduction of the block rendered by the last paced call, launched by the next one (or a flush)
  struct { bool active = false; int slot = 0; uint32_t rows = 0, frames = 0, used = 0; float* bus = nullptr; int accumulate = 0; } paced;
  int stream_slot = 0; // side stream of a single-kernel bank (FM, sampler, per-lane Welsh) in the asynchronous fused path
  bool ctx_touched = true; // the ctx stream has worked on this bank's state since its last asynchronous render waited for it
  int side_mode = 0;    // which side streams carried this bank's last asynchronous work: 0 none, 1 one per base kind, 2 stream_slot
  // welsh: lane permutation.  A bank whose patches are interleaved voice by voice is kept patch-major inside
  // the library (params, state, cold values in INTERNAL lane order) so that it runs on the wave-uniform kernels;
  // perm[internal lane] = caller's voice index, inv = its inverse.  Empty = identity.
  std::vector<uint32_t> perm, inv;
  std::shared_ptr<LaneOrder> order; // device copy of inv, shared with the blocks this bank's renders fill (groove_block: lazy lane order)
  uint8_t* d_wg_cls = nullptr;   // welsh: oscillator class pair of each entry of d_wg_list
  bool tp_full_coef = false;     // welsh: some voice routes the LFO to the resonance (welsh_tp_kernel<.., FULL_COEF>)
  bool tp_pairs = false;         // welsh: every pair of adjacent voices (2i, 2i + 1) shares a patch (welsh_tp_kernel<.., VPW = 2>)
  uint8_t* d_wg_base = nullptr;  // welsh: base kind of each entry of d_wg_list (the all-kinds kernel of small banks)
  uint8_t* d_wg_f32 = nullptr;   // welsh: 1 where the entry's patches carry WF_FILTER_F32 (the fused per-kind kernels)
  uint32_t mix_off[3] = {}, mix_cnt[3] = {}; // welsh: the MIX kernel's three sections of the lists (kernels.h): entries [wg_list_cap + mix_off[s], + mix_cnt[s]) of all four arrays — the
                                             // class-specialised workgroups of the kind-sorted list taken with a stride of three, each section kind-sorted itself
  uint32_t* d_wg_list = nullptr; // welsh: workgroup ids (groups of 4 virtual waves) sorted by kind (kernels.h)
  size_t wg_list_cap = 0;
  uint32_t wgs_of_kind[kWgKinds] = {};  // slice lengths of d_wg_list, in kind order
  float* d_pcm = nullptr;   // sampler bank
  groove_note_event* d_ev[2] = {nullptr, nullptr};
  // pinned staging of queued note events, two slots in rotation: flush_events hands the events to the ctx
  // stream and returns without a host synchronisation (a project with note events in every block would
  // otherwise serialise host and GPU once per block)
  groove_note_event* h_ev[2] = {nullptr, nullptr};
  size_t h_ev_cap[2] = {0, 0};
  hipEvent_t ev_staged[2] = {nullptr, nullptr};
  bool staged[2] = {false, false};
  int ev_slot = 0;
  groove_block* scratch = nullptr; // for render_mix
  std::vector<groove_note_event> pending;
  InlineEvents inline_ev{};  // sampler: this block's events, to ride in the next time-parallel render's arguments (flush_events)
  std::vector<groove_welsh_params> welsh;
  std::vector<groove_fm_params> fm;
  std::vector<groove_sampler_params> sampler;
  std::vector<groove_sample_desc> descs;
};

struct groove_fx {
  groove_ctx* ctx;
  uint32_t kind, n;
  std::vector<groove_fx_params> p;
  float* d_fa = nullptr;   // per-lane float param A (ceiling / limit_min / attenuation)
  float* d_fb = nullptr;   // per-lane float param B (limit_max / ratio)
  uint32_t* d_ua = nullptr; // per-lane uint param (bits)
  float* d_wet = nullptr;
  double* d_coef = nullptr; // [5|6][n]
  double* d_st = nullptr;   // [4][2n]
  float* d_ring = nullptr;  // rows of 2n floats
  size_t ring_rows = 0;
  // uniform geometry
  uint32_t N = 0, w = 0, voices = 1, spacing = 0;
  bool all_wet = true; // every lane has wet == 1 (enables the split reverb path)
  ReverbGeom geo{};
  // reverb, direct all-pass form (kernels.h): second copies of the two all-pass rings (read one, write the other, swap)
  // and a staging block for the comb sum, allocated at first use
  uint64_t ap_alt[2] = {0, 0};
  float* d_tmp = nullptr;
  uint32_t tmp_cap = 0; // frames
  bool ap_busy = false;         // the all-pass lines' last kernel went to the all-pass stream (ap_join before the ctx stream touches them)
  int last_side = -1;           // side stream whose kernels touched this effect last (-1: the ctx stream); fx_acquire_ctx
  hipEvent_t ev_done = nullptr; // end of that use
  bool done_recorded = true;    // false: the last use was a kernel whose end another event marks (a fused render's): ev_done is recorded on demand
};

// Events that only order the library's own streams on one device: no timing, and no system-scope fence
// (cache writeback + invalidate) when they are recorded — the host reads results through
// hipMemcpy / hipStreamSynchronize, which fence by themselves.
#ifndef GROOVE_SOURCE_HASH
#define GROOVE_SOURCE_HASH "unknown" /* groove_amd/Makefile: sha256 of the library's sources, first 16 hex digits */
#endif
constexpr unsigned kSyncEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
constexpr int kBankStreams = 4;                         // shared round-robin by single-kernel banks (FM, sampler, per-lane Welsh)
constexpr int kSideStreams = kBaseKinds + kBankStreams;
static_assert(kBaseKinds == 6 && kBankStreams == 4, "groove_init lists the side streams in creation order"); // + one per Welsh base kind; the ctx stream carries events, reductions and the rest
static_assert(kSideStreams <= 16, "groove_block::ev_ready holds one event per side stream");
struct groove_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipStream_t side_stream[kSideStreams] = {}; // kernels of the other workgroup kinds run beside the main one
  hipStream_t placeholder_stream = nullptr;   // created, never used (groove_init)
  unsigned long long* hb = nullptr;           // -DGROOVE_HEARTBEAT (diagnostic build): [workgroups started, finished] of the per-kind kernels, in coherent host memory
  uint32_t* d_diag = nullptr;                 // DiagCounters (diag.h): the segment guard's counter, read by groove_debug_info
  hipEvent_t ev_fork = nullptr, ev_join[kSideStreams] = {};
  bool side_busy[kSideStreams] = {};    // work enqueued on the side stream since the last join
  bool fork_pending[kSideStreams] = {}; // the side stream has not yet waited for ev_fork
  bool need_fork = true;                // ctx-stream work since the last fork that side streams must wait for
  uint32_t tp_max_voices = kTpMaxVoices; // Welsh banks up to this size render time-parallel (welsh_tp.h); GROOVE_TP_MAX_VOICES overrides, 0 = never
  uint32_t fx_tp_max_lanes = 4096;       // IIR effect banks of up to this many lane-channels (half as many for the 24 dB low-pass) run time-parallel (fx_tp.h: measured crossovers, tools/fx_bench.py)
  // Mid-size Welsh banks (too big for the time-parallel form, too small to fill the chip with one voice-wave per wavefront):
  // the ROLE-SPLIT kernel (welsh_split.h: three wavefronts per 64 voices, pipelined over the block's frames) for banks of up
  // to this many virtual waves; 0 = never.  groove_set_split_max_waves.
  uint32_t split_max_waves = 1024;      // 65,536 voices = one workgroup (twelve wavefronts) per CU; measured (round 3, blocks 5-24): 20,000 voices 0.120 -> 0.090 ms per block, 32,768 0.119 -> 0.090, 65,536 0.123 -> 0.095; 80,000 (a second round of workgroups) 0.135 -> 0.153: not above
  uint32_t split2_max_waves = 1024;      // (= split_max_waves: off by default since the end of round 6 — with the FAST copies of its bodies the all-kinds kernel walks a block of 80,000 - 125,000 voices in 0.091 - 0.095 ms where this form takes 0.119 - 0.125, in one job, tools/ab_env.sh; GROOVE_SPLIT2_MAX_WAVES=2048 brings it back.)  Banks above split_max_waves and up to this many virtual waves (131,072 voices until then): the TWO-role form, front + tangent | back — two workgroups of eight wavefronts per CU, so one round still; measured (blocks 5-24): 100,000 voices 0.1385 -> 0.134 ms per block, 125,000 0.145 -> 0.137 (three roles there: 0.164 / 0.165; two roles at 65,536: 0.106 against three roles' 0.095)
  int split_roles = 4;                   // roles of the form used up to split_max_waves: four (ctl | osc | tangent + quotients | back), measured against three
                                         // (front | tangent | back) in one job: 32,768 voices 0.0830 against 0.0888 ms per block, 65,536 0.0846-0.0855 against
                                         // 0.0903, config #5 0.1003-0.1008 against 0.0999-0.1001.  GROOVE_SPLIT_ROLES=3 / 2: A/B
  uint32_t pipeline_min_waves = 3800;   // banks at least this long (~243,000 voices; 7,000 = ~450,000 until the end of round 6: with the FAST copies of the bodies in both kernels the crossover moved down — in one job, tools/ab_env.sh, mix kernel against all-kinds kernel: 200,000 voices 0.114 against 0.110 ms per block, 250,000 0.120 against 0.121 - 0.123, 300,000 0.125 against 0.158, 400,000 0.139 against 0.188, 500,000 0.164 against 0.212) run the mix kernel (one launch per base kind before round 6) and pipeline their fused blocks; smaller ones take the all-kinds kernel (round 2, blocks 5-44 of the timeline: 300,000 voices 0.275 -> 0.250 ms per block, 500,000 0.357 -> 0.342; 600,000 0.372 against 0.400.  Round 5: the per-kind kernels alone carry the fp32 filter kind and the crossover moved down from ~550,000 — the driver's window, in-job, per-kind against all-kinds: 500,000 voices 0.271 against 0.281, 420,000 0.246 against 0.244, 350,000 0.223 against 0.225: profiles/r05_pipeline_threshold_ab.log)
  // How many of the bank streams exist and are handed out.  Three: with the ctx stream and the four
  // kind streams that is eight streams; a ninth lands on a hardware queue that already carries one of the others, and a
  // project with a bank on it ran three times slower (mixed-131072: 0.12 -> 0.39 ms per block whenever a bank had the
  // fourth bank stream, whichever bank it was; tools/micro/slot_probe.py).
  int bank_streams = 3;
  // Normal-priority streams for the class-specialised Welsh kinds.  Three: the fourth kind shares the first kind's stream
  // (1,000,000 voices: 0.511 against 0.515 ms per block with four).  docs/STREAMS.md item 10: the k-th stream a process
  // creates lands on hardware queue (k - 1) mod 4, so the fifth (and the ninth) shares the ctx stream's queue; the fifth
  // used to be the fourth kind stream and is now a placeholder nobody uses.
  int kind_streams = 3;
  uint32_t look_ahead = 3u; // RenderConsts::look: bit 0 the coefficient look-ahead, bit 1 the LFO look-ahead (kernels.h); GROOVE_LOOK_AHEAD=<bits>, groove_set_look_ahead (tests render the same bank with and without)
  bool mix_kernel = true; // big Welsh banks: the four class-specialised kinds in three balanced launches (kernels.h, the MIX kernel); GROOVE_MIX_KERNEL=0: one launch per base kind (round 5's form, for A/B runs)
  int next_stream_slot = 0;             // round-robin side-stream assignment of single-kernel banks
  uint32_t fm_tp_max_voices = kFmTpMaxVoices;
  // Welsh banks of at least this many voices whose adjacent pairs share a patch render two voices per wavefront
  // (welsh_tp_kernel<.., VPW = 2>): above 3,072 voices the one-voice form no longer fits the SIMDs in one round
  uint32_t tp_vpw2_min_voices = 3073; // groove_set_time_parallel_pair_min_voices (0 = never)
  // events that mark the end of ONE kernel (a block's render, a block's last reduction) are bound to that dispatch's own
  // completion signal instead of being recorded behind it (a barrier packet each, ~5 us of the stream's timeline)
  uint32_t fm_tp_vpw4_min_voices = 4096; // GROOVE_FM_TP_VPW4_MIN_VOICES (0 = never): FM banks of at least this many voices, four voices per wavefront
  // (Round 3 switched this off when fresh runs stalled with it on; the stall was the zero-frame segment of DESIGN.md section 7,
  // which had nothing to do with events.  On again in round 4; GROOVE_BIND_EVENTS=0 for A/B.)
  bool bind_events = true;
  uint32_t fx_seg_max_lanes = 49152;     // biquad banks of up to this many lane-channels take the four-segment kernel (measured, tools/fx_bench.py: 8,192 lane-channels 17.5 -> 9.3 us, 32,768 19.4 -> 15.6, 131,072 42.9 -> 58.4; 0 = never)
  uint32_t fx_tp_wide_min_lanes = 8192;  // from this many lane-channels the time-parallel IIR kernels take 32-wide tiles
  // groove_set_fx_allpass_stream: the side stream that carries the all-passes of chains that END in a reverb (-1: the ctx stream
  // carries them, behind the run).  deferred_ap: lane sums written on that stream that groove_mix_deferred has taken and the NEXT
  // all-pass launch (same stream: ordered) puts on their bus.
  int fx_ap_stream = -1;
  struct { const float* rows = nullptr; float* bus = nullptr; uint32_t n_rows = 0, frames = 0; int accumulate = 0; } deferred_ap;
  hipEvent_t ev_ap_run = nullptr, ev_ap_x = nullptr;
  bool seq_allpass = false;             // GROOVE_FX_SEQ_ALLPASS=1: the sequential all-pass kernel (A/B and bit-identity tests)
  bool chunked_allpass = false;         // GROOVE_FX_CHUNKED_ALLPASS=1: the chunk-parallel all-pass kernel instead of the direct one
  uint32_t sr = GROOVE_DEFAULT_SAMPLE_RATE;
  // Blocking waits (groove_synchronize and every call that hands data to the host) poll the stream with a deadline
  // instead of sleeping inside hipStreamSynchronize: a kernel that does not complete then comes
  // back as an ERROR that names the streams still busy, not as a hang.  0 = wait for ever.  GROOVE_SYNC_TIMEOUT_MS.
  uint32_t sync_timeout_ms = 60000;
  unsigned long long host_waits = 0, host_waits_blocked = 0, host_wait_ns = 0; // wait_deadline: calls, calls that found work pending, time blocked (groove_debug_info)
  bool safe_streams = false;            // GROOVE_SAFE_STREAMS=1: one priority, four streams in all (ctx + three that kinds and banks share)
  bool comm_before_streams = false;     // groove_init_comm: the RCCL communicator (and its streams) existed before the library's own
  int streams_created = 0;              // hipStreamCreate* calls of this ctx, in order: ctx, kind streams, placeholder, bank streams
  std::string err;
  std::vector<groove_bank*> banks;
  std::vector<groove_fx*> fxs;
  float* d_partial = nullptr;
  size_t partial_cap = 0;
  float* d_fpart = nullptr;  // fused path: partial[workgroup][2][frames]
  size_t fpart_cap = 0;
  // groove_bank_render_mix_deferred: two more row buffers in alternation, and the block whose rows are waiting for the next
  // deferred render (or bus_flush) to put them on the bus
  float* d_dpart[2] = {nullptr, nullptr};
  size_t dpart_cap[2] = {0, 0};
  int dpart_next = 0;
  struct { const float* rows = nullptr; float* bus = nullptr; uint32_t n_rows = 0, frames = 0; int accumulate = 0; size_t owned_cap = 0; } deferred;
  // groove_mix_deferred takes the block's row-sum buffer AWAY from the block (owned_cap != 0: the pending rows live in a buffer
  // nobody else can write) and hands the block one of these instead; a consumed buffer comes back here (deferred_taken)
  std::vector<std::pair<float*, size_t>> spare_sums;
  bool f32_filter = true; // WF_FILTER_F32: patches whose 24 dB filter is measured safe in fp32 take the fp32 recurrence in the per-kind kernels (GROOVE_F32_FILTER=0: never)
  std::vector<groove_bank*> paced_order; // banks with a pending paced reduction, in call order (= the order of their sums on a bus)
  float* d_fseg = nullptr;   // fused path: seg[segments][2*frames]
  size_t fseg_cap = 0;
  int16_t* d_i16 = nullptr;
  size_t i16_cap = 0;
  // RCCL (dlopen'ed lazily)
  void* rccl = nullptr;
  void* comm = nullptr;
  int rank = 0, world = 1;
};

namespace {

// Returned through hipError_t-typed helpers whose failure is already described in ctx->err: GHIP passes it on untouched.
constexpr hipError_t kGrooveFailed = hipErrorAssert;
int fail(groove_ctx* ctx, const std::string& msg) {
  g_last_error = msg;
  if (ctx) ctx->err = msg;
  return 1;
}
#define GHIP(ctx, expr)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ == hipErrorNotReady) return 2; /* a deadline passed: wait_deadline wrote the message */ \
    if (e_ == kGrooveFailed) return 1;    /* the callee has written the message (ctx_wait: bus_flush) */ \
    if (e_ != hipSuccess)                                                                  \
      return fail(ctx, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)

// Synchronous copies go through the CTX stream, never the null stream: the null stream is one more normal-priority
// stream for the runtime to map, and with the four kind streams it made five on four hardware queues — the fourth kind
// stream shared its queue with it (rocprofv3 trace, round 2).
RenderConsts render_consts_of(const groove_ctx* ctx) { RenderConsts rc = render_consts(ctx->sr); rc.look = ctx->look_ahead; return rc; }
const char* side_stream_name(int k) {
  static const char* names[] = {"kind stream 0 (Welsh F32 static; also kinds 3 and 4)", "kind stream 1 (Welsh F32 retune; also kind 5)",
                                "kind stream 2 (Welsh smooth-LFO static)", "kind stream 3 (Welsh smooth-LFO retune)",
                                "kind stream 4", "kind stream 5", "bank stream 0", "bank stream 1", "bank stream 2", "bank stream 3"};
  return k >= 0 && k < 10 ? names[k] : "side stream";
}
// Wait for `st` (or, if `ev` is given, for that event) with the ctx's deadline.  hipSuccess when it completed;
// hipErrorNotReady when the deadline passed — ctx->err then names every library stream that still has work.  Nothing is
// cancelled: the work stays queued and a later wait may still see it complete.
hipError_t wait_deadline(groove_ctx* ctx, hipStream_t st, hipEvent_t ev, const char* what) {
  if (ctx->sync_timeout_ms == 0) return ev ? hipEventSynchronize(ev) : hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  const auto limit = std::chrono::milliseconds(ctx->sync_timeout_ms);
  for (uint32_t spins = 0;; ++spins) {
    const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(st);
    if (q != hipErrorNotReady) {
      ctx->host_waits += 1;
      if (spins) { ctx->host_waits_blocked += 1; ctx->host_wait_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
      return q;
    }
    const auto waited = std::chrono::steady_clock::now() - t0;
    if (waited > limit) break;
    // the first ~100 us busy-poll (most waits of this path are that short), then yield, then sleep in growing steps.  (A sleep of
    // 50 us returns after 100 - 150: with sleeps from 2 ms on, a 2.5 ms wait — the end of a 20-block window of a 125,000-voice shard —
    // came back up to 0.1 ms late, 5 us per block of a 0.12 ms block; waits of up to 20 ms now yield.)
    if (spins < 64) continue;
    if (waited < std::chrono::milliseconds(20)) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(waited < std::chrono::milliseconds(200) ? 50 : 500));
  }
  std::string busy;
  if (hipStreamQuery(ctx->stream) == hipErrorNotReady) busy += "ctx stream";
  for (int k = 0; k < kSideStreams; ++k) {
    hipStream_t s = ctx->side_stream[k];
    if (!s) continue;
    bool dup = false;
    for (int j = 0; j < k; ++j) dup = dup || ctx->side_stream[j] == s;
    if (dup) continue;
    if (hipStreamQuery(s) == hipErrorNotReady) { if (!busy.empty()) busy += ", "; busy += side_stream_name(k); }
  }
  // which events of the block pipelines are still pending (render_mix_pipelined): a render that is pending although the
  // reduction it waited for is done is a kernel that does not finish; pending reductions behind done renders would be the ctx
  // stream itself
  std::string pipes;
  for (size_t bi = 0; bi < ctx->banks.size() && bi < 4; ++bi) {
    const groove_bank* bk = ctx->banks[bi];
    for (int slot = 0; slot < 2; ++slot) {
      if (!bk->ev_reduce_done[slot]) continue;
      pipes += "; bank " + std::to_string(bi) + " slot " + std::to_string(slot) + ": reduction " + (hipEventQuery(bk->ev_reduce_done[slot]) == hipSuccess ? "done" : "PENDING") + ", renders";
      for (int k = 0; k < kSideStreams; ++k)
        if (bk->ev_render_done[k][slot]) pipes += std::string(" ") + std::to_string(k) + (hipEventQuery(bk->ev_render_done[k][slot]) == hipSuccess ? ":done" : ":PENDING");
    }
  }
  (void)hipGetLastError(); // hipErrorNotReady is sticky in the runtime's last-error slot
  busy += pipes;
#ifdef GROOVE_HEARTBEAT
  if (ctx->hb) {
    const unsigned long long s0 = ((volatile unsigned long long*)ctx->hb)[0], f0 = ((volatile unsigned long long*)ctx->hb)[1];
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    const unsigned long long s1 = ((volatile unsigned long long*)ctx->hb)[0], f1 = ((volatile unsigned long long*)ctx->hb)[1];
    busy += "; heartbeat of the per-kind kernels: workgroups started " + std::to_string(s0) + " finished " + std::to_string(f0) + ", half a second later started " + std::to_string(s1) + " finished " + std::to_string(f1);
  }
#endif
  fail(ctx, std::string(what) + ": not complete after " + std::to_string(ctx->sync_timeout_ms) + " ms (GROOVE_SYNC_TIMEOUT_MS / groove_set_sync_timeout_ms); still busy: " +
                (busy.empty() ? "nothing (the wait itself raced the completion)" : busy) +
                ".  The work stays queued and a later wait may still see it complete; if it does not, tear the process down (DESIGN.md section 7: the one stall this path has known was an endless loop in a render kernel, fixed in round 4; groove_debug_info's zero_segments counts its trigger).");
  return hipErrorNotReady;
}
int bus_flush(groove_ctx* ctx); // a deferred block's rows onto its bus (groove_bank_render_mix_deferred), defined with the reductions below
hipError_t ctx_wait(groove_ctx* ctx, const char* what = "wait for the ctx stream") {
  if (bus_flush(ctx)) return kGrooveFailed; // (its own message stands.)  Whoever waits for the ctx stream expects every bus to be complete behind it
  return wait_deadline(ctx, ctx->stream, nullptr, what);
}
hipError_t ctx_memcpy(groove_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  // (a copy from or to pageable host memory blocks INSIDE hipMemcpyAsync until the stream gets to it: the deadline has to
  // be applied to the stream first)
  hipError_t e = ctx_wait(ctx, "copy on the ctx stream");
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  if (e != hipSuccess) return e;
  return ctx_wait(ctx, "copy on the ctx stream");
}
hipStream_t side_stream_of(groove_ctx* ctx, int k) {
  if (!ctx->side_stream[k] && hipStreamCreateWithFlags(&ctx->side_stream[k], hipStreamNonBlocking) != hipSuccess) {
    ctx->side_stream[k] = nullptr;
    fail(ctx, "side stream: hipStreamCreate failed");
    return ctx->stream; // degrade to the ctx stream: still ordered, just not concurrent
  }
  return ctx->side_stream[k];
}
// Order the ctx stream after everything enqueued on the side streams, and make later side-stream work
// wait for whatever the ctx stream does next.  Called by every operation that touches bank state or
// parameters outside the pipelined fused render (note events, controls, state download, destroy...).
int ctx_join(groove_ctx* ctx) {
  for (int k = 0; k < kSideStreams; ++k) {
    if (!ctx->side_busy[k]) continue;
    GHIP(ctx, hipEventRecord(ctx->ev_join[k], side_stream_of(ctx, k)));
    GHIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
    ctx->side_busy[k] = false;
  }
  ctx->need_fork = true;
  return 0;
}

// Order the ctx stream after the asynchronous render that produced `blk` (groove_bank_render_async).
// Every ctx-stream operation that reads or writes a block calls this first.
int block_acquire(groove_block* blk) {
  blk->released = false; // the ctx stream is about to use the block again
  blk->free_marked = false;
  if (!blk->ready_mask) return 0;
  groove_ctx* ctx = blk->ctx;
  for (int k = 0; k < kSideStreams; ++k)
    if (blk->ready_mask & (1u << k)) GHIP(ctx, hipStreamWaitEvent(ctx->stream, blk->ev_ready[k], 0));
  blk->ready_mask = 0;
  return 0;
}
// The block's current content, whichever lane order it is in (for readers that do not care: lane sums).
inline const float* block_data(const groove_block* blk) { return blk->order ? blk->d_alt : blk->d; }
// Where a render of bank order `order` (null: the caller's own order) writes; tags the block.
int block_render_target(groove_block* blk, const std::shared_ptr<LaneOrder>& order, uint32_t frames, float** out) {
  if (!order) { blk->order.reset(); *out = blk->d; return 0; }
  if (!blk->d_alt) {
    hipError_t e = hipMalloc(&blk->d_alt, (size_t)2 * blk->cap * blk->n * 4);
    if (e != hipSuccess) return fail(blk->ctx, std::string("block (library lane order): hipMalloc: ") + hipGetErrorString(e));
  }
  blk->order = order;
  blk->order_frames = frames;
  *out = blk->d_alt;
  return 0;
}
// Produce the caller's lane order in blk->d (ctx stream; the caller has acquired the block).
int block_normalise(groove_block* blk) {
  if (!blk->order) return 0;
  groove_ctx* ctx = blk->ctx;
  const uint32_t frames = std::min(blk->order_frames, blk->cap);
  const size_t chs = (size_t)blk->cap * blk->n;
  if (frames) {
    hipLaunchKernelGGL(block_gather_kernel, dim3((blk->n + kThreads - 1) / kThreads, std::min<uint32_t>(frames, 64)), dim3(kThreads), 0, ctx->stream, blk->d, chs,
                       blk->d_alt, chs, blk->order->d_src_lane, blk->n, frames);
    GHIP(ctx, hipGetLastError());
  }
  blk->order.reset();
  return 0;
}
// An effect's memory (IIR state, rings, parameter arrays) is touched by one stream at a time: the ctx stream, or — for the
// stages groove_fx_chain_process_async runs behind a block's asynchronous render — that render's side stream.  ev_done
// marks the end of its last side-stream use; the ctx stream waits for it before it touches the effect again.
hipStream_t side_stream_of(groove_ctx* ctx, int k);
// ev_done of an effect whose last side-stream use was not followed by a record of its own: the CURRENT end of that stream
// (later than needed, never earlier: the stream runs in order)
int fx_done_event(groove_fx* fx) {
  if (fx->done_recorded) return 0;
  GHIP(fx->ctx, hipEventRecord(fx->ev_done, side_stream_of(fx->ctx, fx->last_side)));
  fx->done_recorded = true;
  return 0;
}
int fx_acquire_ctx(groove_fx* fx) {
  if (fx->last_side < 0) return 0;
  if (fx_done_event(fx)) return 1;
  GHIP(fx->ctx, hipStreamWaitEvent(fx->ctx->stream, fx->ev_done, 0));
  fx->last_side = -1;
  return 0;
}
// A bank's state is touched by one set of side streams at a time; changing the set joins first.
int bank_side_mode(groove_bank* b, int mode) {
  if (b->side_mode && b->side_mode != mode && ctx_join(b->ctx)) return 1;
  b->side_mode = mode;
  return 0;
}

template <class T>
std::vector<uint32_t> to_soa(const std::vector<T>& aos) {
  const size_t n = aos.size(), W = sizeof(T) / 4;
  std::vector<uint32_t> soa(W * n);
  for (size_t v = 0; v < n; ++v) {
    uint32_t w[sizeof(T) / 4];
    std::memcpy(w, &aos[v], sizeof(T));
    for (size_t i = 0; i < W; ++i) soa[i * n + v] = w[i];
  }
  return soa;
}
template <class T>
int upload_soa(groove_ctx* ctx, uint32_t* dst, const std::vector<T>& aos) {
  std::vector<uint32_t> soa = to_soa(aos);
  GHIP(ctx, hipMemcpyAsync(dst, soa.data(), soa.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}
inline uint32_t blocks_for(size_t items) { return (uint32_t)((items + kThreads - 1) / kThreads); }

// Welsh: derive + upload parameters only (SoA, cold values, per-wave table, workgroup kinds).
// Number of virtual waves (runs of <= 64 equal parameter records) the lane order `at(i)` gives.
template <class At>
size_t count_virtual_waves(const std::vector<WelshParams>& P, uint32_t n, At&& at) {
  size_t waves = 0;
  for (uint32_t i = 0; i < n;) {
    uint32_t e = i + 1;
    while (e < n && e - i < 64 && std::memcmp(&P[at(e)], &P[at(i)], sizeof(WelshParams)) == 0) ++e;
    ++waves;
    i = e;
  }
  return waves;
}
inline bool runs_are_long(size_t virtual_waves, uint32_t n) {
  // Use the scalar-parameter kernels when the runs are long (at most 1.5x as many virtual waves as
  // physical ones), or when the bank is so small that even one short run per wave leaves the machine
  // (1,024 SIMDs) under-filled: there a partly filled fast wave beats a full slow one.
  const uint32_t phys_waves = (n + 63) / 64;
  return !(virtual_waves > (size_t)phys_waves + phys_waves / 2 + 8 && virtual_waves > 2048);
}
// `regroup`: the state is (being) reset, so the lane order may be chosen afresh.
int welsh_upload_params(groove_bank* b, bool regroup) {
  groove_ctx* ctx = b->ctx;
  const double sr = ctx->sr;
  const uint32_t n = b->n;
  std::vector<WelshParams> Pext(n);
  std::vector<WelshCold> Cext(n);
  for (uint32_t v = 0; v < n; ++v) Pext[v] = derive_welsh(b->welsh[v], sr, Cext[v]);
  if (ctx->f32_filter) { // WF_FILTER_F32 (derive.h welsh_filter_f32_ok): measured once per distinct filter description
    using Key = F32FilterKey;
    if (b->f32_memo_sr != sr) { b->f32_memo.clear(); b->f32_memo_sr = sr; } // (kept across uploads: a control change re-derives the bank, groove_bank_set_param)
    std::map<Key, bool>& memo = b->f32_memo;
    for (uint32_t v = 0; v < n; ++v) {
      const WelshParams& o = Pext[v];
      Key k{};
      k.c0 = o.fc.c0; k.d1 = o.fc.d1; k.c2 = o.fc.c2; k.d3 = o.fc.d3; k.hz = o.cutoff_hz; k.start = o.cutoff_start; k.end = o.cutoff_end;
      k.depth = (o.flags & WF_LFO_CUTOFF) ? o.lfo_depth : 0.0f; k.bits = o.flags & (WF_RETUNE_ENV | WF_LFO_CUTOFF | WF_LFO_RESO | WF_COEF_WIDE);
      auto it = memo.find(k);
      // (~0.6 ms per distinct filter description: a bank of more than 4,096 of them — no project of the reference's shape, one patch
      // per synth — keeps the f64 recurrence, always safe, for the descriptions beyond, instead of seconds of measuring at upload)
      if (it == memo.end()) it = memo.emplace(k, memo.size() < 4096 ? welsh_filter_f32_ok(o, sr) : false).first;
      if (it->second) Pext[v].flags |= WF_FILTER_F32;
    }
  }
  if (regroup) {
    b->perm.clear(); b->inv.clear();
    if (!runs_are_long(count_virtual_waves(Pext, n, [](uint32_t i) { return i; }), n)) {
      // patches interleaved voice by voice: try the patch-major order (stable sort by a hash of the record)
      std::vector<uint64_t> h(n);
      for (uint32_t v = 0; v < n; ++v) {
        uint64_t x = 1469598103934665603ull;
        const unsigned char* bytes = reinterpret_cast<const unsigned char*>(&Pext[v]);
        for (size_t k = 0; k < sizeof(WelshParams); ++k) { x ^= bytes[k]; x *= 1099511628211ull; }
        h[v] = x;
      }
      std::vector<uint32_t> order(n);
      for (uint32_t v = 0; v < n; ++v) order[v] = v;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return h[a] < h[c]; });
      if (runs_are_long(count_virtual_waves(Pext, n, [&](uint32_t i) { return order[i]; }), n)) {
        b->perm = std::move(order);
        b->inv.resize(n);
        for (uint32_t i = 0; i < n; ++i) b->inv[b->perm[i]] = i;
      }
    }
    b->order.reset(); // blocks filled under the old order keep it alive for as long as they hold that content
    if (!b->perm.empty()) {
      auto o = std::make_shared<LaneOrder>();
      o->n = n;
      GHIP(ctx, hipMalloc(&o->d_src_lane, (size_t)n * sizeof(uint32_t)));
      GHIP(ctx, ctx_memcpy(ctx, o->d_src_lane, b->inv.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
      b->order = std::move(o);
    }
  }
  // everything below is in INTERNAL lane order
  std::vector<WelshParams> P(n);
  std::vector<double> cold((size_t)4 * n);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t v = b->perm.empty() ? i : b->perm[i];
    P[i] = Pext[v];
    const WelshCold& c = Cext[v];
    cold[i] = c.tune1; cold[(size_t)n + i] = c.tune2; cold[(size_t)2 * n + i] = c.fixed1; cold[(size_t)3 * n + i] = c.fixed2;
  }
  if (upload_soa(ctx, b->d_params, P)) return 1;
  GHIP(ctx, ctx_memcpy(ctx, b->d_cold, cold.data(), cold.size() * 8, hipMemcpyHostToDevice));
  b->tp_pairs = true; // time-parallel form, two voices per wavefront: voices 2i and 2i + 1 share their parameter words
  for (uint32_t i = 0; i + 1 < n && b->tp_pairs; i += 2) b->tp_pairs = std::memcmp(&P[i], &P[i + 1], sizeof(WelshParams)) == 0;
  b->tp_full_coef = false; // time-parallel form: a voice with the resonance routing keeps six f64 coefficients per frame (welsh_tp.h)
  for (uint32_t i = 0; i < n; ++i) if (P[i].flags & WF_LFO_RESO) { b->tp_full_coef = true; break; } // (WF_COEF_WIDE patches: from the tangent like the rest, round 6)
  // Virtual waves: maximal runs of consecutive voices with identical parameter words, cut at 64.
  std::vector<WaveDesc> W;
  W.reserve((size_t)n / 64 + 64);
  for (uint32_t v = 0; v < n;) {
    uint32_t e = v + 1;
    while (e < n && e - v < 64 && std::memcmp(&P[e], &P[v], sizeof(WelshParams)) == 0) ++e;
    WaveDesc d;
    d.p = P[v]; d.vbase = v; d.count = e - v;
    W.push_back(d);
    v = e;
  }
  for (uint32_t& c : b->wgs_of_kind) c = 0;
  // Otherwise (every voice its own patch, even patch-major) the per-lane kernel serves the whole bank.
  if (!runs_are_long(W.size(), n)) {
    b->n_vwaves = 0;
    return 0;
  }
  // A workgroup runs in ONE instantiation (kernels.h, "Workgroup KINDS"), so it is built from waves that
  // ask for the same one: the waves are ordered by the kind they need and every kind's last workgroup is
  // filled up with empty waves (count 0).  (Cutting the run order into fours made every workgroup of a
  // small many-patch bank a mixture, which runs in the most demanding base kind with the run-time
  // waveform switches: config #2's 32 waves took 0.21 ms per block where their slowest patch needs 0.16.)
  auto kind_of_wave = [](const WelshParams& p) -> uint16_t {
    const int base = welsh_base_kind(p); // == wg_base_kind_of(welsh_lfo_mode(p), welsh_retunes(p)); dsp_core.h
    int cl, c1, c2;
    welsh_body_classes(p, base, cl, c1, c2);
    return (uint16_t)wg_kind_of(base, cl, c1, c2);
  };
  std::vector<uint16_t> kind; // per workgroup
  std::vector<uint8_t> f32_of; // per workgroup: its waves' patches carry WF_FILTER_F32 (a workgroup is uniform in it too: sort key bit 0)
  {
    std::vector<uint32_t> wave_kind(W.size()); // (kind << 1) | fp32-filter flag
    std::vector<uint32_t> order(W.size());
    for (uint32_t w = 0; w < W.size(); ++w) { wave_kind[w] = ((uint32_t)kind_of_wave(W[w].p) << 1) | ((W[w].p.flags & WF_FILTER_F32) ? 1u : 0u); order[w] = w; }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return wave_kind[a] < wave_kind[c]; });
    std::vector<WaveDesc> packed;
    packed.reserve(W.size() + (size_t)kWaves * 64);
    for (size_t i = 0; i < order.size();) {
      size_t e = i;
      while (e < order.size() && wave_kind[order[e]] == wave_kind[order[i]]) ++e;
      for (size_t j = i; j < e; ++j) {
        if ((j - i) % kWaves == 0) { kind.push_back((uint16_t)(wave_kind[order[i]] >> 1)); f32_of.push_back((uint8_t)(wave_kind[order[i]] & 1u)); }
        packed.push_back(W[order[j]]);
      }
      while (packed.size() % kWaves) { // empty waves: no lane active, the first wave's voice as the shadow address
        WaveDesc pad = W[order[i]];
        pad.count = 0;
        packed.push_back(pad);
      }
      i = e;
    }
    W.swap(packed);
  }
  b->n_vwaves = (uint32_t)W.size();
  const uint32_t wgs = b->n_vwaves / kWaves;
  std::vector<uint32_t> wg_list(wgs);
  std::vector<uint8_t> wg_cls(wgs), wg_base(wgs), wg_f32(wgs);
  {
    std::vector<uint32_t> at(kWgKinds + 1, 0);
    for (uint16_t k : kind) b->wgs_of_kind[k] += 1;
    for (int k = 0; k < kWgKinds; ++k) at[k + 1] = at[k] + b->wgs_of_kind[k];
    for (uint32_t g = 0; g < wgs; ++g) {
      const uint32_t slot = at[kind[g]]++;
      wg_list[slot] = g;
      wg_cls[slot] = (uint8_t)(kind[g] % kClassCombos);
      wg_base[slot] = (uint8_t)(kind[g] / kClassCombos);
      wg_f32[slot] = f32_of[g];
    }
  }
  if (b->vwaves_cap < W.size()) {
    if (b->d_waves) GHIP(ctx, hipFree(b->d_waves));
    b->vwaves_cap = W.size() + W.size() / 8 + 16;
    GHIP(ctx, hipMalloc(&b->d_waves, b->vwaves_cap * sizeof(WaveDesc)));
  }
  if (b->wg_list_cap < wgs) {
    if (b->d_wg_list) GHIP(ctx, hipFree(b->d_wg_list));
    b->wg_list_cap = wgs + wgs / 8 + 16;
    // (twice the capacity: the kind-sorted lists in the first half, the mix kernel's striped copy of them in the second)
    GHIP(ctx, hipMalloc(&b->d_wg_list, 2 * b->wg_list_cap * sizeof(uint32_t)));
    if (b->d_wg_cls) GHIP(ctx, hipFree(b->d_wg_cls));
    GHIP(ctx, hipMalloc(&b->d_wg_cls, 2 * b->wg_list_cap));
    if (b->d_wg_base) GHIP(ctx, hipFree(b->d_wg_base));
    GHIP(ctx, hipMalloc(&b->d_wg_base, 2 * b->wg_list_cap));
    if (b->d_wg_f32) GHIP(ctx, hipFree(b->d_wg_f32));
    GHIP(ctx, hipMalloc(&b->d_wg_f32, 2 * b->wg_list_cap));
  }
  GHIP(ctx, ctx_memcpy(ctx, b->d_waves, W.data(), W.size() * sizeof(WaveDesc), hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, b->d_wg_list, wg_list.data(), (size_t)wgs * sizeof(uint32_t), hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, b->d_wg_cls, wg_cls.data(), wgs, hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, b->d_wg_base, wg_base.data(), wgs, hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, b->d_wg_f32, wg_f32.data(), wgs, hipMemcpyHostToDevice));
  { // the mix kernel's sections: slots s, s + 3, s + 6 ... of the class-specialised part of the sorted list (the exact-f64 kinds, last in it, keep their own kernels)
    uint32_t n_spec = 0;
    for (int k = 0; k < 4 * kClassCombos; ++k) n_spec += b->wgs_of_kind[k];
    std::vector<uint32_t> m_list(n_spec);
    std::vector<uint8_t> m_cls(n_spec), m_base(n_spec), m_f32(n_spec);
    uint32_t at = 0;
    for (uint32_t sec = 0; sec < 3; ++sec) {
      b->mix_off[sec] = at;
      for (uint32_t g = sec; g < n_spec; g += 3, ++at) { m_list[at] = wg_list[g]; m_cls[at] = wg_cls[g]; m_base[at] = wg_base[g]; m_f32[at] = wg_f32[g]; }
      b->mix_cnt[sec] = at - b->mix_off[sec];
    }
    if (n_spec) {
      GHIP(ctx, ctx_memcpy(ctx, b->d_wg_list + b->wg_list_cap, m_list.data(), (size_t)n_spec * sizeof(uint32_t), hipMemcpyHostToDevice));
      GHIP(ctx, ctx_memcpy(ctx, b->d_wg_cls + b->wg_list_cap, m_cls.data(), n_spec, hipMemcpyHostToDevice));
      GHIP(ctx, ctx_memcpy(ctx, b->d_wg_base + b->wg_list_cap, m_base.data(), n_spec, hipMemcpyHostToDevice));
      GHIP(ctx, ctx_memcpy(ctx, b->d_wg_f32 + b->wg_list_cap, m_f32.data(), n_spec, hipMemcpyHostToDevice));
    }
  }
  return 0;
}

int bank_derive_and_upload(groove_bank* b) {
  groove_ctx* ctx = b->ctx;
  const double sr = ctx->sr;
  const uint32_t n = b->n;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (b->kind == BANK_WELSH) {
    std::vector<WelshState> S(n, initial_welsh_state()); // every voice starts from the same state: any lane order will do
    if (welsh_upload_params(b, true)) return 1;
    if (upload_soa(ctx, b->d_state, S)) return 1;
  } else if (b->kind == BANK_FM) {
    std::vector<FmParams> P(n);
    std::vector<FmState> S(n, initial_fm_state());
    std::vector<double> ratio(n);
    for (uint32_t v = 0; v < n; ++v) { P[v] = derive_fm(b->fm[v], sr); ratio[v] = b->fm[v].ratio; }
    if (upload_soa(ctx, b->d_params, P)) return 1;
    if (upload_soa(ctx, b->d_state, S)) return 1;
    GHIP(ctx, ctx_memcpy(ctx, b->d_cold, ratio.data(), ratio.size() * 8, hipMemcpyHostToDevice));
  } else {
    std::vector<SamplerParams> P(n);
    std::vector<SamplerState> S(n, SamplerState{0, 0, 0, 0});
    for (uint32_t v = 0; v < n; ++v) {
      const groove_sampler_params& sp = b->sampler[v];
      const groove_sample_desc& d = b->descs[sp.sample_index < b->descs.size() ? sp.sample_index : 0];
      P[v] = SamplerParams{(uint32_t)d.offset, d.length, (double)d.root_hz, sp.gain, sp.one_shot};
    }
    if (upload_soa(ctx, b->d_params, P)) return 1;
    if (upload_soa(ctx, b->d_state, S)) return 1;
  }
  b->pending.clear();
  return 0;
}

int bank_alloc(groove_bank* b) {
  groove_ctx* ctx = b->ctx;
  GHIP(ctx, hipSetDevice(ctx->device));
  // soa_load / soa_store (kernels.h) address a record through ONE buffer resource: its byte size and the
  // row offsets are 32-bit, so words * n * 4 must stay below 4 GiB (26.8 M Welsh voices per bank).
  if ((uint64_t)std::max(b->pw, b->sw) * b->n * 4ull >= (1ull << 32))
    return fail(ctx, "bank too large: parameter/state record x voices must stay below 4 GiB per bank (split the voices over several banks)");
  GHIP(ctx, hipMalloc(&b->d_params, (size_t)b->pw * b->n * 4));
  GHIP(ctx, hipMalloc(&b->d_state, (size_t)b->sw * b->n * 4));
  return 0;
}

// Launch one note-event round resident at ev[0..count) on the device.
int launch_events(groove_bank* b, const groove_note_event* ev, uint32_t count, int all_event, hipStream_t st) {
  groove_ctx* ctx = b->ctx;
  const uint32_t items = all_event >= 0 ? b->n : count;
  const dim3 grid(blocks_for(items)), blk(kThreads);
  if (b->kind == BANK_WELSH)
    hipLaunchKernelGGL(welsh_events_kernel, grid, blk, 0, st, ev, count, all_event, b->d_params,
                       b->d_state, b->d_cold, b->n, (double)ctx->sr);
  else if (b->kind == BANK_FM)
    hipLaunchKernelGGL(fm_events_kernel, grid, blk, 0, st, ev, count, all_event, b->d_params,
                       b->d_state, b->d_cold, b->n, (double)ctx->sr);
  else
    hipLaunchKernelGGL(sampler_events_kernel, grid, blk, 0, st, ev, count, all_event, b->d_params,
                       b->d_state, b->n);
  GHIP(ctx, hipGetLastError());
  return 0;
}

// Apply queued events in order.  Events are cut into rounds in which every voice appears
// at most once, so one thread per event is race-free and order is preserved across rounds.
// `inline_ok`: the caller is about to launch the bank's time-parallel render, which can take a short sorted list of
// events in its arguments (welsh_tp.h, InlineEvents).
int flush_events(groove_bank* b, bool inline_ok = false) {
  if (b->inline_ev.n) { // handed to a render that never ran: they come first, through the ordinary path
    b->pending.insert(b->pending.begin(), b->inline_ev.ev, b->inline_ev.ev + b->inline_ev.n);
    b->inline_ev.n = 0;
    inline_ok = false;
  }
  if (b->pending.empty()) return 0;
  groove_ctx* ctx = b->ctx;
  if (inline_ok && b->kind == BANK_SAMPLER && b->inv.empty() && b->pending.size() <= kInlineEvents) {
    bool sorted = b->pending[0].voice != GROOVE_ALL_VOICES;
    for (size_t i = 1; sorted && i < b->pending.size(); ++i) sorted = b->pending[i].voice != GROOVE_ALL_VOICES && b->pending[i].voice > b->pending[i - 1].voice;
    if (sorted) {
      b->inline_ev.n = (uint32_t)b->pending.size();
      std::memcpy(b->inline_ev.ev, b->pending.data(), b->pending.size() * sizeof(groove_note_event));
      b->pending.clear();
      return 0;
    }
  }
  // Which stream applies the events.  A bank whose renders all run on ONE side stream (side_mode 2: FM,
  // sampler, small or per-lane Welsh banks in the asynchronous forms) and whose state the ctx stream has not
  // touched since gets its events on that same stream: they are ordered with its renders, nothing else reads
  // its state, and the other banks' streams never notice (a project with note events in every block —
  // config #4's staggered starts inside config #5 — would otherwise join and re-fork every stream once per
  // block: mixed-131072 0.229 -> 0.13 ms per block).  Otherwise: the ctx stream, after joining the side streams.
  hipStream_t st = ctx->stream;
  const bool own_stream = b->side_mode == 2 && !b->ctx_touched;
  if (own_stream) {
    st = side_stream_of(ctx, b->stream_slot);
    ctx->side_busy[b->stream_slot] = true;
  } else {
    b->ctx_touched = true;
    if (ctx_join(ctx)) return 1;
  }
  // caller's voice index -> internal lane, into a copy: a failure below leaves `pending` as it was queued,
  // so the next flush maps it exactly once
  std::vector<groove_note_event> ev = b->pending;
  if (!b->inv.empty())
    for (groove_note_event& e : ev)
      if (e.voice != GROOVE_ALL_VOICES) e.voice = b->inv[e.voice];
  // Device and pinned host staging, two slots each in rotation: the events of this flush are handed to the
  // stream and the call returns without a host synchronisation.
  const int hs = b->ev_slot;
  b->ev_slot ^= 1;
  if (b->staged[hs]) GHIP(ctx, wait_deadline(ctx, nullptr, b->ev_staged[hs], "note-event staging slot")); // two flushes ago: long done
  if (b->h_ev_cap[hs] < ev.size()) {
    if (b->h_ev[hs]) GHIP(ctx, hipHostFree(b->h_ev[hs]));
    if (b->d_ev[hs]) GHIP(ctx, hipFree(b->d_ev[hs]));
    b->h_ev_cap[hs] = std::max<size_t>(ev.size() * 2, 1024);
    GHIP(ctx, hipHostMalloc(&b->h_ev[hs], b->h_ev_cap[hs] * sizeof(groove_note_event), hipHostMallocDefault));
    GHIP(ctx, hipMalloc(&b->d_ev[hs], b->h_ev_cap[hs] * sizeof(groove_note_event)));
  }
  if (!b->ev_staged[hs]) GHIP(ctx, hipEventCreateWithFlags(&b->ev_staged[hs], hipEventDisableTiming));
  std::memcpy(b->h_ev[hs], ev.data(), ev.size() * sizeof(groove_note_event));
  groove_note_event* d_ev = b->d_ev[hs];
  GHIP(ctx, hipMemcpyAsync(d_ev, b->h_ev[hs], ev.size() * sizeof(groove_note_event), hipMemcpyHostToDevice, st));
  // fast path: strictly increasing voices, no ALL events → one round
  bool sorted = true;
  for (size_t i = 0; i < ev.size(); ++i) {
    if (ev[i].voice == GROOVE_ALL_VOICES || (i && ev[i].voice <= ev[i - 1].voice)) { sorted = false; break; }
  }
  if (sorted) {
    if (launch_events(b, d_ev, (uint32_t)ev.size(), -1, st)) return 1;
  } else {
    // general path: contiguous runs; a run ends at an ALL event or at a repeated voice
    std::vector<uint32_t> seen_round(b->n, 0);
    uint32_t round = 1;
    size_t start = 0;
    auto launch_run = [&](size_t lo, size_t hi) -> int {
      if (hi <= lo) return 0;
      return launch_events(b, d_ev + lo, (uint32_t)(hi - lo), -1, st);
    };
    for (size_t i = 0; i < ev.size(); ++i) {
      if (ev[i].voice == GROOVE_ALL_VOICES) {
        if (launch_run(start, i)) return 1;
        if (launch_events(b, d_ev, (uint32_t)ev.size(), (int)i, st)) return 1;
        start = i + 1; ++round;
      } else if (ev[i].voice < b->n) {
        if (seen_round[ev[i].voice] == round) {
          if (launch_run(start, i)) return 1;
          start = i; ++round;
        }
        seen_round[ev[i].voice] = round;
      }
    }
    if (launch_run(start, ev.size())) return 1;
  }
  GHIP(ctx, hipEventRecord(b->ev_staged[hs], st));
  b->staged[hs] = true;
  b->pending.clear(); // (the device reads the pinned copy, not this vector)
  return 0;
}

int ensure_partial(groove_ctx* ctx, size_t floats) {
  if (ctx->partial_cap >= floats) return 0;
  if (ctx->d_partial) GHIP(ctx, hipFree(ctx->d_partial));
  GHIP(ctx, hipMalloc(&ctx->d_partial, floats * 4));
  ctx->partial_cap = floats;
  return 0;
}

// The segment buffer of a bus reduction: seg[segs][cols].
int ensure_seg_buffer(groove_ctx* ctx, float** buf, size_t* cap, size_t seg_floats) {
  if (*buf && *cap >= seg_floats) return 0;
  if (*buf) { GHIP(ctx, hipFree(*buf)); *buf = nullptr; *cap = 0; }
  GHIP(ctx, hipMalloc(buf, seg_floats * 4));
  *cap = seg_floats;
  return 0;
}
// bus[f][ch] (+)= column sums of partial[row][ch][frame] on the ctx stream (fixed order: segments of 64 rows, then the
// segments in index order; a single segment's sums go straight to the bus).
constexpr uint32_t kRowsPerSeg = 64;
constexpr size_t kMaxSpareSums = 8; // lane-sum buffers groove_mix_deferred keeps for blocks to take (deferred_taken)
// `done` (optional): an event that completes with the LAST kernel of the reduction — bound to that dispatch's own completion
// signal (hipExtLaunchKernelGGL), not recorded behind it: a recorded event is a barrier packet of its own, ~5 us of the
// stream's timeline (docs/STREAMS.md).
void launch_reduce(groove_ctx* ctx, const float* partial, uint32_t rows, uint32_t frames, float* seg_buf, float* bus_dev, int accumulate, hipEvent_t done = nullptr);
void launch_reduce(groove_ctx* ctx, const float* partial, uint32_t rows, uint32_t frames, float* seg_buf, float* bus_dev, int accumulate, hipEvent_t done) {
  const uint32_t cols = 2 * frames, segs = (rows + kRowsPerSeg - 1) / kRowsPerSeg;
  const dim3 blk(kThreads);
  if (segs == 1) {
    if (done) hipExtLaunchKernelGGL(partial_rows_kernel, dim3(blocks_for(cols), 1), blk, 0, ctx->stream, nullptr, done, 0, partial, rows, cols, kRowsPerSeg, seg_buf, bus_dev, accumulate);
    else hipLaunchKernelGGL(partial_rows_kernel, dim3(blocks_for(cols), 1), blk, 0, ctx->stream, partial, rows, cols, kRowsPerSeg, seg_buf, bus_dev, accumulate);
  } else {
    hipLaunchKernelGGL(partial_rows_kernel, dim3(blocks_for(cols), segs), blk, 0, ctx->stream, partial, rows, cols, kRowsPerSeg, seg_buf, (float*)nullptr, 0);
    if (done) hipExtLaunchKernelGGL(partial_final_kernel, dim3(blocks_for(cols)), blk, 0, ctx->stream, nullptr, done, 0, seg_buf, segs, frames, bus_dev, accumulate);
    else hipLaunchKernelGGL(partial_final_kernel, dim3(blocks_for(cols)), blk, 0, ctx->stream, seg_buf, segs, frames, bus_dev, accumulate);
  }
}
// The pending rows have been handed to a launch (or to bus_flush's reduction): a buffer groove_mix_deferred took from a block is
// a spare from now on (stream order protects it: whoever is given it next writes it behind that launch — DESIGN.md section 5).
void deferred_taken(groove_ctx* ctx) {
  if (ctx->deferred.rows && ctx->deferred.owned_cap) {
    ctx->spare_sums.emplace_back(const_cast<float*>(ctx->deferred.rows), ctx->deferred.owned_cap);
    if (ctx->spare_sums.size() > kMaxSpareSums) { // (a rotation holds one spare per block in flight; beyond that the list only grows when blocks die)
      // the smallest goes; hipFree waits for the device, which is why this only happens past the cap
      auto it = std::min_element(ctx->spare_sums.begin(), ctx->spare_sums.end(), [](const auto& x, const auto& y) { return x.second < y.second; });
      if (it->first != ctx->deferred.rows) { (void)hipFree(it->first); ctx->spare_sums.erase(it); }
    }
  }
  ctx->deferred.rows = nullptr;
  ctx->deferred.owned_cap = 0;
}
void launch_reduce(groove_ctx* ctx, const float* partial, uint32_t rows, uint32_t frames, float* seg_buf, float* bus_dev, int accumulate, hipEvent_t done);
// The pending reduction of a bank's last paced block onto its bus (ctx stream).  host_wait: the HOST waits for the block's render
// kernels (they were submitted a whole call ago), so that the ctx stream carries no cross-queue wait; otherwise (flush points) the
// ctx stream waits for them itself.
int paced_reduce(groove_bank* b, bool host_wait) {
  if (!b->paced.active) return 0;
  groove_ctx* ctx = b->ctx;
  const auto p = b->paced;
  // The fallible steps FIRST: a wait whose deadline passes (return code 2) leaves the record pending — the next call, or any flush
  // point, tries again — so a caller that carries on after a timeout never gets a bus that silently lacks this block.
  for (int k = 0; k < kSideStreams; ++k) {
    if (!(p.used & (1u << k))) continue;
    if (host_wait) GHIP(ctx, wait_deadline(ctx, nullptr, b->ev_render_done[k][p.slot], "groove_bank_render_mix_paced: the previous block's render"));
    else GHIP(ctx, hipStreamWaitEvent(ctx->stream, b->ev_render_done[k][p.slot], 0));
  }
  b->paced.active = false;
  ctx->paced_order.erase(std::remove(ctx->paced_order.begin(), ctx->paced_order.end(), b), ctx->paced_order.end());
  launch_reduce(ctx, b->d_pipe_part[p.slot], p.rows, p.frames, b->d_pipe_seg[p.slot], p.bus, p.accumulate, nullptr);
  GHIP(ctx, hipEventRecord(b->ev_reduce_done[p.slot], ctx->stream));
  b->reduce_recorded[p.slot] = true;
  GHIP(ctx, hipGetLastError());
  return 0;
}
int bus_flush_deferred(groove_ctx* ctx);
// ---- the all-pass stream (groove_set_fx_allpass_stream)
static hipStream_t ap_stream(groove_ctx* ctx) { return side_stream_of(ctx, ctx->fx_ap_stream); }
static int ap_events(groove_ctx* ctx) {
  if (!ctx->ev_ap_run) GHIP(ctx, hipEventCreateWithFlags(&ctx->ev_ap_run, kSyncEventFlags));
  if (!ctx->ev_ap_x) GHIP(ctx, hipEventCreateWithFlags(&ctx->ev_ap_x, kSyncEventFlags));
  return 0;
}
// The ctx stream behind everything the all-pass stream has been given ...
int ap_join(groove_ctx* ctx) {
  if (ctx->fx_ap_stream < 0 || !ctx->side_busy[ctx->fx_ap_stream]) return 0;
  if (ap_events(ctx)) return 1;
  GHIP(ctx, hipEventRecord(ctx->ev_ap_x, ap_stream(ctx)));
  GHIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_ap_x, 0));
  return 0;
}
// ... and the all-pass stream behind everything the ctx stream has been given.
static int ap_follow(groove_ctx* ctx) {
  if (ctx->fx_ap_stream < 0) return 0;
  if (ap_events(ctx)) return 1;
  GHIP(ctx, hipEventRecord(ctx->ev_ap_run, ctx->stream));
  GHIP(ctx, hipStreamWaitEvent(ap_stream(ctx), ctx->ev_ap_run, 0));
  return 0;
}
void launch_reduce(groove_ctx* ctx, const float* partial, uint32_t rows, uint32_t frames, float* seg_buf, float* bus_dev, int accumulate, hipEvent_t done);
// Lane sums that wait for the next all-pass launch and will not get one: the ctx stream reduces them itself, behind the kernel that
// wrote them and ahead of whichever all-pass writes that buffer next.
int bus_flush_ap(groove_ctx* ctx) {
  if (!ctx->deferred_ap.rows) return 0;
  const auto d = ctx->deferred_ap;
  ctx->deferred_ap.rows = nullptr;
  if (ap_join(ctx)) return 1;
  if (ensure_seg_buffer(ctx, &ctx->d_fseg, &ctx->fseg_cap, (size_t)((d.n_rows + kRowsPerSeg - 1) / kRowsPerSeg) * 2 * d.frames)) return 1;
  launch_reduce(ctx, d.rows, d.n_rows, d.frames, ctx->d_fseg, d.bus, d.accumulate, nullptr);
  if (hipGetLastError() != hipSuccess) return fail(ctx, "bus_flush: launch failed");
  return ap_follow(ctx);
}
int bus_flush(groove_ctx* ctx) {
  if (bus_flush_deferred(ctx)) return 1;
  if (bus_flush_ap(ctx)) return 1;
  while (!ctx->paced_order.empty()) // (call order: the order of the banks' sums on a bus)
    if (const int rc = paced_reduce(ctx->paced_order.front(), false)) return rc;
  return 0;
}
int bus_flush_deferred(groove_ctx* ctx) {
  if (!ctx->deferred.rows) return 0;
  const auto d = ctx->deferred;
  deferred_taken(ctx);
  if (ensure_seg_buffer(ctx, &ctx->d_fseg, &ctx->fseg_cap, (size_t)((d.n_rows + kRowsPerSeg - 1) / kRowsPerSeg) * 2 * d.frames)) return 1;
  launch_reduce(ctx, d.rows, d.n_rows, d.frames, ctx->d_fseg, d.bus, d.accumulate);
  return hipGetLastError() == hipSuccess ? 0 : fail(ctx, "bus_flush: launch failed");
}
int reduce_rows(groove_ctx* ctx, const float* rows_dev, uint32_t rows, uint32_t frames, float* bus_dev, int accumulate, hipEvent_t done = nullptr) {
  const uint32_t cols = 2 * frames, segs = (rows + kRowsPerSeg - 1) / kRowsPerSeg;
  if (ensure_seg_buffer(ctx, &ctx->d_fseg, &ctx->fseg_cap, (size_t)segs * cols)) return 1;
  launch_reduce(ctx, rows_dev, rows, frames, ctx->d_fseg, bus_dev, accumulate, done);
  GHIP(ctx, hipGetLastError());
  return 0;
}

constexpr uint32_t kMixSeg = 16384; // floats of one row summed by one workgroup

int mix_one(groove_ctx* ctx, const groove_block* b, uint32_t frames, float* bus, int accumulate, size_t planar_stride = 0) {
  const uint32_t n_seg = (b->n + kMixSeg - 1) / kMixSeg;
  const uint32_t rows = 2 * frames;
  if (ensure_partial(ctx, (size_t)rows * n_seg)) return 1;
  if (n_seg == 1) { // one segment per row: the partial sums are the totals
    hipLaunchKernelGGL(mix_partial_kernel, dim3(1, rows), dim3(kThreads), 0, ctx->stream, block_data(b), b->n, frames,
                       (size_t)b->cap * b->n, kMixSeg, ctx->d_partial, 1u, bus, accumulate, planar_stride);
    GHIP(ctx, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(mix_partial_kernel, dim3(n_seg, rows), dim3(kThreads), 0, ctx->stream, block_data(b), b->n, frames,
                     (size_t)b->cap * b->n, kMixSeg, ctx->d_partial, n_seg, (float*)nullptr, 0, (size_t)0);
  hipLaunchKernelGGL(mix_final_kernel, dim3(blocks_for(rows)), dim3(kThreads), 0, ctx->stream, ctx->d_partial,
                     frames, n_seg, bus, accumulate, planar_stride);
  GHIP(ctx, hipGetLastError());
  return 0;
}

// ---- effects -------------------------------------------------------------------------
int fx_upload_params(groove_fx* fx) {
  groove_ctx* ctx = fx->ctx;
  const uint32_t n = fx->n;
  const double sr = ctx->sr;
  std::vector<float> fa(n), fb(n), wet(n);
  std::vector<uint32_t> ua(n);
  fx->all_wet = true;
  for (uint32_t i = 0; i < n; ++i) {
    const groove_fx_params& p = fx->p[i];
    wet[i] = p.wet;
    if (p.wet < 1.0f) fx->all_wet = false;
    ua[i] = p.bits > 31 ? 31 : p.bits;
    switch (fx->kind) {
      case GROOVE_FX_GAIN: fa[i] = p.ceiling; break;
      case GROOVE_FX_LIMITER:
      case GROOVE_FX_COMPRESSOR: fa[i] = p.limit_min; fb[i] = p.limit_max; break;
      case GROOVE_FX_REVERB: fa[i] = p.attenuation; break;
      default: break;
    }
  }
  GHIP(ctx, ctx_memcpy(ctx, fx->d_fa, fa.data(), n * 4, hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, fx->d_fb, fb.data(), n * 4, hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, fx->d_ua, ua.data(), n * 4, hipMemcpyHostToDevice));
  GHIP(ctx, ctx_memcpy(ctx, fx->d_wet, wet.data(), n * 4, hipMemcpyHostToDevice));
  double probe[5];
  if (rbj_for_kind_h(fx->kind, fx->p[0], sr, probe)) { // any BiQuad 12 dB mode
    std::vector<double> c((size_t)5 * n);
    for (uint32_t i = 0; i < n; ++i) {
      double c5[5];
      rbj_for_kind_h(fx->kind, fx->p[i], sr, c5);
      for (int k = 0; k < 5; ++k) c[(size_t)k * n + i] = c5[k];
    }
    GHIP(ctx, ctx_memcpy(ctx, fx->d_coef, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  } else if (fx->kind == GROOVE_FX_BIQUAD_LP24) {
    std::vector<double> c((size_t)6 * n);
    for (uint32_t i = 0; i < n; ++i) {
      double c6[6];
      lp24_coeffs_h(fx->p[i].cutoff_hz, fx->p[i].passband_ripple, sr, c6);
      for (int k = 0; k < 6; ++k) c[(size_t)k * n + i] = c6[k];
    }
    GHIP(ctx, ctx_memcpy(ctx, fx->d_coef, c.data(), c.size() * 8, hipMemcpyHostToDevice));
  }
  return 0;
}

int fx_setup_state(groove_fx* fx) { // (re)allocate and zero state for the current sample rate
  groove_ctx* ctx = fx->ctx;
  const uint32_t n = fx->n;
  const double sr = ctx->sr;
  const size_t ln = 2 * (size_t)n;
  if (fx->d_st) { GHIP(ctx, hipFree(fx->d_st)); fx->d_st = nullptr; }
  if (fx->d_ring) { GHIP(ctx, hipFree(fx->d_ring)); fx->d_ring = nullptr; }
  fx->ring_rows = 0; fx->w = 0;
  const groove_fx_params& p0 = fx->p[0];
  switch (fx->kind) {
    case GROOVE_FX_BIQUAD_LP12:
    case GROOVE_FX_BIQUAD_HP12:
    case GROOVE_FX_BIQUAD_BP12:
    case GROOVE_FX_BIQUAD_BS12:
    case GROOVE_FX_BIQUAD_AP12:
    case GROOVE_FX_BIQUAD_PEAK12:
    case GROOVE_FX_BIQUAD_LSHELF12:
    case GROOVE_FX_BIQUAD_HSHELF12:
    case GROOVE_FX_BIQUAD_LP24:
      GHIP(ctx, hipMalloc(&fx->d_st, 4 * ln * 8));
      GHIP(ctx, hipMemsetAsync(fx->d_st, 0, 4 * ln * 8, ctx->stream)); // on the ctx stream: a null-stream memset is not ordered with it
      break;
    case GROOVE_FX_DELAY:
      fx->N = delay_frames_h(p0.delay_seconds, sr);
      fx->ring_rows = fx->N;
      break;
    case GROOVE_FX_CHORUS:
      fx->N = delay_frames_h(p0.delay_seconds, sr);
      if (p0.voices > 64) return fail(ctx, "groove_fx: chorus voices > 64"); // the taps are a per-frame loop on the device
      fx->voices = p0.voices < 1 ? 1 : p0.voices;
      fx->spacing = fx->N / fx->voices;
      fx->ring_rows = fx->N;
      break;
    case GROOVE_FX_REVERB: {
      uint64_t base = 0;
      for (int i = 0; i < 6; ++i) {
        const double d = i < 4 ? kCombDelaysH[i] : kAllpassDelaysH[i - 4];
        fx->geo.N[i] = delay_frames_h(d, sr);
        fx->geo.w[i] = 0;
        fx->geo.base[i] = base;
        fx->geo.g[i] = (float)(i < 4 ? decay_gain_h(d, p0.reverb_seconds) : decay_gain_h(d, kAllpassDecaysH[i - 4]));
        base += fx->geo.N[i];
      }
      for (int i = 0; i < 2; ++i) { fx->ap_alt[i] = base; base += fx->geo.N[4 + i]; }
      fx->ring_rows = base;
      break;
    }
    default: break;
  }
  if (fx->ring_rows > (1ull << 26)) return fail(ctx, "groove_fx: delay line longer than 2^26 frames");
  if (fx->ring_rows) {
    GHIP(ctx, hipMalloc(&fx->d_ring, fx->ring_rows * ln * 4));
    GHIP(ctx, hipMemsetAsync(fx->d_ring, 0, fx->ring_rows * ln * 4, ctx->stream));
  }
  return 0;
}

int fx_check_uniform(groove_fx* fx) {
  const groove_fx_params& a = fx->p[0];
  for (uint32_t i = 1; i < fx->n; ++i) {
    const groove_fx_params& b = fx->p[i];
    if ((fx->kind == GROOVE_FX_CHORUS && (a.voices != b.voices || a.delay_seconds != b.delay_seconds)) ||
        (fx->kind == GROOVE_FX_DELAY && a.delay_seconds != b.delay_seconds) ||
        (fx->kind == GROOVE_FX_REVERB && a.reverb_seconds != b.reverb_seconds))
      return fail(fx->ctx, "groove_fx: delay-line geometry (voices / delay_seconds / reverb_seconds) must be uniform across the lanes of one effect bank");
  }
  return 0;
}

// ---- RCCL via dlopen -----------------------------------------------------------------
// Function-pointer types are taken from the prototypes in <rccl/rccl.h>, so a change of the ABI is a
// compile error here instead of a silently wrong call.
using nccl_get_uid_fn = decltype(&ncclGetUniqueId);
using nccl_init_rank_fn = decltype(&ncclCommInitRank);
using nccl_reduce_fn = decltype(&ncclReduce);
using nccl_destroy_fn = decltype(&ncclCommDestroy);
using nccl_count_fn = decltype(&ncclCommCount);
using nccl_errstr_fn = decltype(&ncclGetErrorString);
static_assert(sizeof(ncclUniqueId) == 128, "groove_comm_unique_id hands out a 128-byte id");

void* g_rccl = nullptr; // one RCCL per process, whichever ctx asks first
int rccl_open(groove_ctx* ctx) {
  if (!g_rccl) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      g_rccl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl) break;
    }
  }
  if (!g_rccl) return fail(ctx, std::string("dlopen(librccl) failed: ") + dlerror());
  if (ctx) ctx->rccl = g_rccl;
  return 0;
}

// Test hook (groove_debug_spin): one wave that sleeps until `ticks` of the constant-rate wall clock have passed.
__global__ void debug_spin_kernel(uint64_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

} // namespace

extern "C" {

// ============================================================================ context
// The library's streams, in the order the design of DESIGN.md section 7 wants them created: ctx (highest priority),
// three normal-priority kind streams, the placeholder in fifth place, then the low-priority bank streams.
static bool create_streams(groove_ctx* ctx) {
  // The runtime spreads the streams of one priority over a handful of hardware queues, and streams that
  // share a queue run one after the other.  The ctx stream is created at the highest priority: that
  // gives it a hardware queue of its own, so that a bank's side stream can never land behind it (as a
  // normal-priority stream it shared a queue with the first bank stream and the render-ahead overlap of
  // config #3 was gone: 0.25 ms per block against 0.14), and its short bus reductions, which every
  // pipelined block waits for, are dispatched ahead of the long render kernels.
  int prio_least = 0, prio_greatest = 0;
  bool ok = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess;
  if (ok && ctx->safe_streams) prio_least = prio_greatest = 0;
  auto make = [&](hipStream_t* st, int prio) {
    const bool made = hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) == hipSuccess;
    if (made) ctx->streams_created += 1;
    return made;
  };
  ok = ok && make(&ctx->stream, prio_greatest) && hipEventCreateWithFlags(&ctx->ev_fork, kSyncEventFlags) == hipSuccess;
  // Normal-priority streams for the four class-specialised Welsh kinds (side by side in every block
  // of a big bank; the two exact-f64-LFO kinds, rare, share the first two), and three LOW-priority streams
  // for the single-kernel banks (side by side in a mixed project): never more streams than hardware queues,
  // so no two of them are serialised behind each other by the runtime (see bank_streams).
  // SAFE layout (GROOVE_SAFE_STREAMS=1): one priority and four streams in all — the ctx stream and three side streams
  // that kinds and banks share — so that no two streams of this library ever share a hardware queue and nothing rests
  // on the order in which the process created its streams.  Measured cost: about 8 % at 1,000,000 voices in round 3; none since the
  // walks are paced by the host (profiles/r05_layout_ab.log).
  for (int i = 0; ok && i < kSideStreams; ++i) {
    if (i >= kBaseKinds + ctx->bank_streams) { ok = hipEventCreateWithFlags(&ctx->ev_join[i], kSyncEventFlags) == hipSuccess; continue; }
    if (i == 4 || i == 5) ctx->side_stream[i] = ctx->side_stream[i - 4]; // (safe layout; otherwise re-pointed below)
    else if (i == 3 && ctx->kind_streams == 3) { // three normal-priority streams (see kind_streams); the fifth stream the
      // process creates lands on the ctx stream's hardware queue (so does the ninth): that place is taken by a stream nobody uses
      if (!ctx->safe_streams) ok = make(&ctx->placeholder_stream, 0);
      ctx->side_stream[3] = ctx->side_stream[0];
    }
    else if (i < kBaseKinds) ok = make(&ctx->side_stream[i], 0);
    else if (ctx->safe_streams) ctx->side_stream[i] = ctx->side_stream[(i - kBaseKinds) % 3];
    else ok = make(&ctx->side_stream[i], prio_least);
    ok = ok && hipEventCreateWithFlags(&ctx->ev_join[i], kSyncEventFlags) == hipSuccess;
  }
  // The two exact-f64-LFO kinds run on a stream of their own (round 6; they shared the first two kind streams before): behind another
  // kind's kernel on its stream their kernel's time per block was added to that stream's — the step's long pole with the
  // library-proportioned table (profiles/r06_*).  Which stream: the normal-priority PLACEHOLDER (created so that no working stream shares
  // the ctx stream's hardware queue; the exact kinds' kernel and the ctx stream's two short reductions per block get along on it) —
  // 0.382 / 0.382 ms per block against 0.392 / 0.389 on the first (low-priority) bank stream, in one job, tools/ab_exact_stream.sh, once the
  // kind's register budget let its workgroups be placed at all (kernels.h GROOVE_WAVES_F64).  GROOVE_EXACT_STREAM=bank: the bank stream.
  if (ok && !ctx->safe_streams) {
    const char* e = std::getenv("GROOVE_EXACT_STREAM");
    const bool bank = e && e[0] == 'b';
    if (!bank && ctx->placeholder_stream) ctx->side_stream[4] = ctx->side_stream[5] = ctx->placeholder_stream;
    else if (ctx->bank_streams > 0 && ctx->side_stream[kBaseKinds]) ctx->side_stream[4] = ctx->side_stream[5] = ctx->side_stream[kBaseKinds];
  }
  return ok;
}
static bool side_stream_owned(const groove_ctx* ctx, int i) {
  if (!ctx->side_stream[i] || i == 4 || i == 5 || (i == 3 && ctx->kind_streams == 3)) return false;
  return !(ctx->safe_streams && i >= kBaseKinds);
}
static int comm_init_rank(groove_ctx* ctx, const uint8_t id[128], int rank, int world_size);
static int init_impl(int device_ordinal, const uint8_t* comm_id, int rank, int world_size, groove_ctx** out) {
  if (!out) return fail(nullptr, "groove_init: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, std::string("groove_init: no HIP device available (") + hipGetErrorString(e) +
                             "); this library has no CPU path");
  if (device_ordinal < 0 || device_ordinal >= count) return fail(nullptr, "groove_init: bad device ordinal");
  groove_ctx* ctx = new (std::nothrow) groove_ctx();
  if (!ctx) return fail(nullptr, "groove_init: out of memory");
  ctx->device = device_ordinal;
  if (const char* e = std::getenv("GROOVE_FX_SEQ_ALLPASS")) ctx->seq_allpass = e[0] == '1';
  if (const char* e = std::getenv("GROOVE_SAFE_STREAMS")) ctx->safe_streams = e[0] == '1';
  if (const char* e = std::getenv("GROOVE_MIX_KERNEL")) ctx->mix_kernel = e[0] != '0';
  if (const char* e = std::getenv("GROOVE_LOOK_AHEAD")) ctx->look_ahead = (uint32_t)std::strtoul(e, nullptr, 10) & 7u;
  if (const char* e = std::getenv("GROOVE_SYNC_TIMEOUT_MS")) ctx->sync_timeout_ms = (uint32_t)std::strtoul(e, nullptr, 10);
  if (const char* e = std::getenv("GROOVE_FM_TP_VPW4_MIN_VOICES")) ctx->fm_tp_vpw4_min_voices = (uint32_t)std::strtoul(e, nullptr, 10);
  if (const char* e = std::getenv("GROOVE_BIND_EVENTS")) ctx->bind_events = std::atoi(e) != 0;
  if (const char* e = std::getenv("GROOVE_FX_CHUNKED_ALLPASS")) ctx->chunked_allpass = e[0] == '1';
  if (const char* e = std::getenv("GROOVE_TP_MAX_VOICES")) ctx->tp_max_voices = (uint32_t)std::strtoul(e, nullptr, 10);
  if (const char* e = std::getenv("GROOVE_SPLIT_MAX_WAVES")) ctx->split_max_waves = (uint32_t)std::strtoul(e, nullptr, 10);   // (A/B runs: tools/ab_env.sh)
  if (const char* e = std::getenv("GROOVE_SPLIT2_MAX_WAVES")) ctx->split2_max_waves = (uint32_t)std::strtoul(e, nullptr, 10);
  if (const char* e = std::getenv("GROOVE_SPLIT_ROLES")) { const int r = std::atoi(e); ctx->split_roles = r == 2 || r == 4 ? r : 3; }
  if (const char* e = std::getenv("GROOVE_F32_FILTER")) ctx->f32_filter = e[0] != '0'; // (A/B and the bit-identity tests between kernel forms)
  if (const char* e = std::getenv("GROOVE_PIPELINE_MIN_WAVES")) ctx->pipeline_min_waves = (uint32_t)std::strtoul(e, nullptr, 10); // tests force the pipeline on small banks
  bool ok = hipSetDevice(device_ordinal) == hipSuccess;
  // groove_init_comm: the rank's RCCL communicator first, so that whatever streams RCCL creates for itself exist BEFORE
  // the library's five, which then follow each other in the process's creation order as section 7's layout assumes
  // (with groove_comm_init after groove_init, RCCL's streams became the ctx stream's queue-mates in every rank).
  if (ok && comm_id) {
    if (comm_init_rank(ctx, comm_id, rank, world_size)) { const std::string m = ctx->err; delete ctx; return fail(nullptr, m); }
    ctx->comm_before_streams = true;
  }
  ok = ok && create_streams(ctx);
  ok = ok && hipMalloc(reinterpret_cast<void**>(&ctx->d_diag), sizeof(DiagCounters)) == hipSuccess && hipMemsetAsync(ctx->d_diag, 0, sizeof(DiagCounters), ctx->stream) == hipSuccess;
#ifdef GROOVE_HEARTBEAT
  if (ok && hipHostMalloc(reinterpret_cast<void**>(&ctx->hb), 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) { ctx->hb[0] = 0; ctx->hb[1] = 0; }
#endif
  if (!ok) {
    groove_shutdown(ctx);
    return fail(nullptr, "groove_init: hipSetDevice/hipStreamCreate failed");
  }
  *out = ctx;
  return 0;
}
int groove_init(int device_ordinal, groove_ctx** out) { return init_impl(device_ordinal, nullptr, 0, 1, out); }
int groove_init_comm(int device_ordinal, const uint8_t id[128], int rank, int world_size, groove_ctx** out) {
  if (!id) return fail(nullptr, "groove_init_comm: id is NULL");
  return init_impl(device_ordinal, id, rank, world_size, out);
}
void groove_shutdown(groove_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)ctx_join(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  while (!ctx->banks.empty()) groove_bank_destroy(ctx->banks.back());
  while (!ctx->fxs.empty()) groove_fx_destroy(ctx->fxs.back());
  groove_comm_destroy(ctx);
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  if (ctx->d_fpart) (void)hipFree(ctx->d_fpart);
  for (float* q : ctx->d_dpart) if (q) (void)hipFree(q);
  if (ctx->d_fseg) (void)hipFree(ctx->d_fseg);
  if (ctx->d_i16) (void)hipFree(ctx->d_i16);
  if (ctx->d_diag) (void)hipFree(ctx->d_diag);
  if (ctx->deferred.rows && ctx->deferred.owned_cap) (void)hipFree(const_cast<float*>(ctx->deferred.rows));
  for (auto& sp : ctx->spare_sums) (void)hipFree(sp.first);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  for (int i = 0; i < kSideStreams; ++i) {
    if (side_stream_owned(ctx, i)) { (void)hipStreamSynchronize(ctx->side_stream[i]); (void)hipStreamDestroy(ctx->side_stream[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->placeholder_stream) (void)hipStreamDestroy(ctx->placeholder_stream);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_ap_run) (void)hipEventDestroy(ctx->ev_ap_run);
  if (ctx->ev_ap_x) (void)hipEventDestroy(ctx->ev_ap_x);
  delete ctx;
}
const char* groove_last_error(groove_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }
int groove_set_stream(groove_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, "groove_set_stream: ctx is NULL");
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  if (ctx->own_stream && ctx->stream) GHIP(ctx, hipStreamDestroy(ctx->stream));
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return 0;
}
int groove_synchronize(groove_ctx* ctx) {
  if (!ctx) return fail(nullptr, "groove_synchronize: ctx is NULL");
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx, "groove_synchronize"));
  return 0;
}
int groove_set_sync_timeout_ms(groove_ctx* ctx, uint32_t ms) {
  if (!ctx) return fail(nullptr, "groove_set_sync_timeout_ms: ctx is NULL");
  ctx->sync_timeout_ms = ms;
  return 0;
}
uint32_t groove_sync_timeout_ms(groove_ctx* ctx) { return ctx ? ctx->sync_timeout_ms : 0; }
int groove_debug_spin(groove_ctx* ctx, int side_stream, uint32_t ms) {
  if (!ctx) return fail(nullptr, "groove_debug_spin: ctx is NULL");
  if (side_stream >= kSideStreams) return fail(ctx, "groove_debug_spin: no such side stream");
  int khz = 0;
  GHIP(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
  if (khz <= 0) khz = 100000; // 100 MHz
  hipStream_t st = side_stream < 0 ? ctx->stream : side_stream_of(ctx, side_stream);
  hipLaunchKernelGGL(debug_spin_kernel, dim3(1), dim3(64), 0, st, (uint64_t)ms * (uint64_t)khz);
  GHIP(ctx, hipGetLastError());
  if (side_stream >= 0) ctx->side_busy[side_stream] = true;
  return 0;
}
int groove_debug_info(groove_ctx* ctx, char* out, size_t cap) {
  if (!ctx || !out || cap == 0) return fail(ctx, "groove_debug_info: NULL argument");
  int distinct = 0;
  for (int k = 0; k < kSideStreams; ++k) {
    bool dup = !ctx->side_stream[k];
    for (int j = 0; j < k && !dup; ++j) dup = ctx->side_stream[j] == ctx->side_stream[k];
    distinct += dup ? 0 : 1;
  }
  // the segment guard's counter (diag.h): what the kernels have counted so far (groove_synchronize first for a final figure)
  DiagCounters dc{};
  GHIP(ctx, ctx_memcpy(ctx, &dc, ctx->d_diag, sizeof(dc), hipMemcpyDeviceToHost));
  std::string diag = "\"host_waits\": " + std::to_string(ctx->host_waits) + ", \"host_waits_blocked\": " + std::to_string(ctx->host_waits_blocked) + ", \"host_wait_ms\": " + std::to_string((double)ctx->host_wait_ns * 1e-6) +
                     ", \"zero_segments\": " + std::to_string(dc.zero_segments) + ", \"fast_table_misses\": " + std::to_string(dc.fast_table_misses) + ", \"fast_waves\": " + std::to_string(dc.fast_waves) + ", \"source_hash\": \"" GROOVE_SOURCE_HASH "\", \"mix_kernel\": " + (ctx->mix_kernel ? "true" : "false");
#ifdef GROOVE_DIAG_SHADOW_IN_MIN
  diag += ", \"diag_build\": \"GROOVE_DIAG_SHADOW_IN_MIN\", \"shadow_zero_lanes\": " + std::to_string(dc.shadow_zero_lanes) + ", \"shadow_zero_waves\": " + std::to_string(dc.shadow_zero_waves) + ", \"records\": [";
  for (uint32_t i = 0; i < std::min(dc.records, kDiagRecords); ++i) {
    const DiagRecord& r = dc.rec[i];
    diag += std::string(i ? ", " : "") + "{\"wg\": " + std::to_string(r.wg) + ", \"wave\": " + std::to_string(r.wave) + ", \"lane\": " + std::to_string(r.lane) + ", \"active\": " + std::to_string(r.active) +
            ", \"count\": " + std::to_string(r.count) + ", \"frame\": " + std::to_string(r.frame) + ", \"amp\": [" + std::to_string(r.amp_state) + ", " + std::to_string(r.amp_n) + ", " + std::to_string(r.amp_N) +
            "], \"fil\": [" + std::to_string(r.fil_state) + ", " + std::to_string(r.fil_n) + ", " + std::to_string(r.fil_N) + "]}";
  }
  diag += "]";
#endif
  std::snprintf(out, cap,
                "{\"layout\": \"%s\", \"streams_created\": %d, \"distinct_side_streams\": %d, \"kind_streams\": %d, \"bank_streams\": %d, "
                "\"placeholder_fifth\": %s, \"comm_before_streams\": %s, \"own_ctx_stream\": %s, \"sync_timeout_ms\": %u, %s}",
                ctx->safe_streams ? "safe (one priority, ctx + 3 shared side streams)" : "default (ctx high, 3 kind streams normal, placeholder, bank streams low)",
                ctx->streams_created, distinct, ctx->kind_streams, ctx->safe_streams ? 0 : ctx->bank_streams, ctx->placeholder_stream ? "true" : "false",
                ctx->comm_before_streams ? "true" : "false", ctx->own_stream ? "true" : "false", ctx->sync_timeout_ms, diag.c_str());
  return 0;
}
uint32_t groove_sample_rate(groove_ctx* ctx) { return ctx ? ctx->sr : 0; }
int groove_set_time_parallel_max_voices(groove_ctx* ctx, uint32_t max_voices) {
  if (!ctx) return fail(nullptr, "groove_set_time_parallel_max_voices: ctx is NULL");
  if (ctx_join(ctx)) return 1; // banks change kernels (and side streams) at their next render
  GHIP(ctx, ctx_wait(ctx));
  ctx->tp_max_voices = max_voices;
  return 0;
}
uint32_t groove_time_parallel_max_voices(groove_ctx* ctx) { return ctx ? ctx->tp_max_voices : 0; }
int groove_set_time_parallel_pair_min_voices(groove_ctx* ctx, uint32_t min_voices) {
  if (!ctx) return fail(nullptr, "groove_set_time_parallel_pair_min_voices: ctx is NULL");
  if (ctx_join(ctx)) return 1; // the partial-row count of a bank's renders changes with the form
  GHIP(ctx, ctx_wait(ctx));
  ctx->tp_vpw2_min_voices = min_voices;
  return 0;
}
uint32_t groove_time_parallel_pair_min_voices(groove_ctx* ctx) { return ctx ? ctx->tp_vpw2_min_voices : 0; }
int groove_set_look_ahead(groove_ctx* ctx, uint32_t bits) {
  if (!ctx) return fail(nullptr, "groove_set_look_ahead: ctx is NULL");
  ctx->look_ahead = bits & 7u;
  return 0;
}
uint32_t groove_look_ahead(groove_ctx* ctx) { return ctx ? ctx->look_ahead : 0; }
int groove_set_pipeline_min_waves(groove_ctx* ctx, uint32_t waves) {
  if (!ctx) return fail(nullptr, "groove_set_pipeline_min_waves: ctx is NULL");
  if (ctx_join(ctx)) return 1; // banks change kernels (and side streams) at their next render
  GHIP(ctx, ctx_wait(ctx));
  ctx->pipeline_min_waves = waves;
  return 0;
}
uint32_t groove_pipeline_min_waves(groove_ctx* ctx) { return ctx ? ctx->pipeline_min_waves : 0; }
int groove_set_fx_allpass_stream(groove_ctx* ctx, int on) {
  if (!ctx) return fail(nullptr, "groove_set_fx_allpass_stream: ctx is NULL");
  if (bus_flush(ctx) || ctx_join(ctx)) return 1; // nothing of the old arrangement is in flight when the next chain is submitted
  GHIP(ctx, ctx_wait(ctx));
  for (groove_fx* fx : ctx->fxs) fx->ap_busy = false;
  const int k_ap = kBaseKinds + 1;
  if (on && ctx->bank_streams > 1) // banks that render on that stream move to the others (everything is idle here; their next events go through the ctx stream)
    for (groove_bank* b : ctx->banks)
      if (b->stream_slot == k_ap) { b->stream_slot = kBaseKinds + (ctx->next_stream_slot++ % 2 ? 2 % ctx->bank_streams : 0); b->ctx_touched = true; }
  // the second of the low-priority bank streams: a lone bank renders on the first (measured, tools/ap_stream_ab.sh: 0.0489 -> 0.044
  // ms per block of config #3 there, 0.045 - 0.048 on a normal-priority kind stream, 0.063 - 0.070 on the first kind stream)
  ctx->fx_ap_stream = on ? k_ap : -1;
  return 0;
}
int groove_fx_allpass_stream(groove_ctx* ctx) { return ctx ? (ctx->fx_ap_stream >= 0 ? 1 : 0) : 0; }
int groove_set_split_max_waves(groove_ctx* ctx, uint32_t waves) {
  if (!ctx) return fail(nullptr, "groove_set_split_max_waves: ctx is NULL");
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  ctx->split_max_waves = waves;
  return 0;
}
uint32_t groove_split_max_waves(groove_ctx* ctx) { return ctx ? ctx->split_max_waves : 0; }
int groove_update_sample_rate(groove_ctx* ctx, uint32_t hz) {
  if (!ctx) return fail(nullptr, "groove_update_sample_rate: ctx is NULL");
  if (hz < 1000 || hz > 768000) return fail(ctx, "groove_update_sample_rate: unsupported rate");
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  ctx->sr = hz;
  for (groove_bank* b : ctx->banks)
    if (bank_derive_and_upload(b)) return 1;
  for (groove_fx* fx : ctx->fxs) {
    fx->last_side = -1; // joined above
    if (fx_setup_state(fx)) return 1;
    if (fx_upload_params(fx)) return 1;
  }
  return 0;
}
int groove_event_create(groove_ctx* ctx, void** out_event) {
  if (!ctx || !out_event) return fail(ctx, "groove_event_create: NULL argument");
  hipEvent_t ev;
  GHIP(ctx, hipEventCreate(&ev));
  *out_event = (void*)ev;
  return 0;
}
int groove_event_destroy(groove_ctx* ctx, void* event) {
  if (!ctx || !event) return fail(ctx, "groove_event_destroy: NULL argument");
  GHIP(ctx, hipEventDestroy((hipEvent_t)event));
  return 0;
}
int groove_event_record(groove_ctx* ctx, void* event) {
  if (!ctx || !event) return fail(ctx, "groove_event_record: NULL argument");
  if (bus_flush(ctx)) return 1; // the event marks the end of everything asked for so far
  GHIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
  return 0;
}
int groove_event_elapsed_ms(groove_ctx* ctx, void* start, void* stop, float* out_ms) {
  if (!ctx || !start || !stop || !out_ms) return fail(ctx, "groove_event_elapsed_ms: NULL argument");
  GHIP(ctx, wait_deadline(ctx, nullptr, (hipEvent_t)stop, "groove_event_elapsed_ms"));
  GHIP(ctx, hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

// ============================================================================ blocks
int groove_block_create(groove_ctx* ctx, uint32_t n, uint32_t frames_cap, groove_block** out) {
  if (!ctx || !out) return fail(ctx, "groove_block_create: NULL argument");
  if (n == 0 || frames_cap == 0) return fail(ctx, "groove_block_create: empty block");
  GHIP(ctx, hipSetDevice(ctx->device));
  groove_block* b = new groove_block();
  b->ctx = ctx; b->n = n; b->cap = frames_cap; b->d = nullptr;
  const size_t bytes = (size_t)2 * frames_cap * n * 4;
  hipError_t e = hipMalloc(&b->d, bytes);
  if (e != hipSuccess) { delete b; return fail(ctx, std::string("groove_block_create: hipMalloc: ") + hipGetErrorString(e)); }
  (void)hipMemsetAsync(b->d, 0, bytes, ctx->stream);
  *out = b;
  return 0;
}
int groove_block_destroy(groove_block* b) {
  if (!b) return 0;
  // lane sums of this block that still wait for the next all-pass launch: onto their bus first (the buffer dies with the block)
  if (b->ctx->deferred_ap.rows && (b->ctx->deferred_ap.rows == b->d_sums_ap[0] || b->ctx->deferred_ap.rows == b->d_sums_ap[1])) (void)bus_flush_ap(b->ctx);
  if (b->ready_mask) (void)ctx_join(b->ctx);
  (void)hipStreamSynchronize(b->ctx->stream);
  (void)hipFree(b->d);
  (void)hipFree(b->d_alt);
  (void)hipFree(b->d_sums);
  (void)hipFree(b->d_stage); (void)hipFree(b->d_sums_ap[0]); (void)hipFree(b->d_sums_ap[1]);
  if (b->ev_free) (void)hipEventDestroy(b->ev_free);
  for (hipEvent_t e : b->ev_ready) if (e) (void)hipEventDestroy(e);
  delete b;
  return 0;
}
float* groove_block_device_ptr(groove_block* b) {
  // The pointer escapes: whatever the caller's own kernels do to the block, the row sums the last render left no
  // longer describe it (groove_mix would put the pre-modification audio on the bus).
  if (!b) return nullptr;
  if (b->order && (block_acquire(b) || block_normalise(b))) return nullptr; // the caller's lane order, now that somebody looks
  b->sums_valid = false;
  return b->d;
}
int groove_block_mark_dirty(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_mark_dirty: block is NULL");
  b->sums_valid = false;
  return 0;
}
uint32_t groove_block_lanes(groove_block* b) { return b ? b->n : 0; }
uint32_t groove_block_frames_cap(groove_block* b) { return b ? b->cap : 0; }
int groove_block_upload(groove_block* b, const float* host, uint32_t frames) {
  if (!b || !host) return fail(nullptr, "groove_block_upload: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (frames > b->cap) return fail(ctx, "groove_block_upload: frames > capacity");
  if (block_acquire(b)) return 1;
  b->sums_valid = false;
  b->order.reset(); // the content is replaced
  const size_t per = (size_t)frames * b->n;
  for (int ch = 0; ch < 2; ++ch)
    GHIP(ctx, hipMemcpyAsync(b->d + (size_t)ch * b->cap * b->n, host + ch * per, per * 4, hipMemcpyHostToDevice, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}
int groove_block_download(groove_block* b, float* host, uint32_t frames) {
  if (!b || !host) return fail(nullptr, "groove_block_download: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (frames > b->cap) return fail(ctx, "groove_block_download: frames > capacity");
  if (block_acquire(b) || block_normalise(b)) return 1;
  GHIP(ctx, ctx_wait(ctx, "copy on the ctx stream"));
  const size_t per = (size_t)frames * b->n;
  for (int ch = 0; ch < 2; ++ch)
    GHIP(ctx, hipMemcpyAsync(host + ch * per, b->d + (size_t)ch * b->cap * b->n, per * 4, hipMemcpyDeviceToHost, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}

// ============================================================================ instruments
// Round-robin side-stream assignment of single-kernel banks; the all-pass stream (groove_set_fx_allpass_stream), while there is
// one, is left to the all-passes: a bank that rendered on it would queue its renders behind them.
static int next_bank_slot(groove_ctx* ctx) {
  int slot = kBaseKinds + ctx->next_stream_slot++ % ctx->bank_streams;
  if (slot == ctx->fx_ap_stream && ctx->bank_streams > 1) slot = kBaseKinds + ctx->next_stream_slot++ % ctx->bank_streams;
  return slot;
}
static int bank_finish_create(groove_bank* b, groove_bank** out) {
  b->stream_slot = next_bank_slot(b->ctx);
  if (bank_alloc(b) || bank_derive_and_upload(b)) { groove_bank_destroy(b); return 1; }
  b->ctx->banks.push_back(b);
  *out = b;
  return 0;
}
int groove_welsh_create(groove_ctx* ctx, const groove_welsh_params* p, uint32_t n, groove_bank** out) {
  if (!ctx || !p || !out) return fail(ctx, "groove_welsh_create: NULL argument");
  if (n == 0) return fail(ctx, "groove_welsh_create: n == 0");
  for (uint32_t v = 0; v < n; ++v) // every kernel form must see the same, canonical enumerators
    if (p[v].oscillator_1.waveform >= GROOVE_WAVEFORM_COUNT || p[v].oscillator_2.waveform >= GROOVE_WAVEFORM_COUNT ||
        p[v].lfo_waveform >= GROOVE_WAVEFORM_COUNT || p[v].lfo_routing >= GROOVE_LFO_ROUTING_COUNT)
      return fail(ctx, "groove_welsh_create: waveform or LFO routing enumerator out of range (voice " + std::to_string(v) + ")");
  groove_bank* b = new groove_bank();
  b->ctx = ctx; b->kind = BANK_WELSH; b->n = n;
  b->pw = sizeof(WelshParams) / 4; b->sw = sizeof(WelshState) / 4;
  b->welsh.assign(p, p + n);
  if (hipMalloc(&b->d_cold, (size_t)4 * n * 8) != hipSuccess) { delete b; return fail(ctx, "groove_welsh_create: hipMalloc failed"); }
  return bank_finish_create(b, out);
}
int groove_fm_create(groove_ctx* ctx, const groove_fm_params* p, uint32_t n, groove_bank** out) {
  if (!ctx || !p || !out) return fail(ctx, "groove_fm_create: NULL argument");
  if (n == 0) return fail(ctx, "groove_fm_create: n == 0");
  groove_bank* b = new groove_bank();
  b->ctx = ctx; b->kind = BANK_FM; b->n = n;
  b->pw = sizeof(FmParams) / 4; b->sw = sizeof(FmState) / 4;
  b->fm.assign(p, p + n);
  if (hipMalloc(&b->d_cold, (size_t)n * 8) != hipSuccess) { delete b; return fail(ctx, "groove_fm_create: hipMalloc failed"); }
  return bank_finish_create(b, out);
}
int groove_sampler_create(groove_ctx* ctx, const float* bank_pcm, uint64_t bank_frames,
                          const groove_sample_desc* descs, uint32_t n_samples,
                          const groove_sampler_params* p, uint32_t n, groove_bank** out) {
  if (!ctx || !bank_pcm || !descs || !p || !out) return fail(ctx, "groove_sampler_create: NULL argument");
  if (n == 0 || n_samples == 0 || bank_frames == 0) return fail(ctx, "groove_sampler_create: empty bank");
  if (bank_frames >= (1ull << 32)) return fail(ctx, "groove_sampler_create: bank too large");
  for (uint32_t i = 0; i < n_samples; ++i) {
    if (descs[i].offset > bank_frames || descs[i].length > bank_frames - descs[i].offset)
      return fail(ctx, "groove_sampler_create: sample descriptor exceeds bank");
    if (descs[i].length == 0) return fail(ctx, "groove_sampler_create: empty sample"); // the fetch clamps to length - 1
    if (descs[i].length >= (1u << 20)) return fail(ctx, "groove_sampler_create: sample longer than 2^20 frames");
  }
  for (uint32_t i = 0; i < n; ++i)
    if (p[i].sample_index >= n_samples) return fail(ctx, "groove_sampler_create: sample_index out of range");
  groove_bank* b = new groove_bank();
  b->ctx = ctx; b->kind = BANK_SAMPLER; b->n = n;
  b->pw = sizeof(SamplerParams) / 4; b->sw = sizeof(SamplerState) / 4;
  b->sampler.assign(p, p + n);
  b->descs.assign(descs, descs + n_samples);
  if (hipMalloc(&b->d_pcm, bank_frames * 4) != hipSuccess) { delete b; return fail(ctx, "groove_sampler_create: hipMalloc failed"); }
  if (ctx_memcpy(ctx, b->d_pcm, bank_pcm, bank_frames * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(b->d_pcm); delete b; return fail(ctx, "groove_sampler_create: upload failed"); }
  return bank_finish_create(b, out);
}
int groove_bank_destroy(groove_bank* b) {
  if (!b) return 0;
  groove_ctx* ctx = b->ctx;
  (void)paced_reduce(b, false);
  (void)ctx_join(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  for (int slot = 0; slot < 2; ++slot) {
    (void)hipFree(b->d_pipe_part[slot]); (void)hipFree(b->d_pipe_seg[slot]);
    if (b->ev_reduce_done[slot]) (void)hipEventDestroy(b->ev_reduce_done[slot]);
    for (int k = 0; k < kBaseKinds + 4; ++k) if (b->ev_render_done[k][slot]) (void)hipEventDestroy(b->ev_render_done[k][slot]);
  }
  auto it = std::find(ctx->banks.begin(), ctx->banks.end(), b);
  if (it != ctx->banks.end()) ctx->banks.erase(it);
  if (b->scratch) groove_block_destroy(b->scratch);
  for (int k = 0; k < 2; ++k) { if (b->h_ev[k]) (void)hipHostFree(b->h_ev[k]); if (b->ev_staged[k]) (void)hipEventDestroy(b->ev_staged[k]); }
  (void)hipFree(b->d_params); (void)hipFree(b->d_state); (void)hipFree(b->d_cold); (void)hipFree(b->d_pcm); (void)hipFree(b->d_ev[0]); (void)hipFree(b->d_ev[1]); (void)hipFree(b->d_waves); (void)hipFree(b->d_wg_list); (void)hipFree(b->d_wg_cls); (void)hipFree(b->d_wg_base); (void)hipFree(b->d_wg_f32);
  delete b;
  return 0;
}
uint32_t groove_bank_voices(groove_bank* b) { return b ? b->n : 0; }
int groove_bank_note_events(groove_bank* b, const groove_note_event* ev, uint32_t n_ev) {
  if (!b) return fail(nullptr, "groove_bank_note_events: bank is NULL");
  if (n_ev && !ev) return fail(b->ctx, "groove_bank_note_events: ev is NULL");
  for (uint32_t i = 0; i < n_ev; ++i)
    if (ev[i].voice != GROOVE_ALL_VOICES && ev[i].voice >= b->n)
      return fail(b->ctx, "groove_bank_note_events: voice index out of range");
  b->pending.insert(b->pending.end(), ev, ev + n_ev);
  return 0;
}
int groove_bank_set_param(groove_bank* b, uint32_t voice, uint32_t control_index, double value01) {
  if (!b) return fail(nullptr, "groove_bank_set_param: bank is NULL");
  groove_ctx* ctx = b->ctx;
  if (voice != GROOVE_ALL_VOICES && voice >= b->n) return fail(ctx, "groove_bank_set_param: voice out of range");
  if (b->kind != BANK_WELSH) return fail(ctx, "groove_bank_set_param: only Welsh banks expose controls");
  const double v01 = value01 < 0.0 ? 0.0 : (value01 > 1.0 ? 1.0 : value01);
  const uint32_t lo = voice == GROOVE_ALL_VOICES ? 0 : voice, hi = voice == GROOVE_ALL_VOICES ? b->n : voice + 1;
  for (uint32_t v = lo; v < hi; ++v) {
    groove_welsh_params& p = b->welsh[v];
    switch (control_index) {
      case GROOVE_CTL_WELSH_DCA_GAIN: p.dca_gain = (float)v01; break;
      case GROOVE_CTL_WELSH_DCA_PAN: p.dca_pan = (float)(v01 * 2.0 - 1.0); break; // ControlValue 0..1 → BipolarNormal
      case GROOVE_CTL_WELSH_CUTOFF: p.filter_cutoff_hz = (float)percent_to_frequency_h(v01); break;
      default: return fail(ctx, "groove_bank_set_param: unknown control index");
    }
  }
  // control-plane path: re-derive and re-upload the parameter tables (state is untouched)
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  return welsh_upload_params(b, false); // the state stays where it is: keep the lane order
}
// The block's row-sum buffer for a render of `rows` partial rows (reallocated when it grows).
static float* block_sums(groove_block* blk, uint32_t rows, uint32_t frames) {
  const size_t need = (size_t)rows * 2 * frames;
  if (blk->sums_cap < need) {
    if (blk->d_sums) { (void)hipStreamSynchronize(blk->ctx->stream); (void)hipFree(blk->d_sums); blk->d_sums = nullptr; blk->sums_cap = 0; }
    const size_t cap = std::max(need, (size_t)rows * 2 * std::min<uint32_t>(blk->cap, 4096));
    if (hipMalloc(&blk->d_sums, cap * 4) != hipSuccess) { fail(blk->ctx, "block row sums: hipMalloc failed"); return nullptr; }
    blk->sums_cap = cap;
  }
  blk->sums_valid = false;
  blk->sums_on_ap = false;
  return blk->d_sums;
}
// Small Welsh banks and blocks of up to 256 frames: one wavefront per voice, lanes = time (welsh_tp.h).
static bool use_tp(const groove_bank* b, uint32_t frames) {
  if (frames > kTpMaxFrames || b->ctx->tp_max_voices == 0) return false;
  if (b->kind == BANK_WELSH) {
    // two voices per wavefront move the crossover with the role-split kernel up by three eighths (measured, round 3: 18,432
    // voices 0.069 ms per block against 0.083, 22,528 0.080 against 0.084, 24,576 0.084 against 0.083)
    const uint32_t tmax = b->ctx->tp_max_voices;
    const bool pairs = b->tp_pairs && b->ctx->tp_vpw2_min_voices && b->n >= b->ctx->tp_vpw2_min_voices;
    return b->n <= (pairs ? tmax + (uint32_t)std::min<uint64_t>((uint64_t)tmax * 3 / 8, 0x40000000u) : tmax);
  }
  if (b->kind == BANK_FM) return b->n <= b->ctx->fm_tp_max_voices; // no filter scan: far cheaper per voice than a Welsh voice
  if (b->kind == BANK_SAMPLER) return b->n <= kSamplerTpMaxVoices; // a pure gather
  return false;
}
static uint32_t tp_vpw(const groove_bank* b) { // voices per wavefront of the time-parallel Welsh / FM kernels
  if (b->kind == BANK_FM) { // (parameters per lane: no condition on the patches) four once the one-voice form is past ~3 wavefronts per SIMD
    const uint32_t m = b->ctx->fm_tp_vpw4_min_voices;
    return (m && b->n >= m) ? 4u : 1u;
  }
  return (b->kind == BANK_WELSH && b->tp_pairs && b->ctx->tp_vpw2_min_voices && b->n >= b->ctx->tp_vpw2_min_voices) ? 2u : 1u;
}
static void launch_tp(groove_bank* b, uint32_t frames, bool fused, size_t chs, float* out, float* rows, hipStream_t st, const groove_fx* head = nullptr,
                      hipEvent_t done = nullptr /* completes with the kernel (bound to the dispatch: kernels.h launch_bound) */, const TpPrev* prev = nullptr,
                      uint32_t sampler_vpw = 0 /* sampler only: voices per wavefront (0: the default rule) */) {
  groove_ctx* ctx = b->ctx;
  TpArgs a{b->d_params, b->d_state, out, rows, chs, render_consts_of(ctx), b->n, frames};
  if (prev) a.prev = *prev;
  if (head) { a.bq_coef = head->d_coef; a.bq_st = head->d_st; a.bq_wet = head->d_wet; } // Welsh, block-writing form: the BiQuad head fused (welsh_tp.h)
  if (b->kind == BANK_FM) { a.vpw = tp_vpw(b); launch_fm_tp(a, st, fused, done); }
  else if (b->kind == BANK_SAMPLER) { a.vpw = sampler_vpw; launch_sampler_tp(a, b->d_pcm, b->inline_ev, st, fused, done); b->inline_ev.n = 0; }
  else { a.full_coef = b->tp_full_coef; a.vpw = tp_vpw(b); launch_welsh_tp(a, st, fused, done); }
}
// rows of partial[][2][frames] a bank's fused render writes
static uint32_t fused_rows(const groove_bank* b, uint32_t frames) {
  if (use_tp(b, frames)) return b->kind == BANK_SAMPLER ? sampler_tp_workgroups(b->n) : b->kind == BANK_WELSH ? welsh_tp_grid(b->n, tp_vpw(b)) : welsh_tp_workgroups(b->n, tp_vpw(b));
  return (b->kind == BANK_WELSH && b->n_vwaves) ? (b->n_vwaves + kWaves - 1) / kWaves : blocks_for(b->n);
}
// The one argument block of the wave-uniform Welsh kernels: workgroups [wg_off, wg_off + n_wgs) of the bank's kind-sorted list.
static UniformArgs uniform_args(const groove_bank* b, float* out, float* rows, uint32_t wg_off, size_t chs, uint32_t frames, uint32_t n_wgs) {
  UniformArgs a{b->d_waves, b->d_state, out, rows, b->d_wg_list + wg_off, b->d_wg_cls + wg_off, b->d_wg_f32 + wg_off, chs, render_consts_of(b->ctx), b->n_vwaves, b->n, frames, n_wgs};
  a.diag = b->ctx->d_diag;
#ifdef GROOVE_HEARTBEAT
  a.heartbeat = b->ctx->hb;
#endif
  return a;
}
// A Welsh bank below the per-kind pipeline's threshold: ONE launch for all its workgroups — role-split (welsh_split.h) when the
// bank is mid-size, for the workgroups of the four class-specialised base kinds (the workgroup list is sorted by kind: they
// come first); the rest, or everything, through the all-kinds kernel.
static bool use_split(const groove_bank* b, uint32_t frames) {
  const groove_ctx* ctx = b->ctx;
  if (ctx->split_max_waves == 0) return false; // "never" switches both forms off
  return b->kind == BANK_WELSH && b->n_vwaves && !use_tp(b, frames) && b->n_vwaves <= std::max(ctx->split_max_waves, ctx->split2_max_waves) && frames >= 2 * kSplitChunk;
}
static int split_roles_of(const groove_bank* b) { return b->n_vwaves <= b->ctx->split_max_waves ? b->ctx->split_roles : 2; }
static void launch_welsh_kind(int k, const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done = nullptr);
static void launch_small_uniform(groove_bank* b, const UniformArgs& a, hipStream_t st, bool fused, uint32_t frames, hipEvent_t done = nullptr /* bound to the last launch */) {
  // the workgroup list is sorted by kind: the four class-specialised base kinds first, then the two exact-f64 ones
  uint32_t n_spec = 0, n_f64[2] = {0, 0};
  for (int k = 0; k < 4 * kClassCombos; ++k) n_spec += b->wgs_of_kind[k];
  for (int k = 0; k < kClassCombos; ++k) { n_f64[0] += b->wgs_of_kind[4 * kClassCombos + k]; n_f64[1] += b->wgs_of_kind[5 * kClassCombos + k]; }
  if (n_spec) {
    UniformArgs s = a;
    s.n_wgs = n_spec;
    const hipEvent_t d = (n_f64[0] || n_f64[1]) ? nullptr : done;
    if (use_split(b, frames)) {
      const int roles = split_roles_of(b);
      if (roles == 4) launch_welsh_split4(s, b->d_wg_base, st, fused, d);
      else if (roles == 3) launch_welsh_split(s, b->d_wg_base, st, fused, d);
      else launch_welsh_split2(s, b->d_wg_base, st, fused, d);
    }
    else if (fused) launch_welsh_uniform_any(s, b->d_wg_base, st, d);
    else launch_welsh_uniform_any_unfused(s, b->d_wg_base, st, d);
  }
  // exact-f64 LFO kinds (rare; their bodies need 133 VGPRs): the per-kind kernels, budgeted for them, behind it on the same stream
  uint32_t at = n_spec;
  for (int j = 0; j < 2; ++j) {
    if (!n_f64[j]) continue;
    UniformArgs r = a;
    r.prev = TpPrev{}; // (the launch above carried the previous block's rows)
    r.wg_list = a.wg_list + at; r.wg_cls = a.wg_cls + at; r.n_wgs = n_f64[j];
    launch_welsh_kind(4 + j, r, st, fused, (j == 1 || !n_f64[1]) ? done : nullptr);
    at += n_f64[j];
  }
}
// One base kind's uniform Welsh kernel (kernels.h, "Workgroup KINDS") on stream `st`.
static void launch_welsh_kind(int k, const UniformArgs& a, hipStream_t st, bool fused, hipEvent_t done) {
  // every base kind has class-specialised bodies (round 6: the two exact-f64 kinds too), one translation unit each (csrc/welsh_class.hip)
  switch (k) {
    case 0: launch_welsh_uniform_specialised_0(a, st, fused, done); break;
    case 1: launch_welsh_uniform_specialised_1(a, st, fused, done); break;
    case 2: launch_welsh_uniform_specialised_2(a, st, fused, done); break;
    case 3: launch_welsh_uniform_specialised_3(a, st, fused, done); break;
    case 4: launch_welsh_uniform_specialised_4(a, st, fused, done); break;
    default: launch_welsh_uniform_specialised_5(a, st, fused, done); break;
  }
}
// fused: `rows` = partial rows only.  Otherwise `out` = the planar block and `rows` = its row sums (kernels.h run_frames).
static int launch_render(groove_bank* b, uint32_t frames, bool fused, size_t chs, float* out, float* rows) {
  groove_ctx* ctx = b->ctx;
  b->ctx_touched = true;
  const dim3 grid(blocks_for(b->n)), blk(kThreads);
  if (use_tp(b, frames)) {
    launch_tp(b, frames, fused, chs, out, rows, ctx->stream);
  } else if (b->kind == BANK_WELSH) {
    const RenderConsts rc = render_consts_of(ctx);
    if (b->n_vwaves == 0) { // interleaved bank: per-lane kernel over the physical lanes
      if (fused) hipLaunchKernelGGL(welsh_render_kernel<true>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows, rc);
      else hipLaunchKernelGGL(welsh_render_kernel<false>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows, rc);
    } else {
      // One kernel per base kind present, all running concurrently: the most expensive kind goes
      // out first on the ctx stream (list scheduling, longest first), the others on side streams
      // forked from it, and the ctx stream joins them before the bus reduction.
      if (b->n_vwaves < ctx->pipeline_min_waves) { // small bank: all base kinds in one launch (kernels.h)
        const uint32_t wgs = (b->n_vwaves + kWaves - 1) / kWaves;
        UniformArgs a = uniform_args(b, out, rows, 0, chs, frames, wgs);
        launch_small_uniform(b, a, ctx->stream, fused, frames);
        GHIP(ctx, hipGetLastError());
        return 0;
      }
      uint32_t count[kBaseKinds] = {}, offset[kBaseKinds] = {};
      {
        uint32_t at = 0;
        for (int base = 0; base < kBaseKinds; ++base) {
          offset[base] = at;
          for (int c = 0; c < kClassCombos; ++c) count[base] += b->wgs_of_kind[base * kClassCombos + c];
          at += count[base];
        }
      }
      int present = 0;
      for (uint32_t c : count) present += c ? 1 : 0;
      if (present > 1) GHIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      int side = 0;
      bool first_kind = true;
      for (int k = kBaseKinds - 1; k >= 0; --k) {
        if (!count[k]) continue;
        hipStream_t st = ctx->stream;
        if (!first_kind) {
          st = side_stream_of(ctx, side);
          GHIP(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0));
        }
        UniformArgs a = uniform_args(b, out, rows, offset[k], chs, frames, count[k]);
        launch_welsh_kind(k, a, st, fused);
        if (!first_kind) {
          GHIP(ctx, hipEventRecord(ctx->ev_join[side], st));
          GHIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[side], 0));
          ++side;
        }
        first_kind = false;
        ctx->need_fork = true; // ctx-stream work the pipelined path's side streams must see
      }
    }
  } else if (b->kind == BANK_FM) {
    if (fused) hipLaunchKernelGGL(fm_render_kernel<true>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows);
    else hipLaunchKernelGGL(fm_render_kernel<false>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows);
  } else {
    if (fused) hipLaunchKernelGGL(sampler_render_kernel<true>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows, b->d_pcm);
    else hipLaunchKernelGGL(sampler_render_kernel<false>, grid, blk, 0, ctx->stream, b->d_params, b->d_state, b->n, frames, chs, out, rows, b->d_pcm);
  }
  GHIP(ctx, hipGetLastError());
  return 0;
}
int groove_bank_render(groove_bank* b, uint32_t frames, groove_block* out) {
  if (!b || !out) return fail(nullptr, "groove_bank_render: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (out->n != b->n) return fail(ctx, "groove_bank_render: block lanes != bank voices");
  if (frames > out->cap) return fail(ctx, "groove_bank_render: frames > block capacity");
  if (frames == 0) return 0;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (flush_events(b, use_tp(b, frames))) return 1;
  if (ctx_join(ctx)) return 1; // the bank's state may still be in flight on the side streams (pipelined fused renders)
  b->side_mode = 0;
  out->ready_mask = 0; // joined above
  const uint32_t rows_n = fused_rows(b, frames);
  float* rows = block_sums(out, rows_n, frames);
  if (!rows) return 1;
  auto rendered = [&]() { out->sum_rows = rows_n; out->sum_frames = frames; out->sums_valid = true; return 0; };
  // a regrouped bank renders in ITS lane order (coalesced rows) into the block's second buffer; the caller's order is
  // produced when — and if — something asks for it (groove_block: lazy lane order)
  float* dst = nullptr;
  if (block_render_target(out, b->order, frames, &dst)) return 1;
  return launch_render(b, frames, false, (size_t)out->cap * out->n, dst, rows) || rendered();
}
// groove_bank_render on the side streams: the render kernels start once everything submitted to the
// ctx stream so far has finished (that covers the previous users of `out` and of the bank), and run
// beside whatever the ctx stream is given next; the ctx-stream operations that take `out` wait for
// them (block_acquire).  A host that keeps two blocks per instrument and submits the render of block
// b+1 before the effect chain of block b overlaps the two (bench.py, workload chain-4096: the Welsh
// render of a 4,096-voice bank is one wavefront's serial walk, 0.2 ms whatever else runs).
static bool fx_is_biquad12(uint32_t kind) {
  switch (kind) {
    case GROOVE_FX_BIQUAD_LP12: case GROOVE_FX_BIQUAD_HP12: case GROOVE_FX_BIQUAD_BP12: case GROOVE_FX_BIQUAD_BS12:
    case GROOVE_FX_BIQUAD_AP12: case GROOVE_FX_BIQUAD_PEAK12: case GROOVE_FX_BIQUAD_LSHELF12: case GROOVE_FX_BIQUAD_HSHELF12: return true;
    default: return false;
  }
}
// `head` (may be null): a 12 dB BiQuad effect bank to be applied to the block inside the render kernel — only honoured
// (*head_fused = true) when the bank renders time-parallel and its lanes are in the caller's order.
// `chained`: an effect chain follows on this block (groove_bank_render_chain_async), so the lane sums the render would leave are
// never read — a time-parallel Welsh render then does not write them (2 MB per block for config #3's 4,096 voices).
static int render_async_impl(groove_bank* b, uint32_t frames, groove_block* out, groove_fx* head, bool* head_fused, bool chained = false);
int groove_bank_render_async(groove_bank* b, uint32_t frames, groove_block* out) { return render_async_impl(b, frames, out, nullptr, nullptr); }
static int render_async_impl(groove_bank* b, uint32_t frames, groove_block* out, groove_fx* head, bool* head_fused, bool chained) {
  if (head_fused) *head_fused = false;
  if (!b || !out) return fail(nullptr, "groove_bank_render_async: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (out->n != b->n) return fail(ctx, "groove_bank_render_async: block lanes != bank voices");
  if (frames > out->cap) return fail(ctx, "groove_bank_render_async: frames > block capacity");
  if (frames == 0) return 0;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (flush_events(b, use_tp(b, frames))) return 1;
  const bool was_released = out->released;
  if (block_acquire(out)) return 1; // an earlier asynchronous render into the same block comes first
  const bool tp = use_tp(b, frames);
  const bool small_uniform = !tp && b->kind == BANK_WELSH && b->n_vwaves && b->n_vwaves < ctx->pipeline_min_waves;
  const bool uniform = !tp && b->kind == BANK_WELSH && b->n_vwaves && !small_uniform; // one kernel per base kind
  if (bank_side_mode(b, uniform ? 1 : 2)) return 1;
  if (!out->ev_free) {
    GHIP(ctx, hipEventCreateWithFlags(&out->ev_free, kSyncEventFlags));
    for (int k = 0; k < kSideStreams; ++k) GHIP(ctx, hipEventCreateWithFlags(&out->ev_ready[k], kSyncEventFlags));
  }
  const size_t chs = (size_t)out->cap * out->n;
  const uint32_t rows_n = fused_rows(b, frames);
  const bool no_sums = chained && tp && b->kind == BANK_WELSH;
  float* rows = no_sums ? nullptr : block_sums(out, rows_n, frames);
  if (!rows && !no_sums) return 1;
  if (no_sums) out->sums_valid = false;
  float* dst = nullptr; // a regrouped bank: the block's library-order buffer (lazy lane order)
  if (block_render_target(out, b->order, frames, &dst)) return 1;
  // What the render has to wait for on the ctx stream: the block's consumers, and the bank's state if
  // the ctx stream has worked on it.  A block the host has released (groove_block_release) carries the
  // event of that moment, typically long past, so the render follows the previous one on its stream
  // without a cross-queue wait; otherwise everything submitted so far is waited for.
  const bool free_recorded_now = !(was_released && !b->ctx_touched);
  if (free_recorded_now) GHIP(ctx, hipEventRecord(out->ev_free, ctx->stream));
  b->ctx_touched = false;
  const dim3 blk(kThreads);
  uint32_t used = 0;
  auto begin = [&](int k) -> hipStream_t {
    hipStream_t st = side_stream_of(ctx, k);
    (void)hipStreamWaitEvent(st, out->ev_free, 0);
    // An ev_free recorded in this call is later than any ev_fork; the event of an earlier release may
    // predate one, and then this (shared) side stream still owes the wait for the fork.
    if (ctx->fork_pending[k] && !free_recorded_now) (void)hipStreamWaitEvent(st, ctx->ev_fork, 0);
    ctx->fork_pending[k] = false;
    return st;
  };
  bool ready_bound = false; // the launch itself carries out->ev_ready[k]
  auto end = [&](int k) {
    if (!ready_bound) (void)hipEventRecord(out->ev_ready[k], side_stream_of(ctx, k));
    used |= 1u << k;
    ctx->side_busy[k] = true;
  };
  if (uniform) {
    uint32_t at = 0;
    uint32_t count[kBaseKinds] = {}, offset[kBaseKinds] = {};
    for (int base = 0; base < kBaseKinds; ++base) {
      offset[base] = at;
      for (int c = 0; c < kClassCombos; ++c) count[base] += b->wgs_of_kind[base * kClassCombos + c];
      at += count[base];
    }
    const bool mix = ctx->mix_kernel; // (the MIX kernel, block-writing form: kernels.h; the exact-f64 kinds keep their per-kind kernels)
    for (int k = kBaseKinds - 1; k >= 0; --k) { // most expensive kind first
      if (mix && k < 4) continue;
      if (!count[k]) continue;
      hipStream_t st = begin(k);
      UniformArgs a = uniform_args(b, dst, rows, offset[k], chs, frames, count[k]);
      launch_welsh_kind(k, a, st, false);
      end(k);
    }
    for (int sec = 2; mix && sec >= 0; --sec) {
      if (!b->mix_cnt[sec]) continue;
      hipStream_t st = begin(sec);
      UniformArgs a = uniform_args(b, dst, rows, 0, chs, frames, b->mix_cnt[sec]);
      const size_t o = b->wg_list_cap + b->mix_off[sec];
      a.wg_list = b->d_wg_list + o; a.wg_cls = b->d_wg_cls + o; a.wg_f32 = b->d_wg_f32 + o;
      launch_welsh_uniform_mix_unfused(a, b->d_wg_base + o, st);
      end(sec);
    }
  } else {
    const int k = b->stream_slot;
    hipStream_t st = begin(k);
    const dim3 grid(blocks_for(b->n));
    if (tp) {
      const bool fuse = head && b->kind == BANK_WELSH && !b->order && fx_is_biquad12(head->kind) && head->n == b->n && head->ctx == ctx;
      if (fuse) { // the effect's state moves to this side stream (fx_acquire_ctx)
        if (head->last_side != k) {
          if (!head->ev_done) GHIP(ctx, hipEventCreateWithFlags(&head->ev_done, kSyncEventFlags));
          if (head->last_side < 0) GHIP(ctx, hipEventRecord(head->ev_done, ctx->stream));
          else if (fx_done_event(head)) return 1;
          GHIP(ctx, hipStreamWaitEvent(st, head->ev_done, 0));
        }
      }
      // the block's "ready" event completes with the render kernel itself (no record behind it)
      const bool bind = ctx->bind_events && b->kind == BANK_WELSH;
      launch_tp(b, frames, false, chs, dst, rows, st, fuse ? head : nullptr, bind ? out->ev_ready[k] : nullptr);
      if (fuse) { // the effect's last use ends with that kernel too: its own event is recorded when somebody asks (fx_done_event)
        if (bind) head->done_recorded = false;
        else { GHIP(ctx, hipEventRecord(head->ev_done, st)); head->done_recorded = true; }
        head->last_side = k;
        *head_fused = true;
      }
      ready_bound = bind;
    } else if (small_uniform) { // all base kinds in one launch
      UniformArgs a = uniform_args(b, dst, rows, 0, chs, frames, fused_rows(b, frames));
      launch_small_uniform(b, a, st, false, frames);
    } else if (b->kind == BANK_WELSH) {
      const RenderConsts rc = render_consts_of(ctx);
      hipLaunchKernelGGL(welsh_render_kernel<false>, grid, blk, 0, st, b->d_params, b->d_state, b->n, frames, chs, dst, rows, rc);
    } else if (b->kind == BANK_FM) {
      hipLaunchKernelGGL(fm_render_kernel<false>, grid, blk, 0, st, b->d_params, b->d_state, b->n, frames, chs, dst, rows);
    } else {
      hipLaunchKernelGGL(sampler_render_kernel<false>, grid, blk, 0, st, b->d_params, b->d_state, b->n, frames, chs, dst, rows, b->d_pcm);
    }
    end(k);
  }
  out->ready_mask = used;
  if (!no_sums) { out->sum_rows = rows_n; out->sum_frames = frames; out->sums_valid = true; }
  GHIP(ctx, hipGetLastError());
  return 0;
}
int groove_block_release(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_release: block is NULL");
  groove_ctx* ctx = b->ctx;
  const bool marked = b->free_marked; // nothing has taken the block since the mix that bound ev_free to its last kernel
  if (ctx->fx_ap_stream >= 0 && b->ev_free && b->ready_mask == (1u << ctx->fx_ap_stream)) {
    // the block's last kernel is on the all-pass stream and nothing of the ctx stream's is behind it: the release is recorded there
    // (the ctx stream does not wait for the block); the block stays "pending" for the ctx stream until somebody has waited
    GHIP(ctx, hipEventRecord(b->ev_free, ap_stream(ctx)));
    b->released = true; b->free_marked = false;
    return 0;
  }
  if (block_acquire(b)) return 1;
  if (!b->ev_free) {
    GHIP(ctx, hipEventCreateWithFlags(&b->ev_free, kSyncEventFlags));
    for (int k = 0; k < kSideStreams; ++k) GHIP(ctx, hipEventCreateWithFlags(&b->ev_ready[k], kSyncEventFlags));
  }
  if (!marked) GHIP(ctx, hipEventRecord(b->ev_free, ctx->stream));
  b->released = true;
  return 0;
}
int groove_block_acquire(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_acquire: block is NULL");
  b->sums_valid = false; // the caller is about to run its own kernels on the block (include/groove_hip.h)
  return block_acquire(b);
}
// Fused render + mix of a wave-uniform Welsh bank, pipelined over blocks.  Every base kind has its
// own stream that carries that kind's kernels block after block (a workgroup's state only depends
// on the same workgroup's previous block); the ctx stream carries the bus reductions, each waiting
// for its block's kernels.  Nothing makes block b+1's kernels wait for block b's reduction, so the
// thinly occupied tail of one block (the last, partly filled round of waves) overlaps the head of
// the next: ≈ 15 % at 1,000,000 voices, more for smaller banks.  Two slots of partial rows; a kind's
// render of block b+2 waits for the reduction of block b.
static int render_mix_pipelined(groove_bank* b, uint32_t frames, float* bus_dev, int accumulate, bool paced = false) {
  groove_ctx* ctx = b->ctx;
  const bool tp = use_tp(b, frames);
  const bool small_uniform = !tp && b->kind == BANK_WELSH && b->n_vwaves && b->n_vwaves < ctx->pipeline_min_waves && ctx->pipeline_min_waves > 1;
  const bool uniform = !tp && b->kind == BANK_WELSH && b->n_vwaves && !small_uniform; // one kernel per base kind
  const uint32_t rows = fused_rows(b, frames);
  const uint32_t cols = 2 * frames, rows_per_seg = 64, segs = (rows + rows_per_seg - 1) / rows_per_seg;
  if (bank_side_mode(b, uniform ? 1 : 2)) return 1;
  const int slot = b->pipe_slot;
  b->pipe_slot ^= 1;
  if (b->paced.active && (!paced || b->paced.slot == slot)) // (an unpaced call, or the slot's rows still owed)
    if (const int rc = paced_reduce(b, false)) return rc;
  if (b->pipe_part_cap[slot] < (size_t)rows * cols || b->pipe_seg_cap[slot] < (size_t)segs * cols) {
    if (b->paced.active) if (const int rc = paced_reduce(b, false)) return rc;
    if (ctx_join(ctx)) return 1;
    GHIP(ctx, ctx_wait(ctx));
    if (b->d_pipe_part[slot]) GHIP(ctx, hipFree(b->d_pipe_part[slot]));
    GHIP(ctx, hipMalloc(&b->d_pipe_part[slot], (size_t)rows * cols * 4));
    b->pipe_part_cap[slot] = (size_t)rows * cols;
    if (ensure_seg_buffer(ctx, &b->d_pipe_seg[slot], &b->pipe_seg_cap[slot], (size_t)segs * cols)) return 1;
    b->reduce_recorded[slot] = false;
  }
  if (!b->ev_reduce_done[slot]) {
    GHIP(ctx, hipEventCreateWithFlags(&b->ev_reduce_done[slot], kSyncEventFlags));
    for (int k = 0; k < kSideStreams; ++k) GHIP(ctx, hipEventCreateWithFlags(&b->ev_render_done[k][slot], kSyncEventFlags));
  }
  uint32_t count[kSideStreams] = {}, offset[kSideStreams] = {}; // per stream: Welsh base kinds first, then the bank streams
  // The MIX kernel (kernels.h; round 6): the four class-specialised base kinds in three launches, one per kind stream, each over a
  // third of their workgroups (every third entry of the kind-sorted list); the exact-f64 kinds keep their per-kind kernels.
  const bool mix = uniform && ctx->mix_kernel;
  if (uniform) {
    for (uint32_t base = 0, at = 0; base < (uint32_t)kBaseKinds; ++base) {
      offset[base] = at;
      for (int c = 0; c < kClassCombos; ++c) count[base] += b->wgs_of_kind[base * kClassCombos + c];
      at += count[base];
    }
    if (mix) { for (int sec = 0; sec < 3; ++sec) count[sec] = b->mix_cnt[sec]; count[3] = 0; }
  } else {
    count[b->stream_slot] = rows; // one kernel, on this bank's side stream (the loop below runs once)
  }
  if (ctx->need_fork) { // side streams must see what the ctx stream did since the last join (note events, uploads)
    GHIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (bool& f : ctx->fork_pending) f = true;
    ctx->need_fork = false;
  }
  const RenderConsts rc = render_consts_of(ctx);
  const dim3 blk(kThreads);
  // paced: the HOST waits for the reduction that frees this slot's rows (two blocks back: long done), so that the waits below are
  // dropped when they are made and the render streams carry no wait packet (docs/STREAMS.md item 13)
  // (A deadline that passes here loses nothing either: the streams wait for that reduction themselves below — with a wait packet this
  // once — the block is rendered and registered, and the call reports the deadline when it returns.  Found under GROOVE_SAFE_STREAMS=1,
  // where the bank streams are the kind streams and a blocked stream holds back more: the call used to return BEFORE its render, and a
  // caller that carried on had skipped a block of that bank.)
  int late = 0;
  if (paced && b->reduce_recorded[slot]) {
    const hipError_t e = wait_deadline(ctx, nullptr, b->ev_reduce_done[slot], "groove_bank_render_mix_paced: the slot's previous reduction");
    if (e == hipErrorNotReady) late = 2;
    else GHIP(ctx, e);
  }
  uint32_t used = 0;
  for (int k = kSideStreams - 1; k >= 0; --k) { // most expensive Welsh kind first
    if (!count[k]) continue;
    hipStream_t st = side_stream_of(ctx, k);
    used |= 1u << k;
    if (ctx->fork_pending[k]) { GHIP(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0)); ctx->fork_pending[k] = false; }
    if (b->reduce_recorded[slot]) GHIP(ctx, hipStreamWaitEvent(st, b->ev_reduce_done[slot], 0));
    // the block's "render done" event completes with the render kernel itself (kernels.h launch_bound): no record packet between
    // this block's kernel and the next block's on the stream
    const hipEvent_t done = ctx->bind_events ? b->ev_render_done[k][slot] : nullptr;
    if (uniform && mix && k < 3) {
      UniformArgs a = uniform_args(b, b->d_pipe_part[slot], b->d_pipe_part[slot], 0, 0, frames, count[k]);
      const size_t o = b->wg_list_cap + b->mix_off[k]; // the section's entries of the striped copies (welsh_upload_params)
      a.wg_list = b->d_wg_list + o; a.wg_cls = b->d_wg_cls + o; a.wg_f32 = b->d_wg_f32 + o;
      launch_welsh_uniform_mix(a, b->d_wg_base + o, st, done);
    } else if (uniform) {
      UniformArgs a = uniform_args(b, b->d_pipe_part[slot], b->d_pipe_part[slot], offset[k], 0, frames, count[k]);
      launch_welsh_kind(k, a, st, true, done);
    } else if (tp) {
      launch_tp(b, frames, true, 0, b->d_pipe_part[slot], b->d_pipe_part[slot], st, nullptr, done);
    } else if (small_uniform) { // all base kinds in one launch on this bank's stream
      UniformArgs a = uniform_args(b, b->d_pipe_part[slot], b->d_pipe_part[slot], 0, 0, frames, rows);
      launch_small_uniform(b, a, st, true, frames, done);
    } else if (b->kind == BANK_WELSH) {
      launch_bound(welsh_render_kernel<true>, dim3(rows), blk, st, done, b->d_params, b->d_state, b->n, frames, (size_t)0, b->d_pipe_part[slot], b->d_pipe_part[slot], rc);
    } else if (b->kind == BANK_FM) {
      launch_bound(fm_render_kernel<true>, dim3(rows), blk, st, done, b->d_params, b->d_state, b->n, frames, (size_t)0, b->d_pipe_part[slot], b->d_pipe_part[slot]);
    } else {
      launch_bound(sampler_render_kernel<true>, dim3(rows), blk, st, done, b->d_params, b->d_state, b->n, frames, (size_t)0, b->d_pipe_part[slot], b->d_pipe_part[slot], (const float*)b->d_pcm);
    }
    if (!done) GHIP(ctx, hipEventRecord(b->ev_render_done[k][slot], st));
    if (!paced) GHIP(ctx, hipStreamWaitEvent(ctx->stream, b->ev_render_done[k][slot], 0));
    ctx->side_busy[k] = true;
  }
  b->ctx_touched = false; // the bank's stream(s) have waited for whatever the ctx stream did to its state (ev_fork above)
  if (paced) {
    // this block's reduction is launched by the bank's NEXT paced call (or a flush point); the previous block's is launched now: its
    // renders were submitted a whole call ago, the host waits for them (this block's are already queued behind them)
    if (b->paced.active) {
      int rc = paced_reduce(b, true);
      if (rc == 2) {
        // The host's wait for the previous block's renders passed its deadline.  This block's kernels are already queued and a
        // bank has ONE pending record: the previous block's reduction is queued behind device-side waits instead (it completes
        // when the renders do), this block is registered below, and the call still reports the deadline (2, unchanged).
        late = 2;
        rc = paced_reduce(b, false);
      }
      if (rc) return rc;
    }
    b->paced.active = true; b->paced.slot = slot; b->paced.rows = rows; b->paced.frames = frames; b->paced.used = used; b->paced.bus = bus_dev; b->paced.accumulate = accumulate;
    b->reduce_recorded[slot] = false; // (recorded when the reduction is launched)
    ctx->paced_order.push_back(b);
    GHIP(ctx, hipGetLastError());
    return late;
  }
  launch_reduce(ctx, b->d_pipe_part[slot], rows, frames, b->d_pipe_seg[slot], bus_dev, accumulate);
  GHIP(ctx, hipEventRecord(b->ev_reduce_done[slot], ctx->stream));
  b->reduce_recorded[slot] = true;
  GHIP(ctx, hipGetLastError());
  return 0;
}
// groove_bank_render_mix for a project whose banks render side by side, PACED by the host (include/groove_hip.h).
int groove_bank_render_mix_paced(groove_bank* b, uint32_t frames, float* bus_dev, int accumulate) {
  if (!b || !bus_dev) return fail(nullptr, "groove_bank_render_mix_paced: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (frames == 0) return 0;
  if (frames > 4096) return fail(ctx, "groove_bank_render_mix_paced: frames > 4096");
  GHIP(ctx, hipSetDevice(ctx->device));
  if (bus_flush_deferred(ctx) || bus_flush_ap(ctx)) return 1; // (the other deferrals' pending rows come first; paced reductions of other banks stay pending)
  if (flush_events(b, use_tp(b, frames))) return 1;
  return render_mix_pipelined(b, frames, bus_dev, accumulate, true);
}
// How many partial rows a bank's deferred render writes (0: the bank does not take a deferred form and goes through
// groove_bank_render_mix), and for a sampler bank the voices-per-workgroup spread of that form.
static uint32_t deferred_rows_of(groove_bank* b, uint32_t frames, uint32_t* svpw_out = nullptr) {
  groove_ctx* ctx = b->ctx;
  if (svpw_out) *svpw_out = 0;
  if (ctx->pipeline_min_waves <= 1 || frames == 0 || frames > 4096) return 0;
  if (!use_tp(b, frames)) { // the all-kinds / role-split kernel of a mid-size Welsh bank, its rows summed by the next block's launch
    if (!(b->kind == BANK_WELSH && b->n_vwaves && b->n_vwaves < ctx->pipeline_min_waves)) return 0;
    uint32_t n_spec = 0;
    for (int k = 0; k < 4 * kClassCombos; ++k) n_spec += b->wgs_of_kind[k];
    const uint32_t urows = fused_rows(b, frames);
    return n_spec && urows <= 2048 ? urows : 0;
  }
  uint32_t svpw = 0; // sampler: spread over more workgroups than the form with a reduction launch would (welsh_tp.h)
  if (b->kind == BANK_SAMPLER) { svpw = sampler_tp_vpw_deferred(b->n); if (sampler_tp_workgroups(b->n, svpw) > 512) svpw = 0; }
  // (up to 2,048 rows: the 512 columns' workgroups then take two to four batches of rows, a few us of a render that is long by then)
  const uint32_t rows = svpw ? sampler_tp_workgroups(b->n, svpw) : fused_rows(b, frames);
  if (rows == 0 || rows > 2048 || frames > kTpMaxFrames) return 0;
  if (svpw_out) *svpw_out = svpw;
  return rows;
}
// The two row buffers of the deferred renders, BOTH sized at once and for the LARGEST need among the ctx's banks: growing one of
// them later would have to flush the pending block through the reduction kernels — whose order of additions differs from the
// carried reduction's — and a project's first run would then round one block differently from every later run (seen by
// tools/soak.py: 1 CRC in 138,107 repeats of config #4; with banks of different row counts taking turns — config #5's per-GPU
// share — the later banks of the first block would still have grown the buffers had they been sized for the first caller only).
static int ensure_dpart(groove_ctx* ctx, size_t need, uint32_t frames) {
  if (ctx->dpart_cap[0] >= need && ctx->dpart_cap[1] >= need) return 0;
  for (groove_bank* o : ctx->banks) need = std::max(need, (size_t)deferred_rows_of(o, frames) * 2 * frames);
  if (bus_flush(ctx)) return 1;                      // (the pending rows may live in a buffer that is about to go)
  GHIP(ctx, wait_deadline(ctx, ctx->stream, nullptr, "deferred partial rows"));
  for (int slot = 0; slot < 2; ++slot) {
    if (ctx->dpart_cap[slot] >= need) continue;
    if (ctx->d_dpart[slot]) GHIP(ctx, hipFree(ctx->d_dpart[slot]));
    ctx->d_dpart[slot] = nullptr; ctx->dpart_cap[slot] = 0;
    GHIP(ctx, hipMalloc(&ctx->d_dpart[slot], need * 4));
    ctx->dpart_cap[slot] = need;
  }
  return 0;
}
// Pending PACED reductions of any bank go onto their buses before a call of another form adds to a bus (include/groove_hip.h: "any
// unpaced render or mix flushes them"): call order is the order of a bus's sums.
static int flush_paced(groove_ctx* ctx) {
  if (bus_flush_ap(ctx)) return 1; // (lane sums waiting on the all-pass stream: the order of a bus's sums is the order of the calls)
  while (!ctx->paced_order.empty())
    if (const int rc = paced_reduce(ctx->paced_order.front(), false)) return rc;
  return 0;
}
// Fused render + mix whose bus reduction is left to the bank's NEXT deferred render (welsh_tp.h, tp_reduce_prev) — or to
// whatever waits for the ctx stream, records an event on it or touches a bus (bus_flush).  For banks that render time-parallel
// on the ctx stream with at most 2,048 partial rows; anything else is groove_bank_render_mix.
int groove_bank_render_mix_deferred(groove_bank* b, uint32_t frames, float* bus_dev, int accumulate) {
  if (!b || !bus_dev) return fail(nullptr, "groove_bank_render_mix_deferred: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (frames == 0) return 0;
  // (several banks of a small project may take turns on the ctx stream this way — each render carries the reduction of the one
  // before it, in submission order — instead of side by side on side streams with their cross-queue waits: the caller's choice)
  // A Welsh bank too big for the time-parallel form and too small for the per-kind pipeline (the all-kinds or a role-split
  // kernel on the ctx stream, then two reduction launches in line behind it: ~18 us of a 125,000-voice shard's 130) takes the same
  // deferral, the next block's kernel summing the rows — when the launch that would carry them exists (some workgroup of the four
  // class-specialised kinds) and the rows are few enough (deferred_rows_of).
  uint32_t svpw = 0;
  const uint32_t rows = deferred_rows_of(b, frames, &svpw);
  if (rows == 0) return groove_bank_render_mix(b, frames, bus_dev, accumulate);
  const bool tp = use_tp(b, frames);
  GHIP(ctx, hipSetDevice(ctx->device));
  if (const int rc = flush_paced(ctx)) return rc;
  if (flush_events(b, tp)) return 1;
  if (ctx_join(ctx)) return 1;
  const size_t need = (size_t)rows * 2 * frames;
  if (ensure_dpart(ctx, need, frames)) return 1;
  const int slot = ctx->dpart_next;
  ctx->dpart_next ^= 1;
  TpPrev prev;
  if (ctx->deferred.rows) { prev.rows = ctx->deferred.rows; prev.bus = ctx->deferred.bus; prev.n_rows = ctx->deferred.n_rows; prev.frames = ctx->deferred.frames; prev.accumulate = ctx->deferred.accumulate; }
  deferred_taken(ctx);
  b->ctx_touched = true;
  if (!tp) {
    UniformArgs a = uniform_args(b, ctx->d_dpart[slot], ctx->d_dpart[slot], 0, 0, frames, rows);
    a.prev = prev;
    launch_small_uniform(b, a, ctx->stream, true, frames);
  } else {
    launch_tp(b, frames, true, 0, ctx->d_dpart[slot], ctx->d_dpart[slot], ctx->stream, nullptr, nullptr, prev.rows ? &prev : nullptr, svpw);
  }
  GHIP(ctx, hipGetLastError());
  ctx->deferred.rows = ctx->d_dpart[slot]; ctx->deferred.bus = bus_dev; ctx->deferred.n_rows = rows; ctx->deferred.frames = frames; ctx->deferred.accumulate = accumulate;
  return 0;
}
// groove_bank_render_mix_deferred for SEVERAL small banks of different kinds at once: ONE launch for the whole block (welsh_tp.h
// tp_mixed_kernel: a grid whose workgroups dispatch on their index into the Welsh / FM / sampler time-parallel bodies), one row
// buffer, one carried reduction — Orchestrator::gather_audio's one sum over all instruments (orchestrator.rs:397-410).  Takes
// projects with at most one time-parallel bank of each kind and at most 2,048 rows in all; anything else goes bank by bank
// through groove_bank_render_mix_deferred, in the order given.
int groove_banks_render_mix_deferred(groove_ctx* ctx, groove_bank* const* banks, uint32_t n_banks, uint32_t frames, float* bus_dev, int accumulate) {
  if (!ctx || !bus_dev || (n_banks && !banks)) return fail(ctx, "groove_banks_render_mix_deferred: NULL argument");
  if (frames == 0) return 0;
  if (n_banks == 0) { // nothing patched: silence (orchestrator.rs:1452-1455)
    if (bus_flush(ctx)) return 1;
    if (!accumulate) GHIP(ctx, hipMemsetAsync(bus_dev, 0, (size_t)frames * 8, ctx->stream));
    return 0;
  }
  groove_bank* of_kind[3] = {nullptr, nullptr, nullptr}; // BANK_WELSH, BANK_FM, BANK_SAMPLER
  bool ok = n_banks >= 2 && n_banks <= 3 && ctx->pipeline_min_waves > 1 && frames <= kTpMaxFrames;
  for (uint32_t i = 0; i < n_banks; ++i) {
    groove_bank* b = banks[i];
    if (!b) return fail(ctx, "groove_banks_render_mix_deferred: NULL bank");
    if (b->ctx != ctx) return fail(ctx, "groove_banks_render_mix_deferred: a bank of another ctx");
    const int k = b->kind == BANK_WELSH ? 0 : b->kind == BANK_FM ? 1 : 2;
    if (of_kind[k] || !use_tp(b, frames) || (k == 0 && b->tp_full_coef)) ok = false;
    of_kind[k] = b;
  }
  TpMixedArgs m{};
  uint32_t grid = 0;
  if (ok) {
    // the kinds' workgroup ranges in dispatch order, the most expensive kind first (measured, profiles/r05_mixed_ab.log: Welsh | FM |
    // sampler 0.0408 ms per block for config #5's share, sampler-first orders 0.0414 - 0.0423; 8 sampler voices per wavefront: 2 / 4 /
    // 8 / 16 -> 0.0426 / 0.0413 / 0.0408 / 0.0407, and 16 costs the 4,096-voice project 0.0199 against 0.0183)
    if (groove_bank* b = of_kind[0]) { const uint32_t v = tp_vpw(b); m.welsh = TpMixedBank{b->d_params, b->d_state, b->n, v, grid, welsh_tp_grid(b->n, v)}; grid += m.welsh.n_wg; }
    if (groove_bank* b = of_kind[1]) { const uint32_t v = tp_vpw(b); m.fm = TpMixedBank{b->d_params, b->d_state, b->n, v, grid, welsh_tp_workgroups(b->n, v)}; grid += m.fm.n_wg; }
    if (groove_bank* b = of_kind[2]) {
      m.sampler = TpMixedBank{b->d_params, b->d_state, b->n, kMixedSamplerVpw, grid, mixed_sampler_workgroups(b->n, kMixedSamplerVpw)}; grid += m.sampler.n_wg;
      m.pcm = b->d_pcm;
    }
    if (grid == 0 || grid > 2048) ok = false;
  }
  if (!ok) {
    for (uint32_t i = 0; i < n_banks; ++i)
      if (const int rc = groove_bank_render_mix_deferred(banks[i], frames, bus_dev, accumulate || i > 0)) return rc;
    return 0;
  }
  GHIP(ctx, hipSetDevice(ctx->device));
  if (const int rc = flush_paced(ctx)) return rc;
  for (uint32_t i = 0; i < n_banks; ++i)
    if (flush_events(banks[i], true)) return 1;
  if (ctx_join(ctx)) return 1;
  if (ensure_dpart(ctx, (size_t)grid * 2 * frames, frames)) return 1;
  const int slot = ctx->dpart_next;
  ctx->dpart_next ^= 1;
  if (ctx->deferred.rows) { m.prev.rows = ctx->deferred.rows; m.prev.bus = ctx->deferred.bus; m.prev.n_rows = ctx->deferred.n_rows; m.prev.frames = ctx->deferred.frames; m.prev.accumulate = ctx->deferred.accumulate; }
  deferred_taken(ctx);
  m.rows = ctx->d_dpart[slot]; m.rc = render_consts_of(ctx); m.frames = frames;
  static const InlineEvents no_events{};
  launch_tp_mixed(m, of_kind[2] ? of_kind[2]->inline_ev : no_events, grid, ctx->stream);
  for (uint32_t i = 0; i < n_banks; ++i) banks[i]->ctx_touched = true;
  if (of_kind[2]) of_kind[2]->inline_ev.n = 0;
  GHIP(ctx, hipGetLastError());
  ctx->deferred.rows = ctx->d_dpart[slot]; ctx->deferred.bus = bus_dev; ctx->deferred.n_rows = grid; ctx->deferred.frames = frames; ctx->deferred.accumulate = accumulate;
  return 0;
}
int groove_bus_flush(groove_ctx* ctx) {
  if (!ctx) return fail(nullptr, "groove_bus_flush: ctx is NULL");
  GHIP(ctx, hipSetDevice(ctx->device));
  return bus_flush(ctx);
}
int groove_bank_render_mix(groove_bank* b, uint32_t frames, float* bus_dev, int accumulate) {
  if (!b || !bus_dev) return fail(nullptr, "groove_bank_render_mix: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (frames == 0) return 0;
  if (bus_flush(ctx)) return 1;
  if (frames > 4096) return fail(ctx, "groove_bank_render_mix: frames > 4096");
  GHIP(ctx, hipSetDevice(ctx->device));
  if (flush_events(b, use_tp(b, frames))) return 1;
  // Asynchronous form (kernels on side streams, only the bus reductions on the ctx stream):
  //  - a large Welsh bank (>= ~550,000 voices) runs one kernel per base kind and pipelines its own blocks
  //    (+11 % at 1,000,000 voices, +7 % at 600,000); below that the all-kinds kernel wins, run block after block on the
  //    ctx stream (500,000 voices: 0.342 ms against 0.357; 300,000: 0.250 against 0.275; through the block pipeline
  //    it is no faster: 250,000 voices 0.222 against 0.215);
  //  - in a project of several banks (synths, samplers) every bank takes it, so that the banks of one block
  //    run beside each other instead of one after the other (mixed-131072: 0.46 -> 0.2x ms per block).
  const bool force = ctx->pipeline_min_waves <= 1;
  const bool big = b->kind == BANK_WELSH && b->n_vwaves >= ctx->pipeline_min_waves;
  if (big || force || ctx->banks.size() > 1) return render_mix_pipelined(b, frames, bus_dev, accumulate);
  if (ctx_join(ctx)) return 1; // earlier pipelined blocks of this bank may still be running on the side streams
  const uint32_t rows = fused_rows(b, frames);
  const uint32_t cols = 2 * frames;
  const uint32_t rows_per_seg = 64;
  const uint32_t segs = (rows + rows_per_seg - 1) / rows_per_seg;
  if (ctx->fpart_cap < (size_t)rows * cols) {
    if (ctx->d_fpart) GHIP(ctx, hipFree(ctx->d_fpart));
    GHIP(ctx, hipMalloc(&ctx->d_fpart, (size_t)rows * cols * 4));
    ctx->fpart_cap = (size_t)rows * cols;
  }
  if (ensure_seg_buffer(ctx, &ctx->d_fseg, &ctx->fseg_cap, (size_t)segs * cols)) return 1;
  if (launch_render(b, frames, true, 0, ctx->d_fpart, ctx->d_fpart)) return 1;
  launch_reduce(ctx, ctx->d_fpart, rows, frames, ctx->d_fseg, bus_dev, accumulate);
  GHIP(ctx, hipGetLastError());
  return 0;
}
const char* groove_bank_kernel_form(groove_bank* b, uint32_t frames, int fused) {
  if (!b) return "";
  groove_ctx* ctx = b->ctx;
  if (use_tp(b, frames)) return b->kind == BANK_WELSH ? (tp_vpw(b) == 2 ? "welsh_tp_kernel (time-parallel, two voices per wavefront)" : "welsh_tp_kernel (time-parallel, one wavefront per voice)") : b->kind == BANK_FM ? (tp_vpw(b) == 4 ? "fm_tp_kernel (time-parallel, four voices per wavefront)" : "fm_tp_kernel (time-parallel)") : "sampler_tp_kernel (time-parallel)";
  if (b->kind == BANK_FM) return "fm_render_kernel (serial, one voice per lane)";
  if (b->kind == BANK_SAMPLER) return "sampler_render_kernel (serial, one voice per lane)";
  if (!b->n_vwaves) return "welsh_render_kernel (per-lane parameters)";
  const bool pipelined = fused && (b->n_vwaves >= ctx->pipeline_min_waves || ctx->pipeline_min_waves <= 1);
  if (b->n_vwaves >= ctx->pipeline_min_waves || (fused && ctx->pipeline_min_waves <= 1))
    return pipelined ? (ctx->mix_kernel ? "welsh_render_uniform_mix_kernel (big-bank form: three launches per block over thirds of the kind-sorted workgroups, one launch per base kind for the exact-f64 kinds only; class-specialised bodies, blocks pipelined)"
                                        : "welsh_render_uniform_kernel (one launch per base kind, class-specialised bodies, blocks pipelined)")
                     : "welsh_render_uniform_kernel (one launch per base kind, class-specialised bodies)";
  if (use_split(b, frames) && split_roles_of(b) == 4) return "welsh_render_split4_kernel (role-split: four wavefronts per 64 voices, pipelined over the frames)";
  if (use_split(b, frames)) return split_roles_of(b) == 3 ? "welsh_render_split_kernel (role-split: three wavefronts per 64 voices, pipelined over the frames)"
                                                            : "welsh_render_split_kernel (role-split: two wavefronts per 64 voices, pipelined over the frames)";
  return "welsh_render_uniform_any_kernel (all base kinds in one launch, class-specialised bodies)";
}
int groove_bank_reset(groove_bank* b) {
  if (!b) return fail(nullptr, "groove_bank_reset: bank is NULL");
  groove_ctx* ctx = b->ctx;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (ctx_join(ctx)) return 1;
  b->side_mode = 0;
  b->ctx_touched = true;
  b->pending.clear();
  b->inline_ev.n = 0;
  StateWords w{};
  if (b->kind == BANK_WELSH) { const WelshState s = initial_welsh_state(); std::memcpy(w.w, &s, sizeof(s)); }
  else if (b->kind == BANK_FM) { const FmState s = initial_fm_state(); std::memcpy(w.w, &s, sizeof(s)); }
  else { const SamplerState s{0, 0, 0, 0}; std::memcpy(w.w, &s, sizeof(s)); }
  hipLaunchKernelGGL(state_fill_kernel, dim3(blocks_for(b->n)), dim3(kThreads), 0, ctx->stream, b->d_state, b->n, b->sw, w);
  GHIP(ctx, hipGetLastError());
  return 0;
}
uint32_t groove_bank_state_words(groove_bank* b) { return b ? b->sw : 0; }
int groove_bank_download_state(groove_bank* b, uint32_t* host_words) {
  if (!b || !host_words) return fail(nullptr, "groove_bank_download_state: NULL argument");
  groove_ctx* ctx = b->ctx;
  if (flush_events(b)) return 1;
  if (ctx_join(ctx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  if (b->perm.empty()) {
    GHIP(ctx, ctx_memcpy(ctx, host_words, b->d_state, (size_t)b->sw * b->n * 4, hipMemcpyDeviceToHost));
  } else { // internal lane order -> caller's voice order
    std::vector<uint32_t> tmp((size_t)b->sw * b->n);
    GHIP(ctx, ctx_memcpy(ctx, tmp.data(), b->d_state, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t w = 0; w < b->sw; ++w)
      for (uint32_t i = 0; i < b->n; ++i) host_words[(size_t)w * b->n + b->perm[i]] = tmp[(size_t)w * b->n + i];
  }
  return 0;
}

// ============================================================================ effects
int groove_fx_create(groove_ctx* ctx, uint32_t kind, const groove_fx_params* p, uint32_t n, groove_fx** out) {
  if (!ctx || !p || !out) return fail(ctx, "groove_fx_create: NULL argument");
  if (n == 0) return fail(ctx, "groove_fx_create: n == 0");
  if (kind >= GROOVE_FX_KIND_COUNT) return fail(ctx, "groove_fx_create: unknown effect kind");
  GHIP(ctx, hipSetDevice(ctx->device));
  groove_fx* fx = new groove_fx();
  fx->ctx = ctx; fx->kind = kind; fx->n = n;
  fx->p.assign(p, p + n);
  if (fx_check_uniform(fx)) { delete fx; return 1; }
  if (hipMalloc(&fx->d_fa, n * 4) != hipSuccess || hipMalloc(&fx->d_fb, n * 4) != hipSuccess ||
      hipMalloc(&fx->d_ua, n * 4) != hipSuccess || hipMalloc(&fx->d_wet, n * 4) != hipSuccess ||
      hipMalloc(&fx->d_coef, (size_t)6 * n * 8) != hipSuccess) {
    groove_fx_destroy(fx);
    return fail(ctx, "groove_fx_create: hipMalloc failed");
  }
  if (fx_setup_state(fx) || fx_upload_params(fx)) { groove_fx_destroy(fx); return 1; }
  ctx->fxs.push_back(fx);
  *out = fx;
  return 0;
}
// A reverb whose all-passes went to the all-pass stream last: the ctx stream waits for them before it touches the lines itself.
static int fx_ap_settle(groove_fx* fx) {
  if (!fx->ap_busy) return 0;
  fx->ap_busy = false;
  return ap_join(fx->ctx);
}
int groove_fx_destroy(groove_fx* fx) {
  if (!fx) return 0;
  groove_ctx* ctx = fx->ctx;
  (void)fx_acquire_ctx(fx);
  (void)fx_ap_settle(fx);
  (void)hipStreamSynchronize(ctx->stream);
  if (fx->ev_done) (void)hipEventDestroy(fx->ev_done);
  auto it = std::find(ctx->fxs.begin(), ctx->fxs.end(), fx);
  if (it != ctx->fxs.end()) ctx->fxs.erase(it);
  (void)hipFree(fx->d_fa); (void)hipFree(fx->d_fb); (void)hipFree(fx->d_ua); (void)hipFree(fx->d_wet);
  (void)hipFree(fx->d_coef); (void)hipFree(fx->d_st); (void)hipFree(fx->d_ring); (void)hipFree(fx->d_tmp);
  delete fx;
  return 0;
}
int groove_fx_reset(groove_fx* fx) {
  if (!fx) return fail(nullptr, "groove_fx_reset: fx is NULL");
  groove_ctx* ctx = fx->ctx;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (fx_acquire_ctx(fx) || fx_ap_settle(fx)) return 1;
  const size_t ln = 2 * (size_t)fx->n;
  if (fx->d_st) GHIP(ctx, hipMemsetAsync(fx->d_st, 0, 4 * ln * 8, ctx->stream));
  if (fx->d_ring) GHIP(ctx, hipMemsetAsync(fx->d_ring, 0, fx->ring_rows * ln * 4, ctx->stream));
  fx->w = 0;
  for (uint32_t& w : fx->geo.w) w = 0;
  return 0;
}
// ---- effect stages ---------------------------------------------------------------------
// Can this effect, for a block of `frames`, run as a stage of a fused run (kernels.h, fx_run_kernel)?  The element-wise
// kinds always can; a delay line can when it is at least a block long (no feedback inside the block).
static bool fx_run_capable(const groove_fx* fx, uint32_t frames) {
  switch (fx->kind) {
    case GROOVE_FX_GAIN: case GROOVE_FX_BITCRUSHER: case GROOVE_FX_LIMITER: case GROOVE_FX_COMPRESSOR: return true;
    case GROOVE_FX_DELAY: return fx->N >= frames;
    case GROOVE_FX_CHORUS: {
      const uint32_t nearest = fx->N - (fx->voices - 1) * fx->spacing; // newest tap: pushed this many frames ago
      return nearest >= frames && (fx->voices == 1 || fx->spacing >= frames);
    }
    case GROOVE_FX_REVERB: { // combs in the run, the two all-passes behind it
      uint32_t shortest_comb = fx->geo.N[0];
      for (int i = 1; i < 4; ++i) shortest_comb = std::min(shortest_comb, fx->geo.N[i]);
      return fx->all_wet && shortest_comb >= frames && std::min(fx->geo.N[4], fx->geo.N[5]) >= 32;
    }
    default: return false;
  }
}
static bool fx_is_iir(uint32_t kind) {
  switch (kind) {
    case GROOVE_FX_BIQUAD_LP12: case GROOVE_FX_BIQUAD_HP12: case GROOVE_FX_BIQUAD_BP12: case GROOVE_FX_BIQUAD_BS12:
    case GROOVE_FX_BIQUAD_AP12: case GROOVE_FX_BIQUAD_PEAK12: case GROOVE_FX_BIQUAD_LSHELF12: case GROOVE_FX_BIQUAD_HSHELF12:
    case GROOVE_FX_BIQUAD_LP24: return true;
    default: return false;
  }
}
static void fx_advance(groove_fx* fx, uint32_t frames) { // the delay lines' write positions after a block
  if (fx->kind == GROOVE_FX_DELAY || fx->kind == GROOVE_FX_CHORUS) fx->w = (uint32_t)(((uint64_t)fx->w + frames) % fx->N);
  if (fx->kind == GROOVE_FX_REVERB)
    for (int i = 0; i < 6; ++i) fx->geo.w[i] = (uint32_t)(((uint64_t)fx->geo.w[i] + frames) % fx->geo.N[i]);
}
// One launch for `count` (<= kRunMaxStages) run-capable stages in chain order; a reverb, if there is one, is the last,
// and its all-passes follow.
static int fx_launch_run(groove_ctx* ctx, groove_fx* const* run, uint32_t count, groove_block* io, uint32_t frames, hipStream_t st, bool last) {
  const uint32_t n = io->n;
  const size_t chs = (size_t)io->cap * n;
  FxRunArgs a{};
  for (uint32_t s = 0; s < count; ++s) {
    const groove_fx* fx = run[s];
    a.st[s] = FxRunStage{fx->kind, fx->N, fx->w, fx->voices, fx->spacing, 0, fx->d_ring, fx->d_fa, fx->d_fb, fx->d_ua, fx->d_wet};
  }
  a.n_stages = count; a.n = n;
  a.src = io->d; a.src_chs = chs;
  a.dst = io->d; a.dst_chs = chs;
  groove_fx* rv = run[count - 1]->kind == GROOVE_FX_REVERB ? run[count - 1] : nullptr;
  bool direct = false, on_ap = false;
  if (rv) {
    a.geo = rv->geo;
    // the direct all-pass form wants the comb sum in a staging block (it reads other frames than the one it writes)
    const uint32_t shortest_ap = std::min(rv->geo.N[4], rv->geo.N[5]);
    direct = !ctx->seq_allpass && !ctx->chunked_allpass && frames <= 8 * shortest_ap && (size_t)2 * io->cap * n * 4 <= ((size_t)1 << 30);
    // The all-pass stream: the chain's LAST launch (nothing on the ctx stream reads the block behind it; its lane sums go to the
    // bus through groove_mix / groove_mix_deferred) leaves the ctx stream, so the next block's run follows this one directly.
    on_ap = direct && last && ctx->fx_ap_stream >= 0 && st == ctx->stream && io->ev_free;
    if (!on_ap && rv->ap_busy) { if (ap_join(ctx)) return 1; rv->ap_busy = false; }
    if (direct && on_ap) {
      if (io->stage_cap < io->cap) { // (first use of the block on this path: nothing of the block's is in flight on that stream yet)
        if (io->d_stage) { if (ap_join(ctx)) return 1; GHIP(ctx, wait_deadline(ctx, st, nullptr, "effect staging block")); GHIP(ctx, hipFree(io->d_stage)); io->d_stage = nullptr; io->stage_cap = 0; }
        GHIP(ctx, hipMalloc(&io->d_stage, (size_t)2 * io->cap * n * 4));
        io->stage_cap = io->cap;
      }
      a.dst = io->d_stage; a.dst_chs = (size_t)io->stage_cap * n;
    } else if (direct) {
      if (rv->tmp_cap < io->cap) {
        GHIP(ctx, wait_deadline(ctx, st, nullptr, "effect staging block"));
        if (rv->d_tmp) { GHIP(ctx, hipFree(rv->d_tmp)); rv->d_tmp = nullptr; rv->tmp_cap = 0; }
        GHIP(ctx, hipMalloc(&rv->d_tmp, (size_t)2 * io->cap * n * 4));
        rv->tmp_cap = io->cap;
      }
      a.dst = rv->d_tmp; a.dst_chs = (size_t)rv->tmp_cap * n;
    }
  }
  const dim3 blk(kThreads);
  const uint32_t V = n % 4 == 0 ? 4 : 1, wg_per_ch = fx_wg_per_ch(n, V);
  // the chain's last launch leaves the block's lane sums for groove_mix (kernels.h, fx_row_sum)
  float* rows = nullptr;
  if (on_ap) { // one of the block's two buffers that only that stream's kernels write
    const int f = io->ap_flip;
    const size_t need = (size_t)wg_per_ch * 2 * frames;
    if (io->sums_ap_cap[f] < need) {
      if (io->d_sums_ap[f]) { if (bus_flush_ap(ctx) || ap_join(ctx)) return 1; GHIP(ctx, wait_deadline(ctx, st, nullptr, "block row sums")); GHIP(ctx, hipFree(io->d_sums_ap[f])); io->d_sums_ap[f] = nullptr; io->sums_ap_cap[f] = 0; }
      const size_t cap = std::max(need, (size_t)wg_per_ch * 2 * std::min<uint32_t>(io->cap, 4096));
      GHIP(ctx, hipMalloc(&io->d_sums_ap[f], cap * 4));
      io->sums_ap_cap[f] = cap;
    }
    io->sums_valid = false;
    rows = io->d_sums_ap[f];
  } else if (last) {
    rows = block_sums(io, wg_per_ch, frames);
    if (!rows) return 1;
  }
  a.frames = frames; a.wg_per_ch = wg_per_ch;
  if (ctx->deferred.rows && st == ctx->stream) { // groove_mix_deferred: this launch sums the pending block's rows onto its bus
    a.prev.rows = ctx->deferred.rows; a.prev.bus = ctx->deferred.bus; a.prev.n_rows = ctx->deferred.n_rows; a.prev.frames = ctx->deferred.frames; a.prev.accumulate = ctx->deferred.accumulate;
    deferred_taken(ctx);
  }
  a.rows = (rv && direct) ? nullptr : rows;
  if (rv && !direct) rows = nullptr; // the sequential / chunked all-pass kernels run last and leave none
  if (V == 4) hipLaunchKernelGGL(fx_run_kernel<4>, dim3(2 * wg_per_ch, frames), blk, 0, st, a);
  else hipLaunchKernelGGL(fx_run_kernel<1>, dim3(2 * wg_per_ch, frames), blk, 0, st, a);
  if (rv) {
    const ReverbGeom& g = rv->geo;
    if (direct) {
      AllpassDirectArgs d{};
      d.src = a.dst; d.src_chs = a.dst_chs;
      d.dst = io->d; d.dst_chs = chs;
      d.ring = rv->d_ring;
      for (int i = 0; i < 2; ++i) {
        d.old_base[i] = g.base[4 + i]; d.new_base[i] = rv->ap_alt[i];
        d.N[i] = g.N[4 + i]; d.w[i] = g.w[4 + i]; d.g[i] = g.g[4 + i];
      }
      d.n = n; d.frames = frames;
      d.rows = rows; d.wg_per_ch = wg_per_ch;
      const uint32_t grid_rows = std::max(frames, std::max(g.N[4], g.N[5]));
      hipStream_t ast = st;
      if (on_ap) {
        const int k = ctx->fx_ap_stream;
        ast = ap_stream(ctx);
        if (ap_events(ctx)) return 1;
        GHIP(ctx, hipEventRecord(ctx->ev_ap_run, st));
        GHIP(ctx, hipStreamWaitEvent(ast, ctx->ev_ap_run, 0));
        if (ctx->deferred_ap.rows) { // the previous block's lane sums: written by that stream's last kernel, summed by this one
          d.prev.rows = ctx->deferred_ap.rows; d.prev.bus = ctx->deferred_ap.bus; d.prev.n_rows = ctx->deferred_ap.n_rows; d.prev.frames = ctx->deferred_ap.frames; d.prev.accumulate = ctx->deferred_ap.accumulate;
          ctx->deferred_ap.rows = nullptr;
        }
        if (V == 4) hipLaunchKernelGGL(fx_reverb_allpass_direct_kernel<4>, dim3(2 * wg_per_ch, grid_rows), blk, 0, ast, d);
        else hipLaunchKernelGGL(fx_reverb_allpass_direct_kernel<1>, dim3(2 * wg_per_ch, grid_rows), blk, 0, ast, d);
        GHIP(ctx, hipEventRecord(io->ev_ready[k], ast)); // the block is ready when that kernel is
        io->ready_mask |= 1u << k;
        ctx->side_busy[k] = true;
        rv->ap_busy = true;
      } else {
        if (V == 4) hipLaunchKernelGGL(fx_reverb_allpass_direct_kernel<4>, dim3(2 * wg_per_ch, grid_rows), blk, 0, ast, d);
        else hipLaunchKernelGGL(fx_reverb_allpass_direct_kernel<1>, dim3(2 * wg_per_ch, grid_rows), blk, 0, ast, d);
      }
      for (int i = 0; i < 2; ++i) std::swap(rv->geo.base[4 + i], rv->ap_alt[i]);
    } else if (ctx->seq_allpass) {
      hipLaunchKernelGGL(fx_reverb_allpass_kernel<32>, dim3(blocks_for(2 * (size_t)n)), blk, 0, st, io->d, n, frames, chs, rv->d_ring, g);
    } else { // chunks of one line length, parallel inside (kernels.h)
      const uint32_t T = std::max<uint32_t>(1, std::min<uint32_t>(64, 2 * n / 512));
      hipLaunchKernelGGL(fx_reverb_allpass_chunked_kernel, dim3((2 * n + T - 1) / T), dim3(kAllpassThreads), 0, st, io->d, n, frames, chs, rv->d_ring, g, T);
    }
  }
  for (uint32_t s = 0; s < count; ++s) fx_advance(run[s], frames);
  GHIP(ctx, hipGetLastError());
  if (rows) { io->sum_rows = wg_per_ch; io->sum_frames = frames; io->sums_valid = true; io->sums_on_ap = on_ap; }
  return 0;
}
// The kinds with feedback inside a block: the IIR filters, and delay lines shorter than the block.
static int fx_launch_serial(groove_fx* fx, groove_block* io, uint32_t frames, hipStream_t st) {
  groove_ctx* ctx = fx->ctx;
  const uint32_t n = fx->n;
  const size_t chs = (size_t)io->cap * n;
  const dim3 blk(kThreads), lanes_grid(blocks_for(2 * (size_t)n));
  // (64-frame chunks instead of 16 were measured for the serial IIR kernels: no change at any bank size — the walk is
  // bound by its dependent f64 chain, not by the chunk loads.)
  switch (fx->kind) {
    case GROOVE_FX_BIQUAD_LP12:
    case GROOVE_FX_BIQUAD_HP12:
    case GROOVE_FX_BIQUAD_BP12:
    case GROOVE_FX_BIQUAD_BS12:
    case GROOVE_FX_BIQUAD_AP12:
    case GROOVE_FX_BIQUAD_PEAK12:
    case GROOVE_FX_BIQUAD_LSHELF12:
    case GROOVE_FX_BIQUAD_HSHELF12:
      // few lane-channels: one wavefront each, frames over its lanes (fx_tp.h); many: one thread each, frames serial
      if (frames <= kTpMaxFrames && 2 * (size_t)n <= ctx->fx_tp_max_lanes && ctx->tp_max_voices) {
        if (2 * (size_t)n >= ctx->fx_tp_wide_min_lanes) hipLaunchKernelGGL(fx_biquad_tp_kernel<kFxTileWide>, dim3((n + kFxTileWide - 1) / kFxTileWide, 2), dim3(kFxTpThreads), 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
        else hipLaunchKernelGGL(fx_biquad_tp_kernel<kFxTile>, dim3((n + kFxTile - 1) / kFxTile, 2), dim3(kFxTpThreads), 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
      }
      else if (frames <= (uint32_t)(kBqSegs * kBqSegMax) && frames >= 16 && 2 * (size_t)n <= ctx->fx_seg_max_lanes) // four time segments per lane-channel
        hipLaunchKernelGGL(fx_biquad_seg_kernel, dim3((2 * n + 63) / 64), blk, 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
      else
        hipLaunchKernelGGL(fx_biquad_kernel<16>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
      break;
    case GROOVE_FX_BIQUAD_LP24:
      if (frames <= kTpMaxFrames && 4 * (size_t)n <= ctx->fx_tp_max_lanes && ctx->tp_max_voices) {
        if (2 * (size_t)n >= ctx->fx_tp_wide_min_lanes) hipLaunchKernelGGL(fx_lp24_tp_kernel<kFxTileWide>, dim3((n + kFxTileWide - 1) / kFxTileWide, 2), dim3(kFxTpThreads), 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
        else hipLaunchKernelGGL(fx_lp24_tp_kernel<kFxTile>, dim3((n + kFxTile - 1) / kFxTile, 2), dim3(kFxTpThreads), 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
      }
      else
        hipLaunchKernelGGL(fx_lp24_kernel<16>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_coef, fx->d_st, fx->d_wet);
      break;
    case GROOVE_FX_DELAY:
      if (fx->N >= 16) // chunked loads need every read of a chunk to precede its writes: N >= chunk
        hipLaunchKernelGGL(fx_delay_kernel<16>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->N, fx->w, fx->d_wet);
      else hipLaunchKernelGGL(fx_delay_kernel<1>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->N, fx->w, fx->d_wet);
      break;
    case GROOVE_FX_CHORUS: {
      const uint32_t nearest = fx->N - (fx->voices - 1) * fx->spacing;
      if (nearest >= 16 && (fx->voices == 1 || fx->spacing >= 16))
        hipLaunchKernelGGL(fx_chorus_kernel<16>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->N, fx->w, fx->voices, fx->spacing, fx->d_wet);
      else hipLaunchKernelGGL(fx_chorus_kernel<1>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->N, fx->w, fx->voices, fx->spacing, fx->d_wet);
      break;
    }
    case GROOVE_FX_REVERB: {
      if (fx_ap_settle(fx)) return 1;
      uint32_t shortest = fx->geo.N[0];
      for (int i = 1; i < 6; ++i) shortest = std::min(shortest, fx->geo.N[i]);
      if (shortest >= 8) hipLaunchKernelGGL(fx_reverb_kernel<8>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->geo, fx->d_fa, fx->d_wet);
      else hipLaunchKernelGGL(fx_reverb_kernel<1>, lanes_grid, blk, 0, st, io->d, n, frames, chs, fx->d_ring, fx->geo, fx->d_fa, fx->d_wet);
      break;
    }
    default: return fail(ctx, "groove_fx_process: unknown kind");
  }
  fx_advance(fx, frames);
  GHIP(ctx, hipGetLastError());
  return 0;
}
static int fx_chain_check(groove_fx* const* chain, uint32_t n_fx, groove_block* io, uint32_t frames, const char* who) {
  if (!io || (n_fx && !chain)) return fail(nullptr, std::string(who) + ": NULL argument");
  groove_ctx* ctx = io->ctx;
  for (uint32_t i = 0; i < n_fx; ++i) {
    if (!chain[i]) return fail(ctx, std::string(who) + ": NULL effect");
    if (chain[i]->ctx != ctx) return fail(ctx, std::string(who) + ": effect and block belong to different contexts");
    if (chain[i]->n != io->n) return fail(ctx, std::string(who) + ": block lanes != effect lanes");
    for (uint32_t j = 0; j < i; ++j)
      if (chain[j] == chain[i]) return fail(ctx, std::string(who) + ": the same effect twice in one chain");
  }
  if (frames > io->cap) return fail(ctx, std::string(who) + ": frames > block capacity");
  return 0;
}
int groove_fx_chain_process(groove_fx* const* chain, uint32_t n_fx, groove_block* io, uint32_t frames) {
  if (fx_chain_check(chain, n_fx, io, frames, "groove_fx_chain_process")) return 1;
  groove_ctx* ctx = io->ctx;
  if (frames == 0 || n_fx == 0) return 0;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (block_acquire(io) || block_normalise(io)) return 1; // effects carry per-lane parameters in the caller's order
  uint32_t last_stage = n_fx; // the last stage that launches anything (the Mixer is the identity)
  for (uint32_t i = 0; i < n_fx; ++i) if (chain[i]->kind != GROOVE_FX_MIXER) last_stage = i;
  groove_fx* run[kRunMaxStages];
  uint32_t count = 0;
  auto flush = [&](bool last) -> int {
    if (!count) return 0;
    const int rc = fx_launch_run(ctx, run, count, io, frames, ctx->stream, last);
    count = 0;
    return rc;
  };
  for (uint32_t i = 0; i < n_fx; ++i) {
    groove_fx* fx = chain[i];
    if (fx->kind == GROOVE_FX_MIXER) continue; // identity
    if (fx_acquire_ctx(fx)) return 1;
    io->sums_valid = false; // the block is transformed in place
    if (fx_run_capable(fx, frames)) {
      run[count++] = fx;
      if (count == kRunMaxStages || fx->kind == GROOVE_FX_REVERB) { if (flush(i == last_stage)) return 1; }
    } else {
      if (flush(false)) return 1;
      if (fx_launch_serial(fx, io, frames, ctx->stream)) return 1;
    }
  }
  return flush(true);
}
// The LEADING stages of a chain that walk the block's frames one lane-channel per thread (the IIR filters, delay lines
// shorter than the block) on the side stream that carries `io`'s pending asynchronous render, right behind it: these
// are latency-bound kernels of a few hundred wavefronts, and behind the render of block b+1 they run beside the wide,
// HBM-bound stages of block b on the ctx stream instead of in front of them (config #3: the BiQuad took 29 us of the
// ctx stream's 82 us per block).  *n_done = how many stages were taken; the caller passes the rest of the chain to
// groove_fx_chain_process when it gets to the block.  Nothing is taken (and 0 returned) when the block has no pending
// render on exactly one side stream, or when the chain does not start with such a stage.
int groove_fx_chain_process_async(groove_fx* const* chain, uint32_t n_fx, groove_block* io, uint32_t frames, uint32_t* n_done) {
  if (!n_done) return fail(nullptr, "groove_fx_chain_process_async: n_done is NULL");
  *n_done = 0;
  if (fx_chain_check(chain, n_fx, io, frames, "groove_fx_chain_process_async")) return 1;
  groove_ctx* ctx = io->ctx;
  if (frames == 0 || n_fx == 0) return 0;
  int k = -1;
  for (int j = 0; j < kSideStreams; ++j)
    if (io->ready_mask & (1u << j)) k = (k == -1) ? j : -2;
  if (k < 0) return 0; // no pending render, or one spread over several kind streams
  if (io->order) return 0; // the block is in the library's lane order: the ctx stream produces the caller's first
  GHIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = side_stream_of(ctx, k);
  uint32_t done = 0;
  for (uint32_t i = 0; i < n_fx; ++i) {
    groove_fx* fx = chain[i];
    if (fx->kind == GROOVE_FX_MIXER) { ++done; continue; }
    // Only the IIR filters: they walk the frames serially whatever the block length, so an effect is EITHER always ahead
    // of the ctx stream's walk OR never.  (A delay line is a serial kernel for blocks longer than the line and a stage of the
    // fused run otherwise: taken here for some blocks and left to the ctx stream for others, its blocks would be
    // submitted out of order — found by the ragged-block test.)
    if (!fx_is_iir(fx->kind)) break;
    if (fx->last_side != k) { // the stream that used the effect last: the ctx stream (or another side stream)
      if (!fx->ev_done) GHIP(ctx, hipEventCreateWithFlags(&fx->ev_done, kSyncEventFlags));
      if (fx->last_side < 0) GHIP(ctx, hipEventRecord(fx->ev_done, ctx->stream));
      else if (fx_done_event(fx)) return 1;
      GHIP(ctx, hipStreamWaitEvent(st, fx->ev_done, 0));
    }
    io->sums_valid = false;
    if (fx_launch_serial(fx, io, frames, st)) return 1;
    GHIP(ctx, hipEventRecord(fx->ev_done, st));
    fx->done_recorded = true;
    fx->last_side = k;
    ++done;
  }
  if (done) {
    GHIP(ctx, hipEventRecord(io->ev_ready[k], st)); // the block is ready when these stages are
    ctx->side_busy[k] = true;
  }
  *n_done = done;
  return 0;
}
// groove_bank_render_async + groove_fx_chain_process_async in one call, which lets the library FUSE the chain's first
// stage into the render kernel when it can (a 12 dB BiQuad behind a bank that renders time-parallel: welsh_tp.h HEAD_BQ).
int groove_bank_render_chain_async(groove_bank* b, uint32_t frames, groove_block* out, groove_fx* const* chain, uint32_t n_fx, uint32_t* n_done) {
  if (!n_done) return fail(nullptr, "groove_bank_render_chain_async: n_done is NULL");
  *n_done = 0;
  if (!b || !out) return fail(nullptr, "groove_bank_render_chain_async: NULL argument");
  if (fx_chain_check(chain, n_fx, out, frames, "groove_bank_render_chain_async")) return 1;
  uint32_t first = 0; // the first stage that does anything (the Mixer is the identity)
  while (first < n_fx && chain[first]->kind == GROOVE_FX_MIXER) ++first;
  groove_fx* head = first < n_fx ? chain[first] : nullptr;
  bool fused = false;
  if (render_async_impl(b, frames, out, head, &fused, head != nullptr)) return 1;
  if (frames == 0) return 0;
  uint32_t taken = 0;
  if (fused) taken = first + 1;
  uint32_t more = 0;
  if (taken < n_fx && groove_fx_chain_process_async(chain + taken, n_fx - taken, out, frames, &more)) return 1;
  *n_done = taken + more;
  return 0;
}
int groove_fx_process(groove_fx* fx, groove_block* io, uint32_t frames) {
  if (!fx || !io) return fail(nullptr, "groove_fx_process: NULL argument");
  return groove_fx_chain_process(&fx, 1, io, frames);
}
int groove_fx_set_params(groove_fx* fx, const groove_fx_params* p, uint32_t n) {
  if (!fx || !p) return fail(nullptr, "groove_fx_set_params: NULL argument");
  groove_ctx* ctx = fx->ctx;
  if (n != fx->n) return fail(ctx, "groove_fx_set_params: n != lanes");
  std::vector<groove_fx_params> old = fx->p;
  fx->p.assign(p, p + n);
  const groove_fx_params &a = old[0], &b = fx->p[0];
  if (fx_check_uniform(fx) ||
      ((fx->kind == GROOVE_FX_CHORUS || fx->kind == GROOVE_FX_DELAY) && (a.delay_seconds != b.delay_seconds || (fx->kind == GROOVE_FX_CHORUS && a.voices != b.voices))) ||
      (fx->kind == GROOVE_FX_REVERB && a.reverb_seconds != b.reverb_seconds)) {
    fx->p = old;
    return fail(ctx, "groove_fx_set_params: delay-line geometry cannot change after creation");
  }
  if (fx_acquire_ctx(fx) || fx_ap_settle(fx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  return fx_upload_params(fx);
}
int groove_fx_set_param(groove_fx* fx, uint32_t lane, uint32_t control_index, double value01) {
  if (!fx) return fail(nullptr, "groove_fx_set_param: fx is NULL");
  groove_ctx* ctx = fx->ctx;
  if (lane != GROOVE_ALL_VOICES && lane >= fx->n) return fail(ctx, "groove_fx_set_param: lane out of range");
  const double v = value01 < 0.0 ? 0.0 : (value01 > 1.0 ? 1.0 : value01);
  const uint32_t lo = lane == GROOVE_ALL_VOICES ? 0 : lane, hi = lane == GROOVE_ALL_VOICES ? fx->n : lane + 1;
  for (uint32_t i = lo; i < hi; ++i) {
    groove_fx_params& p = fx->p[i];
    switch (control_index) {
      case GROOVE_CTL_FX_CEILING: p.ceiling = (float)v; break;
      case GROOVE_CTL_FX_BITS: p.bits = (uint32_t)(v * 16.0); break;
      case GROOVE_CTL_FX_CUTOFF: p.cutoff_hz = (float)percent_to_frequency_h(v); break;
      case GROOVE_CTL_FX_Q: p.q = (float)(v * v * 10.0 + 0.707); break; /* denormalize_q */
      case GROOVE_CTL_FX_PASSBAND_RIPPLE: p.passband_ripple = (float)(v * v * 10.0 + 0.707); break;
      case GROOVE_CTL_FX_ATTENUATION: p.attenuation = (float)v; break;
      case GROOVE_CTL_FX_WET: p.wet = (float)v; break;
      case GROOVE_CTL_FX_THRESHOLD: p.limit_min = (float)v; break; /* Compressor threshold (limit_min doubles as it: groove_types.h) */
      default: return fail(ctx, "groove_fx_set_param: unknown control index");
    }
  }
  if (fx_acquire_ctx(fx) || fx_ap_settle(fx)) return 1;
  GHIP(ctx, ctx_wait(ctx));
  return fx_upload_params(fx);
}

// ============================================================================ mix bus
int groove_mix(groove_ctx* ctx, groove_block* const* blocks, uint32_t n_blocks, uint32_t frames,
               float* bus_dev, int accumulate) {
  if (!ctx || !bus_dev) return fail(ctx, "groove_mix: NULL argument");
  if (bus_flush(ctx)) return 1;
  if (n_blocks && !blocks) return fail(ctx, "groove_mix: blocks is NULL");
  if (frames == 0) return 0;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (n_blocks == 0) { // nothing patched: silence (orchestrator.rs:1452-1455)
    if (!accumulate) GHIP(ctx, hipMemsetAsync(bus_dev, 0, (size_t)frames * 8, ctx->stream));
    return 0;
  }
  for (uint32_t i = 0; i < n_blocks; ++i) {
    if (!blocks[i]) return fail(ctx, "groove_mix: NULL block");
    if (frames > blocks[i]->cap) return fail(ctx, "groove_mix: frames > block capacity");
    if (block_acquire(blocks[i])) return 1;
    groove_block* blk = blocks[i];
    if (blk->sums_valid && blk->sum_frames == frames) { // the render left the block's lane sums: reduce those rows
      // a block of the render-ahead rotation (it has its events): this reduction is normally its last consumer before the host
      // releases it, so the block's "free" event is bound to the reduction's last kernel and the release records nothing
      const bool mark = blk->ev_free && ctx->bind_events;
      if (reduce_rows(ctx, blk->sums_on_ap ? blk->d_sums_ap[blk->ap_flip] : blk->d_sums, blk->sum_rows, frames, bus_dev, accumulate || i > 0, mark ? blk->ev_free : nullptr)) return 1;
      if (blk->sums_on_ap && ap_follow(ctx)) return 1; // (that stream's later all-passes rewrite the buffer)
      blk->free_marked = mark;
    } else if (mix_one(ctx, blocks[i], frames, bus_dev, accumulate || i > 0)) return 1;
  }
  return 0;
}
// groove_mix for ONE block whose lane sums are valid (a render or an effect chain has just left them), with the reduction of
// those few rows DEFERRED: the next fused effect-chain launch on the ctx stream (fx_run_kernel), the next deferred render, or
// bus_flush puts them on the bus.  Anything else: groove_mix.
int groove_mix_deferred(groove_ctx* ctx, groove_block* blk, uint32_t frames, float* bus_dev, int accumulate) {
  if (!ctx || !bus_dev || !blk) return fail(ctx, "groove_mix_deferred: NULL argument");
  if (frames == 0) return 0;
  if (blk->ctx != ctx) return fail(ctx, "groove_mix_deferred: the block belongs to another ctx");
  if (!(blk->sums_valid && blk->sum_frames == frames && blk->sum_rows <= 2048 && frames <= 4096)) return groove_mix(ctx, &blk, 1, frames, bus_dev, accumulate);
  GHIP(ctx, hipSetDevice(ctx->device));
  if (blk->sums_on_ap && ctx->fx_ap_stream >= 0 && blk->ready_mask == (1u << ctx->fx_ap_stream)) {
    // The rows are being written on the all-pass stream: that stream's NEXT all-pass launch sums them (ordered behind their
    // writer without a wait; the ctx stream is not held up for the block).  Pending rows of any other kind go first, and the
    // all-pass stream behind them: the order of a bus's sums is the order of the calls.
    if (ctx->deferred.rows || ctx->deferred_ap.rows || !ctx->paced_order.empty()) { if (bus_flush(ctx) || ap_follow(ctx)) return 1; }
    ctx->deferred_ap.rows = blk->d_sums_ap[blk->ap_flip]; ctx->deferred_ap.bus = bus_dev; ctx->deferred_ap.n_rows = blk->sum_rows; ctx->deferred_ap.frames = frames; ctx->deferred_ap.accumulate = accumulate;
    blk->ap_flip ^= 1; // the block's next chain writes the other buffer: this one is read one launch from now
    blk->sums_valid = false; blk->sums_on_ap = false;
    return 0;
  }
  if (bus_flush(ctx)) return 1; // an earlier pending block nobody carried: its own reduction launch, first (order of the bus's sums)
  if (block_acquire(blk)) return 1;
  if (blk->sums_on_ap) { const int rc = groove_mix(ctx, &blk, 1, frames, bus_dev, accumulate); return rc; } // (not this path's buffers)
  ctx->deferred.rows = blk->d_sums; ctx->deferred.bus = bus_dev; ctx->deferred.n_rows = blk->sum_rows; ctx->deferred.frames = frames; ctx->deferred.accumulate = accumulate;
  // the rows leave the block: whatever writes the block's lane sums next (its next render may run on another stream) cannot touch them
  ctx->deferred.owned_cap = blk->sums_cap;
  blk->d_sums = nullptr; blk->sums_cap = 0; blk->sums_valid = false;
  // a spare at least as large as the buffer that left (a smaller one would make block_sums free and reallocate — device-wide
  // synchronisations inside the paced walk — until every buffer of the rotation had grown): the smallest that fits
  int pick = -1;
  for (int i = 0; i < (int)ctx->spare_sums.size(); ++i)
    if (ctx->spare_sums[i].second >= ctx->deferred.owned_cap && (pick < 0 || ctx->spare_sums[i].second < ctx->spare_sums[pick].second)) pick = i;
  if (pick >= 0) { blk->d_sums = ctx->spare_sums[pick].first; blk->sums_cap = ctx->spare_sums[pick].second; ctx->spare_sums.erase(ctx->spare_sums.begin() + pick); }
  return 0;
}
// HOST PACING.  A wait for another queue's event costs the waiting stream 7 - 9 us of its timeline whether or not the event has
// completed by the time the stream gets there (docs/STREAMS.md item 13), but the runtime drops a wait whose event is already
// complete when the call is made.  An offline host has nothing else to do: it can wait for the event itself, and the streams
// then carry no wait packets at all.
int groove_block_wait_ready(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_wait_ready: block is NULL");
  groove_ctx* ctx = b->ctx;
  for (int k = 0; k < kSideStreams; ++k)
    if (b->ready_mask & (1u << k)) GHIP(ctx, wait_deadline(ctx, nullptr, b->ev_ready[k], "groove_block_wait_ready"));
  b->ready_mask = 0; // complete: what the host submits from now on is ordered behind it without a device-side wait
  return 0;
}
int groove_block_wait_released(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_wait_released: block is NULL");
  if (b->released && b->ev_free) {
    GHIP(b->ctx, wait_deadline(b->ctx, nullptr, b->ev_free, "groove_block_wait_released"));
    // a release recorded on the all-pass stream sits behind the block's last kernel there: the host has seen that complete
    if (b->ctx->fx_ap_stream >= 0) b->ready_mask &= ~(1u << b->ctx->fx_ap_stream);
  }
  return 0;
}
int groove_block_accumulate(groove_block* dst, groove_block* src, uint32_t frames, int accumulate) {
  if (!dst || !src) return fail(nullptr, "groove_block_accumulate: NULL argument");
  groove_ctx* ctx = dst->ctx;
  if (frames > dst->cap || frames > src->cap) return fail(ctx, "groove_block_accumulate: frames > block capacity");
  if (frames == 0) return 0;
  GHIP(ctx, hipSetDevice(ctx->device));
  if (block_acquire(dst) || block_acquire(src)) return 1;
  dst->sums_valid = false;
  if (src->n == dst->n) {
    if (block_normalise(src)) return 1;                 // element-wise: both in the caller's lane order
    if (accumulate ? block_normalise(dst) : 0) return 1;
    if (!accumulate) dst->order.reset();
    const size_t total = (size_t)2 * frames * src->n;
    const uint32_t g = (uint32_t)std::min<size_t>(blocks_for(total), 256 * 16);
    hipLaunchKernelGGL(block_add_kernel, dim3(g), dim3(kThreads), 0, ctx->stream, dst->d, (size_t)dst->cap * dst->n,
                       src->d, (size_t)src->cap * src->n, src->n, frames, accumulate);
    GHIP(ctx, hipGetLastError());
    return 0;
  }
  if (dst->n == 1) { dst->order.reset(); return mix_one(ctx, src, frames, dst->d, accumulate, (size_t)dst->cap); } // a lane sum has no order
  return fail(ctx, "groove_block_accumulate: lane counts differ and the destination is not a 1-lane block");
}
int groove_block_zero(groove_block* b) {
  if (!b) return fail(nullptr, "groove_block_zero: NULL argument");
  if (block_acquire(b)) return 1;
  b->sums_valid = false;
  b->order.reset();
  GHIP(b->ctx, hipMemsetAsync(b->d, 0, (size_t)2 * b->cap * b->n * 4, b->ctx->stream));
  return 0;
}
int groove_bus_create(groove_ctx* ctx, size_t frames, float** out_dev) {
  if (!ctx || !out_dev) return fail(ctx, "groove_bus_create: NULL argument");
  GHIP(ctx, hipSetDevice(ctx->device));
  GHIP(ctx, hipMalloc(out_dev, std::max<size_t>(frames, 1) * 8));
  // zeroed ON THE CTX STREAM: hipMemset runs on the null stream, which the (non-blocking) ctx stream does
  // not wait for, and may land after the first kernels that write the bus
  GHIP(ctx, hipMemsetAsync(*out_dev, 0, std::max<size_t>(frames, 1) * 8, ctx->stream));
  return 0;
}
int groove_bus_destroy(groove_ctx* ctx, float* bus_dev) {
  if (!ctx) return fail(nullptr, "groove_bus_destroy: ctx is NULL");
  GHIP(ctx, ctx_wait(ctx));
  GHIP(ctx, hipFree(bus_dev));
  return 0;
}
int groove_bus_zero(groove_ctx* ctx, float* bus_dev, size_t frames) {
  if (!ctx || !bus_dev) return fail(ctx, "groove_bus_zero: NULL argument");
  if (bus_flush(ctx)) return 1;
  GHIP(ctx, hipMemsetAsync(bus_dev, 0, frames * 8, ctx->stream));
  return 0;
}
int groove_download(groove_ctx* ctx, const float* dev, float* host, size_t n_floats) {
  if (!ctx || !dev || !host) return fail(ctx, "groove_download: NULL argument");
  GHIP(ctx, ctx_wait(ctx, "copy on the ctx stream"));
  GHIP(ctx, hipMemcpyAsync(host, dev, n_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}
int groove_upload(groove_ctx* ctx, float* dev, const float* host, size_t n_floats) {
  if (!ctx || !dev || !host) return fail(ctx, "groove_upload: NULL argument");
  GHIP(ctx, ctx_wait(ctx, "copy on the ctx stream"));
  GHIP(ctx, hipMemcpyAsync(dev, host, n_floats * 4, hipMemcpyHostToDevice, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}
int groove_bus_to_i16(groove_ctx* ctx, const float* bus_dev, size_t frames, int16_t* host_out) {
  if (!ctx || !bus_dev || !host_out) return fail(ctx, "groove_bus_to_i16: NULL argument");
  if (bus_flush(ctx)) return 1;
  const size_t count = frames * 2;
  if (count == 0) return 0;
  if (ctx->i16_cap < count) {
    if (ctx->d_i16) GHIP(ctx, hipFree(ctx->d_i16));
    GHIP(ctx, hipMalloc(&ctx->d_i16, count * 2));
    ctx->i16_cap = count;
  }
  hipLaunchKernelGGL(bus_to_i16_kernel, dim3(blocks_for(count)), dim3(kThreads), 0, ctx->stream, bus_dev, count, ctx->d_i16);
  GHIP(ctx, hipGetLastError());
  GHIP(ctx, ctx_wait(ctx, "copy on the ctx stream"));
  GHIP(ctx, hipMemcpyAsync(host_out, ctx->d_i16, count * 2, hipMemcpyDeviceToHost, ctx->stream));
  GHIP(ctx, ctx_wait(ctx));
  return 0;
}

// ============================================================================ multi-GPU
int groove_comm_unique_id(groove_ctx* ctx, uint8_t id_out[128]) { // ctx may be NULL (groove_init_comm wants the id before any ctx exists)
  if (!id_out) return fail(ctx, "groove_comm_unique_id: NULL argument");
  if (rccl_open(ctx)) return 1;
  auto f = (nccl_get_uid_fn)dlsym(g_rccl, "ncclGetUniqueId");
  if (!f) return fail(ctx, "ncclGetUniqueId not found");
  ncclUniqueId id;
  const ncclResult_t rc = f(&id);
  if (rc != ncclSuccess) return fail(ctx, "ncclGetUniqueId failed");
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}
static int comm_init_rank(groove_ctx* ctx, const uint8_t id[128], int rank, int world_size) {
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(ctx, "groove_comm_init: bad rank/world");
  if (ctx->comm) return fail(ctx, "groove_comm_init: this ctx already has a communicator");
  if (rccl_open(ctx)) return 1;
  GHIP(ctx, hipSetDevice(ctx->device));
  auto f = (nccl_init_rank_fn)dlsym(ctx->rccl, "ncclCommInitRank");
  if (!f) return fail(ctx, "ncclCommInitRank not found");
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = f(&comm, world_size, uid, rank);
  if (rc != ncclSuccess) {
    auto es = (nccl_errstr_fn)dlsym(ctx->rccl, "ncclGetErrorString");
    return fail(ctx, std::string("ncclCommInitRank failed: ") + (es ? es(rc) : "?"));
  }
  ctx->comm = comm; ctx->rank = rank; ctx->world = world_size;
  return 0;
}
int groove_comm_init(groove_ctx* ctx, const uint8_t id[128], int rank, int world_size) {
  if (!ctx || !id) return fail(ctx, "groove_comm_init: NULL argument");
  return comm_init_rank(ctx, id, rank, world_size);
}
int groove_comm_ranks(groove_ctx* ctx, int* out_ranks) {
  if (!ctx || !out_ranks) return fail(ctx, "groove_comm_ranks: NULL argument");
  *out_ranks = 1;
  if (!ctx->comm) return 0; // no communicator: one rank
  auto f = (nccl_count_fn)dlsym(ctx->rccl, "ncclCommCount");
  if (!f) return fail(ctx, "ncclCommCount not found");
  int count = 0;
  const ncclResult_t rc = f((ncclComm_t)ctx->comm, &count);
  if (rc != ncclSuccess) return fail(ctx, "ncclCommCount failed");
  *out_ranks = count;
  return 0;
}
int groove_comm_destroy(groove_ctx* ctx) {
  if (!ctx || !ctx->comm) return 0;
  auto f = (nccl_destroy_fn)dlsym(ctx->rccl, "ncclCommDestroy");
  if (f) f((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  return 0;
}
int groove_bus_reduce(groove_ctx* ctx, float* bus_dev, size_t frames_total, int root) {
  if (!ctx || !bus_dev) return fail(ctx, "groove_bus_reduce: NULL argument");
  if (bus_flush(ctx)) return 1;
  if (ctx->world == 1 && !ctx->comm) return 0; // single GPU: the bus is already complete
  if (!ctx->comm) return fail(ctx, "groove_bus_reduce: communicator not initialised");
  auto f = (nccl_reduce_fn)dlsym(ctx->rccl, "ncclReduce");
  if (!f) return fail(ctx, "ncclReduce not found");
  // in place on the root
  const ncclResult_t rc = f(bus_dev, bus_dev, frames_total * 2, ncclFloat32, ncclSum, root, (ncclComm_t)ctx->comm, ctx->stream);
  if (rc != ncclSuccess) {
    auto es = (nccl_errstr_fn)dlsym(ctx->rccl, "ncclGetErrorString");
    return fail(ctx, std::string("ncclReduce failed: ") + (es ? es(rc) : "?"));
  }
  return 0;
}

} // extern "C"
