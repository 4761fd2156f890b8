/* groove_hip.h — C ABI of libgroove_hip.so, the MI355X (gfx950) render path that
 * drops in behind the reference's instrument/effect trait surface.
 *
 * The reference has no FFI of its own: its "plugin" boundary is the
 * set of Rust traits an entity implements (`ensnare_core::traits`, glued by
 * /root/reference/proc-macros/src/entity.rs:29-139).  Each entry point below names the
 * trait method(s) it replaces and the call site in the reference that drives it.  All
 * citations are relative to /root/reference.  INTEGRATION.md shows the Rust `extern "C"`
 * block and the `entities`-crate wrappers a maintainer would add.
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on failure; groove_last_error() gives the
 *     message (audio-path trait methods are infallible in the reference; graph edits
 *     return anyhow::Result, orchestrator.rs:263-304 — the library never aborts);
 *   - the caller owns host memory; the library owns device memory behind opaque handles;
 *   - handles are not thread-safe: one HIP stream per ctx, matching the reference's
 *     single-threaded audio path (orchestrator.rs:367-470).  A ctx and everything created
 *     from it belong to one thread at a time; different contexts are independent and may be
 *     driven from different threads at once (tests/test_gpu_deferred.py::test_two_contexts_*);
 *   - device blocks are planar, frame-major fp32:  block[ch][frame][voice], ch 0 = left,
 *     so one wavefront's store is 64 consecutive floats; the bus is bus[frame][2];
 *   - note events and parameter changes take effect at the next block start, exactly the
 *     granularity of Orchestrator::tick (orchestrator.rs:856-859: handle_work once per
 *     tick, then gather_audio over the whole buffer).
 */
#ifndef GROOVE_HIP_H
#define GROOVE_HIP_H

#include "groove_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct groove_ctx groove_ctx;     /* one device + one stream                       */
typedef struct groove_bank groove_bank;   /* an instrument: N homogeneous voices, one lane each */
typedef struct groove_fx groove_fx;       /* an effect applied per lane to an N-lane block  */
typedef struct groove_block groove_block; /* device stereo block [2][frames_cap][n] fp32    */

/* ---- context ------------------------------------------------------------------------ */
/* Orchestrator::new_with + device selection (orchestrator.rs:522-568). */
int groove_init(int device_ordinal, groove_ctx** out);
/* groove_init for one rank of a multi-GPU job: the rank's RCCL communicator is created FIRST (id from
 * groove_comm_unique_id(NULL, ...) on rank 0, broadcast by the launcher), the library's own streams after it, so
 * that RCCL's internal streams do not land between them.  Equivalent to groove_init followed by
 * groove_comm_init otherwise — which is the faster order on this part (measured, DESIGN.md section 6: the communicator first
 * costs the million-voice path 19 %) and the one bench.py's ranks use.  GROOVE_SAFE_STREAMS=1 in the environment selects the conservative stream layout (one
 * priority, four streams in all) for either form. */
int groove_init_comm(int device_ordinal, const uint8_t id[128], int rank, int world_size, groove_ctx** out);
void groove_shutdown(groove_ctx* ctx);
/* ctx may be NULL: returns the last error of the calling thread. */
const char* groove_last_error(groove_ctx* ctx);
/* Use the caller's hipStream_t (e.g. a torch stream) instead of the ctx's own. */
int groove_set_stream(groove_ctx* ctx, void* hip_stream);
/* Waits for everything submitted so far (every stream of the library).  Like every blocking call of this API
 * (downloads, groove_bus_to_i16, groove_event_elapsed_ms ...) it polls with a DEADLINE: if the work has not completed
 * after groove_sync_timeout_ms() milliseconds the call returns non-zero and groove_last_error() names the streams
 * that are still busy — a stalled kernel surfaces as an error, never as a hang and never as an abort.  Nothing is
 * cancelled; the caller may wait again or tear the process down.  Default 60,000 ms (GROOVE_SYNC_TIMEOUT_MS in the
 * environment at groove_init), 0 = wait for ever.  No reference counterpart (the reference's audio path is
 * synchronous CPU code, orchestrator.rs:367-470). */
int groove_synchronize(groove_ctx* ctx);
int groove_set_sync_timeout_ms(groove_ctx* ctx, uint32_t ms);
uint32_t groove_sync_timeout_ms(groove_ctx* ctx);
/* Test / diagnosis hooks.  groove_debug_spin: one idle kernel that occupies a library stream for `ms` milliseconds
 * (side_stream = -1: the ctx stream; 0..5 kind streams; 6.. bank streams) — how the tests block a stream on purpose.
 * groove_debug_info: a JSON object with the stream layout of this ctx and `zero_segments` — how often a Welsh kernel's wave
 * found that its ACTIVE lanes had zero frames to their next envelope boundary (csrc/diag.h: the counted assertion behind
 * DESIGN.md section 7; it must read 0, and the GPU tests, smoke() and bench.py require that).  It waits for the ctx stream;
 * call groove_synchronize first for a figure that includes the side streams' latest kernels.  `host_waits`,
 * `host_waits_blocked`, `host_wait_ms`: how often the library's own waits (groove_synchronize, the paced calls, ...) ran, how
 * many of them found the device still busy, and for how long in total the calling thread was blocked in them — a paced walk
 * that is never blocked is bound by the host's submission, not by the device. */
int groove_debug_spin(groove_ctx* ctx, int side_stream, uint32_t ms);
int groove_debug_info(groove_ctx* ctx, char* out, size_t cap);
/* Configurable::update_sample_rate fan-out (orchestrator.rs:125-127, 1019-1022, 1389-1394).
 * Re-derives every bank/effect created on this ctx and resets their state. */
int groove_update_sample_rate(groove_ctx* ctx, uint32_t hz);
uint32_t groove_sample_rate(groove_ctx* ctx);
/* Tuning: Welsh banks of up to this many voices render a block (of up to 256 frames) TIME-PARALLEL, one
 * wavefront per voice with the frames spread over its lanes (csrc/welsh_tp.h), instead of one voice per
 * lane walking the frames serially — the form that is latency-bound below ~250,000 voices.  Same results
 * to f64 rounding of the filter.  0 = always the serial kernels.  Default 16,384 (GROOVE_TP_MAX_VOICES in
 * the environment overrides it at groove_init).  No reference counterpart. */
int groove_set_time_parallel_max_voices(groove_ctx* ctx, uint32_t max_voices);
uint32_t groove_time_parallel_max_voices(groove_ctx* ctx);
/* Tuning: time-parallel Welsh banks of at least this many voices, whose adjacent voices (2i, 2i + 1) share a patch, render
 * TWO voices per wavefront (32 lanes x 8 frames each) instead of one (64 lanes x 4 frames): half the wavefronts, 25 % less
 * issue per voice, the right trade once the one-voice form needs more wavefronts than the SIMDs hold at once; such banks stay
 * time-parallel up to 11/8 of groove_set_time_parallel_max_voices' limit (22,528 voices by default).  Default 3,073;
 * 0 = never; 1 = whenever the pairs allow (tests).
 * No reference counterpart. */
int groove_set_time_parallel_pair_min_voices(groove_ctx* ctx, uint32_t min_voices);
uint32_t groove_time_parallel_pair_min_voices(groove_ctx* ctx);
/* Tuning: Welsh banks of at least this many (virtual) wavefronts — 64 voices each; default 3,800 = ~243,000 voices — run
 * one kernel per base kind and pipeline consecutive fused blocks; smaller ones take one launch for all kinds.  1 forces
 * the per-kind pipelined form for every size (tests and bench.py's parity sample use it to run the kernels of the
 * million-voice path on a small bank).  GROOVE_PIPELINE_MIN_WAVES in the environment sets it at groove_init. */
int groove_set_pipeline_min_waves(groove_ctx* ctx, uint32_t waves);
uint32_t groove_pipeline_min_waves(groove_ctx* ctx);
/* Which look-aheads the wave-uniform Welsh kernels may use (csrc/kernels.h): bit 0 — a wave whose live voices share the filter
 * envelope's stage computes 64 frames' filter coefficients in one pass, lane = frame (bit-identical to the per-lane retune); bit 1 —
 * a wave whose live voices share the LFO's phase evaluates what the LFO does to the oscillators (pitch / pulse-width routing) the same
 * way, exactly, where each lane would advance recurrences (within 1e-6 of them).  Default 3; GROOVE_LOOK_AHEAD in the environment
 * sets it at groove_init.  Tests render the same bank with and without.  Bit 2 (tests): count the waves that take the FAST copy of
 * their block body (groove_debug_info: fast_waves). */
int groove_set_look_ahead(groove_ctx* ctx, uint32_t bits);
uint32_t groove_look_ahead(groove_ctx* ctx);
/* Tuning, for a render-ahead walk of blocks through an effect chain that ENDS in a reverb (config #3): on = the reverb's two
 * all-passes — the chain's last kernel, which nothing on the ctx stream reads behind — are launched on a side stream of the
 * library, so that the NEXT block's fused run follows this block's run directly and the all-passes overlap it (0.0489 -> 0.044 ms
 * per block of config #3).  Semantics are unchanged: the block is "pending" like one a groove_bank_render_async is filling
 * (every call that reads it orders itself behind the kernel; groove_block_wait_ready / groove_block_wait_released wait for it);
 * its lane sums reach the bus through groove_mix, or — deferred — through groove_mix_deferred, in which case the next all-pass
 * launch (or any flush point) sums them.  Only blocks that have been through groove_block_release or an asynchronous render
 * (they have their events) take the side stream; everything else runs as before.  Results are bit-identical either way.
 * Default off.  No reference counterpart (the reference transforms one sample at a time, orchestrator.rs:446-454). */
int groove_set_fx_allpass_stream(groove_ctx* ctx, int on);
int groove_fx_allpass_stream(groove_ctx* ctx);
/* Tuning: Welsh banks of up to this many (virtual) wavefronts that are too big for the time-parallel form render ROLE-SPLIT:
 * four wavefronts per 64 voices — envelopes + LFO, oscillators, cutoff tangent + coefficient quotients, filter + gains — pipelined
 * over the block's frames through LDS, so that a bank which cannot fill the chip with voices fills it with the parts of a voice's
 * frame (csrc/welsh_split.h).  Same results bit for bit as the serial kernels run without their LFO look-ahead (groove_set_look_ahead),
 * within 2e-6 of them with.  Default 1,024 (65,536 voices: one workgroup per CU; a two-role form of the same kernel for banks of up
 * to twice that is off by default since the serial kernels' FAST bodies outrun it: GROOVE_SPLIT2_MAX_WAVES=2048 in the environment
 * brings it back); 0 = never.  No reference counterpart. */
int groove_set_split_max_waves(groove_ctx* ctx, uint32_t waves);
uint32_t groove_split_max_waves(groove_ctx* ctx);
/* HIP events on the ctx stream, for measurement (bench.py): create / record / elapsed. */
int groove_event_create(groove_ctx* ctx, void** out_event);
int groove_event_destroy(groove_ctx* ctx, void* event);
int groove_event_record(groove_ctx* ctx, void* event);
int groove_event_elapsed_ms(groove_ctx* ctx, void* start, void* stop, float* out_ms);

/* ---- blocks --------------------------------------------------------------------------- */
/* The buffer a `generate_batch_values(&mut [StereoSample])` call fills
 * (entities/src/instruments/metronome.rs:23-35), for n lanes at once. */
int groove_block_create(groove_ctx* ctx, uint32_t n, uint32_t frames_cap, groove_block** out);
int groove_block_destroy(groove_block* b);
/* The raw device pointer.  Handing it out invalidates the lane sums the last render left with the block (groove_mix
 * then reads the block itself); a caller that keeps the pointer and writes the block again later calls
 * groove_block_mark_dirty (or groove_block_acquire) before the next groove_mix. */
float* groove_block_device_ptr(groove_block* b);
int groove_block_mark_dirty(groove_block* b);
uint32_t groove_block_lanes(groove_block* b);
uint32_t groove_block_frames_cap(groove_block* b);
/* host [2][frames][n] fp32 <-> device */
int groove_block_upload(groove_block* b, const float* host, uint32_t frames);
int groove_block_download(groove_block* b, float* host, uint32_t frames);

/* dst (+)= src.  Equal lane counts: element-wise.  dst with ONE lane: the lanes of src are summed
 * (a Synthesizer summing its voice store; an effect summing its sources, orchestrator.rs:438-457). */
int groove_block_accumulate(groove_block* dst, groove_block* src, uint32_t frames, int accumulate);
int groove_block_zero(groove_block* b);

/* ---- instruments (Ticks + Generates<StereoSample> + HandlesMidi + Controllable) --------- */
/* WelshSynth::new_with(&WelshSynthParams) (settings/src/instruments.rs:71-76) for n voices. */
int groove_welsh_create(groove_ctx* ctx, const groove_welsh_params* p, uint32_t n, groove_bank** out);
/* FmSynth::new_with(&FmSynthParams) (settings/src/instruments.rs:89-92). */
int groove_fm_create(groove_ctx* ctx, const groove_fm_params* p, uint32_t n, groove_bank** out);
/* Sampler::new_with / Drumkit::new_with (settings/src/instruments.rs:81-88); the shared
 * mono sample bank is uploaded once and stays resident. */
int groove_sampler_create(groove_ctx* ctx, const float* bank_pcm, uint64_t bank_frames,
                          const groove_sample_desc* descs, uint32_t n_samples,
                          const groove_sampler_params* p, uint32_t n, groove_bank** out);
int groove_bank_destroy(groove_bank* bank);
uint32_t groove_bank_voices(groove_bank* bank);
/* HandlesMidi::handle_midi_message → PlaysNotes::note_on/note_off (orchestrator.rs:737-739;
 * settings/src/patches.rs:928, 821).  Applied in order, at the start of the next render. */
int groove_bank_note_events(groove_bank* bank, const groove_note_event* ev, uint32_t n_ev);
/* Controllable::control_set_param_by_index (proc-macros/src/control.rs:211-226); value01 is a
 * ControlValue in 0..1 (orchestration/src/lib.rs:43-47).  voice = GROOVE_ALL_VOICES for all. */
int groove_bank_set_param(groove_bank* bank, uint32_t voice, uint32_t control_index, double value01);
/* Ticks::tick(frames) + Generates::generate_batch_values: fills out[2][frames][n]. */
int groove_bank_render(groove_bank* bank, uint32_t frames, groove_block* out);
/* The same render submitted to the library's own side streams: it starts once everything submitted
 * to the ctx stream so far has finished (the previous users of `out` and of the bank), and runs BESIDE
 * what the ctx stream is given next.  Operations that take `out` (groove_fx_process, groove_mix,
 * groove_block_accumulate / download / upload / zero, another render into it) wait for it by
 * themselves; a caller that hands groove_block_device_ptr(out) to its own kernels on the ctx stream
 * calls groove_block_acquire(out) first.  Purpose: the per-entity walk of Orchestrator::run
 * (orchestrator.rs:397-457) has the instruments of block b+1 depend on nothing the effect chains of
 * block b produce, so a host that keeps two blocks per instrument submits render(b+1) before the
 * effects of block b and the two overlap (bench.py --workload chain-4096).  Results are identical to
 * groove_bank_render's. */
int groove_bank_render_async(groove_bank* bank, uint32_t frames, groove_block* out);
/* Orders the ctx stream after the asynchronous render that last filled `b` (no-op otherwise). */
int groove_block_acquire(groove_block* b);
/* HOST PACING (no reference counterpart; the reference's audio path is synchronous).  A device-side wait for another queue's
 * event costs the waiting stream 7 - 9 us whether or not the event has completed when the stream gets there, but a wait whose
 * event is already complete WHEN THE CALL IS MADE is dropped.  An offline host that has nothing else to do waits itself:
 * groove_block_wait_ready blocks the host (with the deadline of groove_synchronize) until the asynchronous render — and chain
 * head — that last filled `b` has finished, after which the ctx-stream operations on `b` carry no wait;
 * groove_block_wait_released blocks it until the point marked by the last groove_block_release(b) has passed, after which the
 * next groove_bank_render_async into `b` carries none.  With four blocks per instrument in rotation and the renders submitted
 * two blocks ahead both return at once in steady state (bench.py --workload chain-4096). */
int groove_block_wait_ready(groove_block* b);
int groove_block_wait_released(groove_block* b);
/* Marks the end of the block's consumers so far (one event on the ctx stream).  The next
 * groove_bank_render_async into `b` then waits for this point only — not for everything submitted to
 * the ctx stream by the time of that call — provided nothing uses the block on the ctx stream in
 * between.  With three blocks per instrument in rotation the released block's consumers are a whole
 * step in the past, the render needs no cross-queue wait and follows the previous one directly. */
int groove_block_release(groove_block* b);
/* Fused form of "tick every leaf and add its value to the running sum"
 * (orchestrator.rs:397-410) for instruments patched straight into the main mixer: renders
 * and accumulates into bus_dev[frames][2] (device) without materialising the block.
 * accumulate = 0 overwrites the bus, 1 adds to it.
 * Asynchronous like every call here: the bus is complete for anything ordered after this call on the
 * ctx stream (groove_download, groove_bus_to_i16, groove_bus_reduce, groove_synchronize ...).  Large
 * Welsh banks pipeline consecutive calls (block b+1's voice kernels, on the library's own per-kind
 * streams, do not wait for block b's bus sum); calls that touch the bank in between (note events,
 * groove_bank_set_param, groove_bank_render, groove_bank_download_state) join that pipeline first, so
 * the order of effects is the order of calls.  The voices' lane order inside the library is its own
 * business (a bank whose patches are interleaved voice by voice is kept patch-major): every index in
 * this API is the caller's voice index. */
int groove_bank_render_mix(groove_bank* bank, uint32_t frames, float* bus_dev, int accumulate);
/* groove_bank_render_mix whose bus reduction is DEFERRED: the block's partial rows are put on the bus by the bank's next deferred
 * render (its workgroups each add up a slice while their parameter loads are under way: no reduction launch), or — at the latest —
 * by the next call that waits for the ctx stream (groove_synchronize, downloads), records an event on it, or touches a bus
 * (groove_mix, groove_bank_render_mix, groove_bus_zero / _to_i16 / _reduce, groove_bus_flush).  For a host that renders a lone
 * small bank block after block (the reference's offline loop, orchestrator.rs:367-470, with one instrument): a time-parallel
 * bank's step is then one launch instead of two (256 Welsh voices 0.016 -> 0.013 ms per block).  The banks of a small project may
 * also take turns this way — each render carries the reduction of the one before it, in submission order, all on the ctx stream —
 * instead of running side by side with their cross-queue waits (config #5's 16,384-voice share of a GPU).  Banks it does not apply to
 * (not time-parallel, more than 2,048 partial rows — 16,384 paired Welsh voices) are rendered by groove_bank_render_mix.  The bus
 * is the same sum to fp32 rounding, but NOT the same bits as groove_bank_render_mix's, nor from one call pattern to another: the
 * order in which a column's rows are added depends on which launch carries them (a following deferred render adds them in batches
 * of eight rows per lane, partitioned by ITS grid; groove_bus_flush and the flush points add them in segments of 64 rows).  A
 * host that needs bit-reproducible buses calls groove_bank_render_mix. */
int groove_bank_render_mix_deferred(groove_bank* bank, uint32_t frames, float* bus_dev, int accumulate);
/* One block of a project of several SMALL banks of different kinds — Orchestrator::gather_audio's one sum over all instruments
 * (orchestrator.rs:397-410) — in ONE launch: the workgroups of one grid dispatch on their index into the Welsh / FM / sampler
 * time-parallel bodies (the most expensive kind first), they share one buffer of partial rows, and the block's one bus reduction is
 * deferred exactly as above (carried by the next such call, or flushed).  For projects with at most one time-parallel bank of each
 * kind and at most 2,048 partial rows in all (config #5's share of one of eight GPUs: 8,192 Welsh + 4,096 FM + 4,096 sampler
 * voices, three launches whose durations added -> one); anything else is rendered bank by bank by groove_bank_render_mix_deferred in
 * the order given (bus = banks[0] (+)= ... ).  Bit-reproducible from run to run; the same sum as bank by bank to fp32 rounding, not
 * the same bits (one reduction over all rows instead of one per bank).  No reference counterpart as a call: the reference ticks
 * every instrument inside gather_audio. */
int groove_banks_render_mix_deferred(groove_ctx* ctx, groove_bank* const* banks, uint32_t n_banks, uint32_t frames, float* bus_dev, int accumulate);
int groove_bus_flush(groove_ctx* ctx);
/* groove_bank_render_mix for a project whose banks render SIDE BY SIDE on the library's streams, PACED by the host (see
 * groove_block_wait_ready): the call blocks the host, as needed, until the reduction that frees this bank's slot of partial rows
 * (two of its blocks back) and the render of its PREVIOUS block have finished, so that neither the render streams nor the ctx
 * stream ever carry a cross-queue wait; the reduction of the block rendered by THIS call is launched by the bank's next paced
 * call, or by whatever flushes a bus (groove_bus_flush, groove_synchronize, a download, groove_event_record, any unpaced render
 * or mix).  The banks' sums reach a bus in the order of the calls, as with groove_bank_render_mix; same bits.  For the offline
 * host that renders several instruments patched straight into the main mixer block after block (orchestrator.rs:397-410):
 * config #5 on one GPU and its per-GPU share of eight.  No reference counterpart. */
int groove_bank_render_mix_paced(groove_bank* bank, uint32_t frames, float* bus_dev, int accumulate);
/* Every voice back to its freshly created state (oscillator phases, envelopes idle, filter memory,
 * sampler cursors); parameters stay; queued note events are dropped.  What the reference gets by
 * re-running a project from the top: Orchestrator::skip_to_start (orchestrator.rs:971-983) followed by
 * the Configurable::update_sample_rate fan-out that resets every entity (orchestrator.rs:125-127) —
 * without re-deriving or re-uploading the parameter tables. */
int groove_bank_reset(groove_bank* bank);
/* Which kernel form a render of `frames` frames of this bank takes right now (a static string; diagnosis: bench.py
 * prints it beside every parity figure so that the figure names the kernels it checked). */
const char* groove_bank_kernel_form(groove_bank* bank, uint32_t frames, int fused);
/* Raw state snapshot (checkpoint / debugging): words = groove_bank_state_words(). */
uint32_t groove_bank_state_words(groove_bank* bank);
int groove_bank_download_state(groove_bank* bank, uint32_t* host_words /* [words][n] */);

/* ---- effects (TransformsAudio) ---------------------------------------------------------- */
/* `Foo::new_with(&FooParams)` for n lanes (settings/src/effects.rs:59-117).  p has n entries. */
int groove_fx_create(groove_ctx* ctx, uint32_t kind, const groove_fx_params* p, uint32_t n, groove_fx** out);
int groove_fx_destroy(groove_fx* fx);
/* Clears the effect's memory (IIR state, delay lines, ring positions); parameters stay. */
int groove_fx_reset(groove_fx* fx);
/* TransformsAudio::transform_audio over a block, in place (orchestrator.rs:438-457). */
int groove_fx_process(groove_fx* fx, groove_block* inout, uint32_t frames);
/* The effects patched behind one instrument, in patch order, over one block: what the reference's gather walk does when
 * it meets a chain of TransformsAudio entities above a source (orchestrator.rs:438-457: each effect transforms the sum
 * of what is patched into it).  Same result, bit for bit, as groove_fx_process on chain[0] .. chain[n_fx-1] in turn;
 * the library fuses the stages that have no feedback inside a block (element-wise kinds, delay lines at least a block
 * long) into one pass over the block.  An effect may appear once. */
int groove_fx_chain_process(groove_fx* const* chain, uint32_t n_fx, groove_block* inout, uint32_t frames);
/* The chain's LEADING stages that have feedback inside a block (the IIR filters; delay lines shorter than the block),
 * submitted to the side stream that carries `inout`'s pending groove_bank_render_async, right behind that render:
 * *n_done stages are taken — IIR filters only (0 when the block has no pending render on one side stream, or when the chain
 * starts with another kind of stage); an effect that is processed ahead like this must be processed ahead in EVERY block
 * (its blocks would otherwise be submitted out of order); the caller hands chain + *n_done to groove_fx_chain_process when it reaches the block.  Same
 * bits as processing the whole chain there.  Purpose: in the render-ahead walk (groove_bank_render_async above) these
 * narrow, latency-bound kernels then run beside the wide stages of the PREVIOUS block instead of in front of them on
 * the ctx stream.  Parameter changes of those effects must be made before this call for the block it processes
 * (the effect is one block ahead of the ctx stream's walk).  No reference counterpart. */
int groove_fx_chain_process_async(groove_fx* const* chain, uint32_t n_fx, groove_block* inout, uint32_t frames, uint32_t* n_done);
/* groove_bank_render_async(bank, frames, out) followed by groove_fx_chain_process_async(chain, n_fx, out, frames, n_done) as ONE
 * call — which lets the library fuse the chain's first stage into the render kernel itself when it can: a 12 dB BiQuad
 * behind a bank that renders time-parallel (the frames of a voice's block are then already spread over a wavefront's lanes,
 * and the filter is one more scan on values that sit in registers: no separate launch, and the block is not read and
 * written a second time).  Same results as the two calls to f64 rounding of the filter's start states.  *n_done as above. */
int groove_bank_render_chain_async(groove_bank* bank, uint32_t frames, groove_block* out, groove_fx* const* chain, uint32_t n_fx,
                                   uint32_t* n_done);
/* Controllable for effects; lane = GROOVE_ALL_VOICES for all lanes. */
int groove_fx_set_param(groove_fx* fx, uint32_t lane, uint32_t control_index, double value01);
/* Replace all per-lane parameters (non-UNIFORM fields only may change). */
int groove_fx_set_params(groove_fx* fx, const groove_fx_params* p, uint32_t n);

/* ---- mix bus (Orchestrator::gather_audio, orchestrator.rs:367-470) ------------------------ */
/* bus_dev[f][ch] (+)= sum over blocks and lanes of block[ch][f][lane]. */
int groove_mix(groove_ctx* ctx, groove_block* const* blocks, uint32_t n_blocks, uint32_t frames,
               float* bus_dev, int accumulate);
/* groove_mix of ONE block that a render or an effect chain has just filled (its lane sums are still valid: a few short rows),
 * with the reduction of those rows DEFERRED the way groove_bank_render_mix_deferred defers a render's: the next effect-chain
 * launch on the ctx stream (or the next deferred render) sums them onto the bus in passing, and groove_bus_flush — or any call
 * that waits for the ctx stream, records an event on it or touches a bus — does it at the latest.  For the host that walks a
 * chain block after block (render, effects, main-mixer sum: orchestrator.rs:397-457): the step loses a launch that is all
 * latency.  A block without valid lane sums (or with more than 2,048 rows) is mixed by groove_mix at once.  No reference
 * counterpart. */
int groove_mix_deferred(groove_ctx* ctx, groove_block* block, uint32_t frames, float* bus_dev, int accumulate);
/* Device scratch the caller can use as a bus: frames*2 floats, zeroed. */
int groove_bus_create(groove_ctx* ctx, size_t frames, float** out_dev);
int groove_bus_destroy(groove_ctx* ctx, float* bus_dev);
int groove_bus_zero(groove_ctx* ctx, float* bus_dev, size_t frames);
int groove_download(groove_ctx* ctx, const float* dev, float* host, size_t n_floats);
int groove_upload(groove_ctx* ctx, float* dev, const float* host, size_t n_floats);
/* WAV sink quantisation (orchestration/src/helpers.rs:79-91): interleaved i16 = (x*32767) as i16
 * (truncate toward zero, saturating), computed on the device from bus_dev[frames][2]. */
int groove_bus_to_i16(groove_ctx* ctx, const float* bus_dev, size_t frames, int16_t* host_out);

/* ---- multi-GPU (no reference counterpart) ----------------------------------- */
/* One process per GPU.  Rank 0 calls groove_comm_unique_id, the launcher broadcasts the 128
 * bytes, every rank calls groove_comm_init; groove_bus_reduce sums bus_dev[frames_total][2]
 * (fp32) onto `root` with one RCCL reduce over xGMI. */
int groove_comm_unique_id(groove_ctx* ctx /* may be NULL */, uint8_t id_out[128]);
int groove_comm_init(groove_ctx* ctx, const uint8_t id[128], int rank, int world_size);
/* Ranks that joined the communicator (ncclCommCount); 1 when no communicator has been set up.  The
 * launcher checks it against the number of GPUs it asked for (bench.py: `rccl_ranks`). */
int groove_comm_ranks(groove_ctx* ctx, int* out_ranks);
int groove_comm_destroy(groove_ctx* ctx);
int groove_bus_reduce(groove_ctx* ctx, float* bus_dev, size_t frames_total, int root);

#ifdef __cplusplus
}
#endif
#endif /* GROOVE_HIP_H */
