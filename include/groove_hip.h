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
 *     single-threaded audio path (orchestrator.rs:367-470);
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
void groove_shutdown(groove_ctx* ctx);
/* ctx may be NULL: returns the last error of the calling thread. */
const char* groove_last_error(groove_ctx* ctx);
/* Use the caller's hipStream_t (e.g. a torch stream) instead of the ctx's own. */
int groove_set_stream(groove_ctx* ctx, void* hip_stream);
int groove_synchronize(groove_ctx* ctx);
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
float* groove_block_device_ptr(groove_block* b);
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
/* Every voice back to its freshly created state (oscillator phases, envelopes idle, filter memory,
 * sampler cursors); parameters stay; queued note events are dropped.  What the reference gets by
 * re-running a project from the top: Orchestrator::skip_to_start (orchestrator.rs:971-983) followed by
 * the Configurable::update_sample_rate fan-out that resets every entity (orchestrator.rs:125-127) —
 * without re-deriving or re-uploading the parameter tables. */
int groove_bank_reset(groove_bank* bank);
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
/* Controllable for effects; lane = GROOVE_ALL_VOICES for all lanes. */
int groove_fx_set_param(groove_fx* fx, uint32_t lane, uint32_t control_index, double value01);
/* Replace all per-lane parameters (non-UNIFORM fields only may change). */
int groove_fx_set_params(groove_fx* fx, const groove_fx_params* p, uint32_t n);

/* ---- mix bus (Orchestrator::gather_audio, orchestrator.rs:367-470) ------------------------ */
/* bus_dev[f][ch] (+)= sum over blocks and lanes of block[ch][f][lane]. */
int groove_mix(groove_ctx* ctx, groove_block* const* blocks, uint32_t n_blocks, uint32_t frames,
               float* bus_dev, int accumulate);
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
int groove_comm_unique_id(groove_ctx* ctx, uint8_t id_out[128]);
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
