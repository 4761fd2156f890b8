"""Python host-side mirror of the reference's entity trait surface, over the C ABI.

Names follow the reference traits: `Configurable.update_sample_rate`,
`HandlesMidi.handle_midi_message` → `note_on`/`note_off`, `Ticks.tick` +
`Generates.generate_batch_values`, `TransformsAudio.transform_audio`,
`Controllable.control_set_param_by_index`, and `Orchestrator.gather_audio`
(/root/reference/orchestration/src/orchestrator.rs:367-470).  This is the test/bench-side
binding; the compiled host layer (C++ Orchestrator, WAV sink) lives in groove_amd/host/.
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from . import abi_types as T

_fp = C.POINTER(C.c_float)


class Context:
    """One GPU + one HIP stream (groove_ctx)."""

    def __init__(self, device=0, comm=None):
        """comm = (unique id bytes, rank, world): groove_init_comm — the rank's RCCL communicator before the streams."""
        self.L = _lib.load()
        h = C.c_void_p()
        if comm is None:
            _lib.check(self.L.groove_init(device, C.byref(h)))
        else:
            uid, rank, world = comm
            buf = (C.c_uint8 * 128).from_buffer_copy(uid)
            _lib.check(self.L.groove_init_comm(device, buf, rank, world, C.byref(h)))
        self.h = h
        self.device = device

    @staticmethod
    def new_comm_unique_id():
        """groove_comm_unique_id(NULL, ...): rank 0's id, before any ctx exists."""
        L = _lib.load()
        buf = (C.c_uint8 * 128)()
        _lib.check(L.groove_comm_unique_id(None, buf))
        return bytes(buf)

    @property
    def sync_timeout_ms(self):
        return self.L.groove_sync_timeout_ms(self.h)

    @sync_timeout_ms.setter
    def sync_timeout_ms(self, ms):
        _lib.check(self.L.groove_set_sync_timeout_ms(self.h, int(ms)), self.h)

    def debug_spin(self, side_stream, ms):
        _lib.check(self.L.groove_debug_spin(self.h, side_stream, ms), self.h)

    def debug_info(self):
        import json
        buf = C.create_string_buffer(8192)
        _lib.check(self.L.groove_debug_info(self.h, buf, len(buf)), self.h)
        return json.loads(buf.value.decode())

    # Configurable
    def update_sample_rate(self, hz):
        _lib.check(self.L.groove_update_sample_rate(self.h, hz), self.h)

    @property
    def sample_rate(self):
        return self.L.groove_sample_rate(self.h)

    @property
    def time_parallel_max_voices(self):
        return self.L.groove_time_parallel_max_voices(self.h)

    @time_parallel_max_voices.setter
    def time_parallel_max_voices(self, n):
        _lib.check(self.L.groove_set_time_parallel_max_voices(self.h, n), self.h)

    @property
    def time_parallel_pair_min_voices(self):
        return self.L.groove_time_parallel_pair_min_voices(self.h)

    @time_parallel_pair_min_voices.setter
    def time_parallel_pair_min_voices(self, n):
        _lib.check(self.L.groove_set_time_parallel_pair_min_voices(self.h, n), self.h)

    @property
    def look_ahead(self):
        return self.L.groove_look_ahead(self.h)

    @look_ahead.setter
    def look_ahead(self, bits):
        _lib.check(self.L.groove_set_look_ahead(self.h, bits), self.h)

    @property
    def pipeline_min_waves(self):
        return self.L.groove_pipeline_min_waves(self.h)

    @pipeline_min_waves.setter
    def pipeline_min_waves(self, n):
        _lib.check(self.L.groove_set_pipeline_min_waves(self.h, n), self.h)

    @property
    def fx_allpass_stream(self):
        """A chain's closing reverb leaves its all-passes on a side stream (include/groove_hip.h groove_set_fx_allpass_stream)."""
        return bool(self.L.groove_fx_allpass_stream(self.h))

    @fx_allpass_stream.setter
    def fx_allpass_stream(self, on):
        _lib.check(self.L.groove_set_fx_allpass_stream(self.h, 1 if on else 0), self.h)

    @property
    def split_max_waves(self):
        return self.L.groove_split_max_waves(self.h)

    @split_max_waves.setter
    def split_max_waves(self, n):
        _lib.check(self.L.groove_set_split_max_waves(self.h, n), self.h)

    def flush_bus(self):
        _lib.check(self.L.groove_bus_flush(self.h), self.h)

    def set_stream(self, hip_stream):
        _lib.check(self.L.groove_set_stream(self.h, C.c_void_p(hip_stream)), self.h)

    def synchronize(self):
        _lib.check(self.L.groove_synchronize(self.h), self.h)

    # events for measurement
    def event(self):
        e = C.c_void_p()
        _lib.check(self.L.groove_event_create(self.h, C.byref(e)), self.h)
        return e

    def record(self, ev):
        _lib.check(self.L.groove_event_record(self.h, ev), self.h)

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        _lib.check(self.L.groove_event_elapsed_ms(self.h, a, b, C.byref(ms)), self.h)
        return ms.value

    def block(self, n, frames_cap=T.BLOCK_FRAMES):
        return Block(self, n, frames_cap)

    def bus(self, frames):
        return Bus(self, frames)

    # Orchestrator::gather_audio over materialised blocks
    def mix(self, blocks, frames, bus, accumulate=False):
        arr = (C.c_void_p * len(blocks))(*[b.h for b in blocks])
        _lib.check(self.L.groove_mix(self.h, arr, len(blocks), frames, bus.ptr, 1 if accumulate else 0), self.h)

    def render_mix_banks_deferred(self, instruments, bus, frames, accumulate=False, at_frame=0):
        """groove_banks_render_mix_deferred: one block of several small instruments of different kinds in ONE launch (one bus sum
        over all of them, deferred); projects it does not fit are rendered instrument by instrument, in the order given."""
        arr = (C.c_void_p * len(instruments))(*[i.h for i in instruments])
        ptr = bus.at(at_frame) if at_frame else bus.ptr
        _lib.check(self.L.groove_banks_render_mix_deferred(self.h, arr, len(instruments), frames, ptr, 1 if accumulate else 0), self.h)

    def mix_deferred(self, block, frames, bus, accumulate=False):
        """groove_mix_deferred: the block's few lane-sum rows ride in the next effect-chain launch (or flush_bus)."""
        _lib.check(self.L.groove_mix_deferred(self.h, block.h, frames, bus.ptr, 1 if accumulate else 0), self.h)

    # the effects patched behind one instrument, in patch order, over one block (same bits as one transform_audio each)
    def transform_chain(self, effects, block, frames=None):
        frames = block.cap if frames is None else frames
        arr = (C.c_void_p * len(effects))(*[e.h for e in effects])
        _lib.check(self.L.groove_fx_chain_process(arr, len(effects), block.h, frames), self.h)

    def transform_chain_async(self, effects, block, frames=None):
        """groove_fx_chain_process_async: the chain's leading IIR stages behind the block's pending render, on its side
        stream.  Returns how many stages were taken (the rest goes to transform_chain later)."""
        frames = block.cap if frames is None else frames
        arr = (C.c_void_p * len(effects))(*[e.h for e in effects])
        done = C.c_uint32(0)
        _lib.check(self.L.groove_fx_chain_process_async(arr, len(effects), block.h, frames, C.byref(done)), self.h)
        return done.value

    # multi-GPU
    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        _lib.check(self.L.groove_comm_unique_id(self.h, buf), self.h)
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(self.L.groove_comm_init(self.h, buf, rank, world), self.h)

    def comm_ranks(self):
        n = C.c_int()
        _lib.check(self.L.groove_comm_ranks(self.h, C.byref(n)), self.h)
        return n.value

    def bus_reduce(self, bus, frames_total, root=0):
        _lib.check(self.L.groove_bus_reduce(self.h, bus.ptr, frames_total, root), self.h)

    def close(self):
        if self.h:
            self.L.groove_shutdown(self.h)
            self.h = None


class Block:
    """Device stereo block [2][frames_cap][n] fp32."""

    def __init__(self, ctx, n, frames_cap):
        self.ctx, self.n, self.cap = ctx, n, frames_cap
        h = C.c_void_p()
        _lib.check(ctx.L.groove_block_create(ctx.h, n, frames_cap, C.byref(h)), ctx.h)
        self.h = h

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert host.shape[0] == 2 and host.shape[2] == self.n
        _lib.check(self.ctx.L.groove_block_upload(self.h, host.ctypes.data_as(_fp), host.shape[1]), self.ctx.h)

    def download(self, frames=None):
        frames = self.cap if frames is None else frames
        out = np.empty((2, frames, self.n), dtype=np.float32)
        _lib.check(self.ctx.L.groove_block_download(self.h, out.ctypes.data_as(_fp), frames), self.ctx.h)
        return out

    def device_ptr(self):
        return self.ctx.L.groove_block_device_ptr(self.h)

    def mark_dirty(self):
        _lib.check(self.ctx.L.groove_block_mark_dirty(self.h), self.ctx.h)

    def release(self):
        """groove_block_release: the block's consumers so far are all that the next asynchronous render into it waits for."""
        _lib.check(self.ctx.L.groove_block_release(self.h), self.ctx.h)

    def wait_ready(self):
        """Host pacing: block the host until the asynchronous render that last filled this block has finished."""
        _lib.check(self.ctx.L.groove_block_wait_ready(self.h), self.ctx.h)

    def wait_released(self):
        """Host pacing: block the host until the point of the last release() has passed on the ctx stream."""
        _lib.check(self.ctx.L.groove_block_wait_released(self.h), self.ctx.h)

    def destroy(self):
        if self.h:
            self.ctx.L.groove_block_destroy(self.h)
            self.h = None


class Bus:
    """Device stereo bus [frames][2] fp32."""

    def __init__(self, ctx, frames):
        self.ctx, self.frames = ctx, frames
        p = C.c_void_p()
        _lib.check(ctx.L.groove_bus_create(ctx.h, frames, C.byref(p)), ctx.h)
        self.ptr = p

    def at(self, frame):
        """Device pointer to bus[frame]."""
        return C.c_void_p(self.ptr.value + frame * 8)

    def zero(self):
        _lib.check(self.ctx.L.groove_bus_zero(self.ctx.h, self.ptr, self.frames), self.ctx.h)

    def download(self, frames=None):
        frames = self.frames if frames is None else frames
        out = np.empty((frames, 2), dtype=np.float32)
        _lib.check(self.ctx.L.groove_download(self.ctx.h, self.ptr, out.ctypes.data_as(_fp), frames * 2), self.ctx.h)
        return out

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert host.ndim == 2 and host.shape[1] == 2 and host.shape[0] <= self.frames
        _lib.check(self.ctx.L.groove_upload(self.ctx.h, self.ptr, host.ctypes.data_as(_fp), host.size), self.ctx.h)

    def to_i16(self, frames=None):
        frames = self.frames if frames is None else frames
        out = np.empty((frames, 2), dtype=np.int16)
        _lib.check(self.ctx.L.groove_bus_to_i16(self.ctx.h, self.ptr, frames, out.ctypes.data_as(C.POINTER(C.c_int16))), self.ctx.h)
        return out

    def destroy(self):
        if self.ptr:
            self.ctx.L.groove_bus_destroy(self.ctx.h, self.ptr)
            self.ptr = None


class _Slice:
    """A view of a Bus starting at `frame` (duck-types Bus.ptr for render_mix / mix)."""

    def __init__(self, bus, frame):
        self.ptr = bus.at(frame)


class Instrument:
    """IsInstrument: a bank of n homogeneous voices, one lane each."""

    def __init__(self, ctx, handle, n):
        self.ctx, self.h, self.n = ctx, handle, n

    # HandlesMidi
    def handle_midi_events(self, events):
        _lib.check(self.ctx.L.groove_bank_note_events(self.h, events, len(events)), self.ctx.h)

    def note_on(self, voice, key, velocity=127):
        self.handle_midi_events(T.note_events([(voice, key, True)]))

    def note_off(self, voice, key=0, velocity=0):
        self.handle_midi_events(T.note_events([(voice, key, False)]))

    # Controllable
    def control_set_param_by_index(self, index, value01, voice=T.ALL_VOICES):
        _lib.check(self.ctx.L.groove_bank_set_param(self.h, voice, index, value01), self.ctx.h)

    # Ticks + Generates
    def generate_batch_values(self, block, frames=None):
        frames = block.cap if frames is None else frames
        _lib.check(self.ctx.L.groove_bank_render(self.h, frames, block.h), self.ctx.h)

    tick = generate_batch_values

    def generate_batch_values_async(self, block, frames=None):
        """groove_bank_render_async: the render runs beside what the ctx stream is given next."""
        frames = block.cap if frames is None else frames
        _lib.check(self.ctx.L.groove_bank_render_async(self.h, frames, block.h), self.ctx.h)

    def generate_batch_values_chain_async(self, block, effects, frames=None):
        """groove_bank_render_chain_async: the asynchronous render with the chain's leading IIR stages behind it (the first
        one fused into the render kernel when the library can).  Returns how many stages of `effects` were taken."""
        frames = block.cap if frames is None else frames
        arr = (C.c_void_p * len(effects))(*[e.h for e in effects])
        done = C.c_uint32(0)
        _lib.check(self.ctx.L.groove_bank_render_chain_async(self.h, frames, block.h, arr, len(effects), C.byref(done)), self.ctx.h)
        return done.value

    def render_mix(self, bus, frames, accumulate=False, at_frame=0):
        ptr = bus.at(at_frame) if at_frame else bus.ptr
        _lib.check(self.ctx.L.groove_bank_render_mix(self.h, frames, ptr, 1 if accumulate else 0), self.ctx.h)

    def render_mix_paced(self, bus, frames, accumulate=False, at_frame=0):
        """groove_bank_render_mix_paced: banks side by side, the host waits for the events itself; this block's bus reduction is
        launched by the bank's next paced call (or a flush point)."""
        ptr = bus.at(at_frame) if at_frame else bus.ptr
        _lib.check(self.ctx.L.groove_bank_render_mix_paced(self.h, frames, ptr, 1 if accumulate else 0), self.ctx.h)

    def render_mix_deferred(self, bus, frames, accumulate=False, at_frame=0):
        """groove_bank_render_mix_deferred: the block's bus reduction is left to this bank's next deferred render (or to the next
        call that waits for the ctx stream, records an event on it or touches a bus)."""
        ptr = bus.at(at_frame) if at_frame else bus.ptr
        _lib.check(self.ctx.L.groove_bank_render_mix_deferred(self.h, frames, ptr, 1 if accumulate else 0), self.ctx.h)

    def kernel_form(self, frames=T.BLOCK_FRAMES, fused=True):
        return self.ctx.L.groove_bank_kernel_form(self.h, frames, 1 if fused else 0).decode()

    def reset(self):
        """groove_bank_reset: every voice back to its freshly created state."""
        _lib.check(self.ctx.L.groove_bank_reset(self.h), self.ctx.h)

    def download_state(self):
        w = self.ctx.L.groove_bank_state_words(self.h)
        out = np.empty((w, self.n), dtype=np.uint32)
        _lib.check(self.ctx.L.groove_bank_download_state(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32))), self.ctx.h)
        return out

    def destroy(self):
        if self.h:
            self.ctx.L.groove_bank_destroy(self.h)
            self.h = None


class WelshSynth(Instrument):
    """WelshSynth::new_with(&WelshSynthParams) for n voices."""

    def __init__(self, ctx, params):
        h = C.c_void_p()
        _lib.check(ctx.L.groove_welsh_create(ctx.h, params, len(params), C.byref(h)), ctx.h)
        super().__init__(ctx, h, len(params))


class FmSynth(Instrument):
    def __init__(self, ctx, params):
        h = C.c_void_p()
        _lib.check(ctx.L.groove_fm_create(ctx.h, params, len(params), C.byref(h)), ctx.h)
        super().__init__(ctx, h, len(params))


class Sampler(Instrument):
    """Sampler / Drumkit over a shared mono sample bank."""

    def __init__(self, ctx, pcm, descs, params):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        h = C.c_void_p()
        _lib.check(ctx.L.groove_sampler_create(ctx.h, pcm.ctypes.data_as(_fp), pcm.size, descs, len(descs),
                                               params, len(params), C.byref(h)), ctx.h)
        super().__init__(ctx, h, len(params))


class Effect:
    """IsEffect: TransformsAudio per lane over a block."""

    def __init__(self, ctx, kind, params):
        self.ctx, self.kind, self.n = ctx, kind, len(params)
        h = C.c_void_p()
        _lib.check(ctx.L.groove_fx_create(ctx.h, kind, params, len(params), C.byref(h)), ctx.h)
        self.h = h

    def transform_audio(self, block, frames=None):
        frames = block.cap if frames is None else frames
        _lib.check(self.ctx.L.groove_fx_process(self.h, block.h, frames), self.ctx.h)

    def control_set_param_by_index(self, index, value01, lane=T.ALL_VOICES):
        _lib.check(self.ctx.L.groove_fx_set_param(self.h, lane, index, value01), self.ctx.h)

    def set_params(self, params):
        _lib.check(self.ctx.L.groove_fx_set_params(self.h, params, len(params)), self.ctx.h)

    def reset(self):
        _lib.check(self.ctx.L.groove_fx_reset(self.h), self.ctx.h)

    def destroy(self):
        if self.h:
            self.ctx.L.groove_fx_destroy(self.h)
            self.h = None
