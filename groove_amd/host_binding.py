"""ctypes binding of groove_amd/host/libgroove_host.so — the compiled (C++) host layer that
mirrors the reference's Orchestrator / entity surface above the C ABI."""
import ctypes as C
import os

import numpy as np

from . import abi_types as T

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(HERE, "host", "libgroove_host.so")
_fp = C.POINTER(C.c_float)

STEP_FLAT, STEP_SLOPE, STEP_LOGARITHMIC, STEP_EXPONENTIAL, STEP_TRIGGERED = range(5)


def load():
    if not os.path.exists(HOST_LIB):
        raise RuntimeError(f"{HOST_LIB} not found: run __graft_entry__.build()")
    L = C.CDLL(HOST_LIB)
    vp, i, u32, d = C.c_void_p, C.c_int, C.c_uint32, C.c_double
    sig = {
        "gh_orchestrator_new": (vp, [i, u32, d]), "gh_orchestrator_free": (None, [vp]),
        "gh_last_error": (C.c_char_p, [vp]), "gh_add_toy_source": (i, [vp, d]),
        "gh_add_welsh": (i, [vp, C.POINTER(T.WelshParams), u32]), "gh_add_fm": (i, [vp, C.POINTER(T.FmParams), u32]),
        "gh_add_drumkit": (i, [vp, _fp, C.c_uint64, C.POINTER(T.SampleDesc), u32, C.POINTER(C.c_int)]),
        "gh_add_effect": (i, [vp, u32, C.POINTER(T.FxParams)]),
        "gh_patch": (i, [vp, i, i]), "gh_patch_chain_to_main_mixer": (i, [vp, C.POINTER(C.c_int), u32]),
        "gh_unpatch_all": (None, [vp]), "gh_set_render_ahead": (None, [vp, i]), "gh_set_fused_direct": (None, [vp, i]), "gh_connect_midi_downstream": (i, [vp, i, i]),
        "gh_add_timer": (i, [vp, d]), "gh_add_sequencer": (i, [vp]),
        "gh_sequencer_insert": (i, [vp, i, i, i, d, d]), "gh_sequencer_set_end": (i, [vp, i, d]),
        "gh_add_control_trip": (i, [vp, i, C.c_char_p, d]), "gh_control_trip_add_step": (i, [vp, i, i, d, d, d]),
        "gh_control_step_value": (d, [i, d, d, d]), "gh_last_allocated_voice": (i, [vp, i]),
        "gh_gather_audio": (i, [vp, u32, _fp]), "gh_performance_frames": (C.c_uint64, [vp]),
        "gh_run": (C.c_int64, [vp, u32, _fp, C.c_uint64, i]), "gh_render_to_wav": (i, [vp, u32, C.c_char_p]),
        "gh_synthetic_kit": (i, [u32, _fp, C.c_uint64, C.POINTER(T.SampleDesc), u32, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L


def synthetic_kit(sample_rate=T.DEFAULT_SAMPLE_RATE):
    """The sample bank `groove-cli-hip --synthetic-kit` builds (data only, no GPU): (pcm, descs, key_to_sample)."""
    L = load()
    frames, nd = C.c_uint64(), C.c_uint32()
    assert L.gh_synthetic_kit(sample_rate, None, 0, None, 0, None, C.byref(frames), C.byref(nd)) == 0
    pcm = np.empty(frames.value, dtype=np.float32)
    descs = (T.SampleDesc * nd.value)()
    k2s = (C.c_int * 128)()
    assert L.gh_synthetic_kit(sample_rate, pcm.ctypes.data_as(_fp), pcm.size, descs, nd.value, k2s, None, None) == 0
    return pcm, descs, list(k2s)


class Orchestrator:
    MAIN_MIXER = 0

    def __init__(self, device=0, sample_rate=T.DEFAULT_SAMPLE_RATE, bpm=128.0):
        self.L = load()
        self.h = self.L.gh_orchestrator_new(device, sample_rate, bpm)
        if not self.h:
            raise RuntimeError("Orchestrator: no HIP device / context (there is no CPU path)")

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.gh_last_error(self.h).decode())

    def add_toy_source(self, level): return self.L.gh_add_toy_source(self.h, level)
    def add_welsh(self, patch, voices=8): return self.L.gh_add_welsh(self.h, C.byref(patch), voices)
    def add_fm(self, patch, voices=8): return self.L.gh_add_fm(self.h, C.byref(patch), voices)

    def add_drumkit(self, pcm, descs, key_to_sample):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        k2s = (C.c_int * 128)(*key_to_sample)
        return self.L.gh_add_drumkit(self.h, pcm.ctypes.data_as(_fp), pcm.size, descs, len(descs), k2s)

    def add_effect(self, kind, params): return self.L.gh_add_effect(self.h, kind, C.byref(params))
    def patch(self, source, sink): return self.L.gh_patch(self.h, source, sink)

    def patch_chain_to_main_mixer(self, uids):
        arr = (C.c_int * len(uids))(*uids)
        return self.L.gh_patch_chain_to_main_mixer(self.h, arr, len(uids))

    def unpatch_all(self): self.L.gh_unpatch_all(self.h)
    def set_fused_direct(self, on):
        """Instruments patched straight into the main mixer render fused onto the bus (default) or through their blocks."""
        self.L.gh_set_fused_direct(self.h, 1 if on else 0)

    def set_render_ahead(self, on):
        """Offline runs: False = block by block, True = instruments one block ahead of the effects whenever the graph allows
        (the default, "auto", does so only for instruments whose render is long enough to be worth the hand-over)."""
        self.L.gh_set_render_ahead(self.h, 2 if on else 0)
    def connect_midi_downstream(self, uid, channel): self._chk(self.L.gh_connect_midi_downstream(self.h, uid, channel))
    def add_timer(self, beats): return self.L.gh_add_timer(self.h, beats)
    def add_sequencer(self): return self.L.gh_add_sequencer(self.h)

    def sequencer_insert(self, uid, channel, key, start_beat, duration_beats):
        self._chk(self.L.gh_sequencer_insert(self.h, uid, channel, key, start_beat, duration_beats))

    def sequencer_set_end(self, uid, beats): self._chk(self.L.gh_sequencer_set_end(self.h, uid, beats))

    def add_control_trip(self, target, param, start_beat=0.0):
        uid = self.L.gh_add_control_trip(self.h, target, param.encode(), start_beat)
        if uid < 0:
            raise RuntimeError(self.L.gh_last_error(self.h).decode())
        return uid

    def control_trip_add_step(self, uid, kind, start, end, beats):
        self._chk(self.L.gh_control_trip_add_step(self.h, uid, kind, start, end, beats))

    def last_allocated_voice(self, uid): return self.L.gh_last_allocated_voice(self.h, uid)

    def gather_audio(self, frames):
        out = np.zeros((frames, 2), dtype=np.float32)
        self._chk(self.L.gh_gather_audio(self.h, frames, out.ctypes.data_as(_fp)))
        return out

    def performance_frames(self): return self.L.gh_performance_frames(self.h)

    def run(self, buffer_frames=64, performance=False):
        cap = self.performance_frames() + buffer_frames
        out = np.zeros((cap, 2), dtype=np.float32)
        n = self.L.gh_run(self.h, buffer_frames, out.ctypes.data_as(_fp), cap, 1 if performance else 0)
        if n < 0:
            raise RuntimeError(self.L.gh_last_error(self.h).decode())
        return out[:n]

    def render_to_wav(self, path, buffer_frames=256):
        self._chk(self.L.gh_render_to_wav(self.h, buffer_frames, path.encode()))

    def close(self):
        if self.h:
            self.L.gh_orchestrator_free(self.h)
            self.h = None
