"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package `groove_amd` (see oracle_dsp.hpp header
for the parity status: most per-voice DSP is "parity unpinned" because the reference's
arithmetic lives in an un-vendored dependency).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

import sys
if REPO not in sys.path:
    sys.path.insert(0, REPO)
from groove_amd import abi_types as T  # noqa: E402  (shared POD layouts only; no product code paths)


def build(native=False, ref=True):
    targets = ["all"] + (["native"] if native else []) + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


def _load(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build(native=name.endswith("native.so"))
    return C.CDLL(path)


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _bind(lib):
    d, u32, vp, i = C.c_double, C.c_uint32, C.c_void_p, C.c_int
    sig = {
        "oracle_note_to_frequency": (d, [i]), "oracle_semis_and_cents": (d, [i, d]), "oracle_octaves": (d, [i]),
        "oracle_percent_to_frequency": (d, [d]), "oracle_frequency_to_percent": (d, [d]),
        "oracle_denormalize_q": (d, [d]), "oracle_mma_concave": (d, [d]), "oracle_mma_convex": (d, [d]),
        "oracle_wav_quantise": (i, [d]), "oracle_bitcrush_f32": (C.c_float, [C.c_float, u32]),
        "oracle_dca": (None, [d, d, d, _dp]),
        "oracle_rbj_lowpass": (None, [d, d, d, _dp]), "oracle_rbj_highpass": (None, [d, d, d, _dp]),
        "oracle_rbj_for_kind": (i, [u32, C.POINTER(T.FxParams), d, _dp]),
        "oracle_lp24_coeffs": (None, [d, d, d, _dp]), "oracle_lp24_run": (None, [d, d, d, _dp, _dp, u32]),
        "oracle_biquad_df1_run": (None, [_dp, _dp, _dp, u32]),
        "oracle_oscillator_run": (None, [C.POINTER(T.OscillatorParams), d, d, _dp, _dp, u32, C.POINTER(u32)]),
        "oracle_envelope_run": (None, [C.POINTER(T.EnvelopeParams), d, u32, _dp, u32]),
        "oracle_performance_total_frames": (C.c_uint64, [d, d, d]),
        "oracle_run_frames": (C.c_uint64, [d, d, d, u32]),
        "oracle_run_performance_frames": (C.c_uint64, [d, d, d, u32]),
        "oracle_welsh_create": (vp, [C.POINTER(T.WelshParams), u32, u32]),
        "oracle_fm_create": (vp, [C.POINTER(T.FmParams), u32, u32]),
        "oracle_sampler_create": (vp, [_fp, C.c_uint64, C.POINTER(T.SampleDesc), u32, C.POINTER(T.SamplerParams), u32, u32]),
        "oracle_bank_destroy": (None, [vp]),
        "oracle_bank_note_events": (None, [vp, C.POINTER(T.NoteEvent), u32]),
        "oracle_bank_render": (None, [vp, u32, _dp]),
        "oracle_bank_set_param": (C.c_int, [vp, u32, u32, C.c_double]),
        "oracle_bank_render_bus": (None, [vp, u32, _dp]),
        "oracle_bank_render_bus_mt": (None, [vp, u32, _dp, u32]),
        "oracle_bank_render_bus_blocks_mt": (None, [vp, u32, u32, _dp, u32]),
        "oracle_hardware_concurrency": (C.c_uint, []),
        "oracle_fx_create": (vp, [u32, C.POINTER(T.FxParams), u32, u32]),
        "oracle_fx_destroy": (None, [vp]),
        "oracle_fx_set_params": (None, [vp, C.POINTER(T.FxParams), u32]),
        "oracle_fx_process": (None, [vp, _dp, u32]),
        "oracle_mix": (None, [_dp, u32, u32, _dp, i]),
        "oracle_graph_create": (vp, [u32]), "oracle_graph_destroy": (None, [vp]),
        "oracle_graph_add_source_const": (i, [vp, d]), "oracle_graph_add_toy_effect": (i, [vp]),
        "oracle_graph_add_effect": (i, [vp, u32, C.POINTER(T.FxParams)]),
        "oracle_graph_add_instrument": (i, [vp, vp]),
        "oracle_graph_patch": (i, [vp, i, i]), "oracle_graph_unpatch_all": (None, [vp]),
        "oracle_graph_note_events": (None, [vp, i, C.POINTER(T.NoteEvent), u32]),
        "oracle_graph_gather": (None, [vp, u32, _dp]),
        "oracle_graph_set_bpm": (None, [vp, d]), "oracle_graph_skip_to_start": (None, [vp]),
        "oracle_graph_add_control_trip": (i, [vp, i, u32, d]), "oracle_graph_trip_add_step": (i, [vp, i, i, d, d, d]),
        "oracle_control_step_value": (d, [i, d, d, d]), "oracle_graph_tick": (None, [vp, u32, _dp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_LIB = None


def lib(native=False):
    global _LIB
    if native:
        return _bind(_load("liboracle_native.so"))
    if _LIB is None:
        # GROOVE_ORACLE_LIB: another build of the same oracle (liboracle_asan.so: tests/test_host_sanitizers.py runs the
        # known-answer tests against the AddressSanitizer + UBSan build in a child process)
        _LIB = _bind(_load(os.environ.get("GROOVE_ORACLE_LIB", "liboracle.so")))
    return _LIB


def ref_lib():
    """oracle/_ref/libfilters004.so — the reference's in-tree C filter text, compiled by
    `make -C oracle ref` from /root/reference/doc/filters004.txt; None when unavailable."""
    path = os.path.join(HERE, "_ref", "libfilters004.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/doc/filters004.txt"):
            build(ref=True)
        if not os.path.exists(path):
            return None
    r = C.CDLL(path)
    r.ref_szxform.argtypes = [C.c_double] * 8 + [_dp, _fp]
    r.ref_szxform.restype = None
    r.ref_iir_run.argtypes = [_fp, C.c_uint, _fp, _fp, C.c_uint]
    r.ref_iir_run.restype = None
    return r


def _dptr(a):
    return a.ctypes.data_as(_dp)


class Bank:
    """Oracle instrument bank (n voices); out blocks are f64 [2][frames][n]."""

    def __init__(self, handle, n, lib_=None):
        self.h, self.n, self.L = handle, n, lib_ or lib()

    @classmethod
    def welsh(cls, params, sr=T.DEFAULT_SAMPLE_RATE, lib_=None):
        L = lib_ or lib()
        return cls(L.oracle_welsh_create(params, len(params), sr), len(params), L)

    @classmethod
    def fm(cls, params, sr=T.DEFAULT_SAMPLE_RATE, lib_=None):
        L = lib_ or lib()
        return cls(L.oracle_fm_create(params, len(params), sr), len(params), L)

    @classmethod
    def sampler(cls, pcm, descs, params, sr=T.DEFAULT_SAMPLE_RATE, lib_=None):
        L = lib_ or lib()
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        h = L.oracle_sampler_create(pcm.ctypes.data_as(_fp), pcm.size, descs, len(descs), params, len(params), sr)
        return cls(h, len(params), L)

    def note_events(self, ev):
        self.L.oracle_bank_note_events(self.h, ev, len(ev))

    def set_param(self, control_index, value01, voice=T.ALL_VOICES):
        """Controllable::control_set_param_by_index on a sounding instrument (the voices' state stays)."""
        assert self.L.oracle_bank_set_param(self.h, voice, control_index, value01) == 0

    def render(self, frames):
        out = np.zeros((2, frames, self.n), dtype=np.float64)
        self.L.oracle_bank_render(self.h, frames, _dptr(out))
        return out

    def render_bus(self, frames, threads=1):
        bus = np.zeros((frames, 2), dtype=np.float64)
        if threads > 1:
            self.L.oracle_bank_render_bus_mt(self.h, frames, _dptr(bus), threads)
        else:
            self.L.oracle_bank_render_bus(self.h, frames, _dptr(bus))
        return bus

    def render_bus_blocks(self, frames, blocks, threads):
        """`blocks` consecutive blocks with PERSISTENT workers: each of `threads` threads owns its share of the voices from one spawn
        to one join (oracle_bank_render_bus_blocks_mt).  Returns bus[blocks * frames][2]."""
        bus = np.zeros((blocks * frames, 2), dtype=np.float64)
        self.L.oracle_bank_render_bus_blocks_mt(self.h, frames, blocks, _dptr(bus), threads)
        return bus

    def release(self):
        """Hand ownership to a Graph."""
        h, self.h = self.h, None
        return h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_bank_destroy(self.h)
            self.h = None


class Fx:
    def __init__(self, kind, params, sr=T.DEFAULT_SAMPLE_RATE):
        self.L = lib()
        self.n = len(params)
        self.h = self.L.oracle_fx_create(kind, params, self.n, sr)

    def set_params(self, params):
        self.L.oracle_fx_set_params(self.h, params, len(params))

    def process(self, block):
        """block: f64 [2][frames][n], modified in place and returned."""
        assert block.dtype == np.float64 and block.flags.c_contiguous and block.shape[2] == self.n
        self.L.oracle_fx_process(self.h, _dptr(block), block.shape[1])
        return block

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_fx_destroy(self.h)
            self.h = None


def mix(block, bus=None):
    """bus[f][ch] (+)= sum_v block[ch][f][v]."""
    L = lib()
    _, frames, n = block.shape
    acc = bus is not None
    if bus is None:
        bus = np.zeros((frames, 2), dtype=np.float64)
    L.oracle_mix(_dptr(np.ascontiguousarray(block)), n, frames, _dptr(bus), 1 if acc else 0)
    return bus


class Graph:
    """Orchestrator patch graph with the reference's per-frame DFS gather (uid 0 = main mixer)."""
    MAIN_MIXER = 0

    def __init__(self, sr=T.DEFAULT_SAMPLE_RATE):
        self.L = lib()
        self.h = self.L.oracle_graph_create(sr)

    def add_source(self, level):
        return self.L.oracle_graph_add_source_const(self.h, level)

    def add_toy_effect(self):
        return self.L.oracle_graph_add_toy_effect(self.h)

    def add_effect(self, kind, params):
        return self.L.oracle_graph_add_effect(self.h, kind, C.byref(params))

    def add_instrument(self, bank):
        return self.L.oracle_graph_add_instrument(self.h, bank.release())

    def patch(self, source, sink):
        return self.L.oracle_graph_patch(self.h, source, sink)

    def patch_chain_to_main_mixer(self, uids):
        """Orchestrator::patch_chain_to_main_mixer, orchestrator.rs:306-325."""
        chain = list(uids) + [self.MAIN_MIXER]
        for a, b in zip(chain[:-1], chain[1:]):
            if self.patch(a, b) != 0:
                return -1
        return 0

    def unpatch_all(self):
        self.L.oracle_graph_unpatch_all(self.h)

    def note_events(self, uid, ev):
        self.L.oracle_graph_note_events(self.h, uid, ev, len(ev))

    def gather(self, frames):
        bus = np.zeros((frames, 2), dtype=np.float64)
        self.L.oracle_graph_gather(self.h, frames, _dptr(bus))
        return bus

    # automation (ControlTrip) and the block loop (Orchestrator::tick)
    STEP_FLAT, STEP_SLOPE, STEP_LOGARITHMIC, STEP_EXPONENTIAL = range(4)

    def set_bpm(self, bpm):
        self.L.oracle_graph_set_bpm(self.h, bpm)

    def skip_to_start(self):
        self.L.oracle_graph_skip_to_start(self.h)

    def add_control_trip(self, target_uid, control_index, start_beat=0.0):
        t = self.L.oracle_graph_add_control_trip(self.h, target_uid, control_index, start_beat)
        assert t >= 0
        return t

    def trip_add_step(self, trip, kind, start, end, beats):
        assert self.L.oracle_graph_trip_add_step(self.h, trip, kind, start, end, beats) == 0

    def tick(self, frames):
        """One block: trips' values at the block start → effect parameters, gather, clock += frames."""
        bus = np.zeros((frames, 2), dtype=np.float64)
        self.L.oracle_graph_tick(self.h, frames, _dptr(bus))
        return bus

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_graph_destroy(self.h)
            self.h = None
