"""ctypes loader for libgroove_hip.so (the C ABI in include/groove_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C groove_amd`.  There is
no fallback: if the shared object is missing, or the machine has no HIP device, loading or
`groove_init` fails loudly.
"""
import ctypes as C
import os

from . import abi_types as T

HERE = os.path.dirname(os.path.abspath(__file__))
# GROOVE_LIB_PATH: another BUILD of the same library (tools/ab_bench.sh variants, the diagnostic builds of csrc/diag.h) —
# same ABI, same loader; there is still no other implementation to fall back to.
LIB_PATH = os.environ.get("GROOVE_LIB_PATH") or os.path.join(HERE, "libgroove_hip.so")

# Every symbol include/groove_hip.h declares: name → (restype, argtypes).
_vp, _u32, _i, _d = C.c_void_p, C.c_uint32, C.c_int, C.c_double
_fp = C.POINTER(C.c_float)
_vpp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "groove_init": (_i, [_i, _vpp]),
    "groove_shutdown": (None, [_vp]),
    "groove_last_error": (C.c_char_p, [_vp]),
    "groove_set_stream": (_i, [_vp, _vp]),
    "groove_synchronize": (_i, [_vp]),
    "groove_set_sync_timeout_ms": (_i, [_vp, _u32]),
    "groove_sync_timeout_ms": (_u32, [_vp]),
    "groove_debug_spin": (_i, [_vp, _i, _u32]),
    "groove_debug_info": (_i, [_vp, C.c_char_p, C.c_size_t]),
    "groove_init_comm": (_i, [_i, C.POINTER(C.c_uint8), _i, _i, _vpp]),
    "groove_block_mark_dirty": (_i, [_vp]),
    "groove_update_sample_rate": (_i, [_vp, _u32]),
    "groove_sample_rate": (_u32, [_vp]),
    "groove_set_time_parallel_max_voices": (_i, [_vp, _u32]),
    "groove_time_parallel_max_voices": (_u32, [_vp]),
    "groove_set_time_parallel_pair_min_voices": (_i, [_vp, _u32]),
    "groove_time_parallel_pair_min_voices": (_u32, [_vp]),
    "groove_set_pipeline_min_waves": (_i, [_vp, _u32]),
    "groove_pipeline_min_waves": (_u32, [_vp]),
    "groove_set_look_ahead": (_i, [_vp, _u32]),
    "groove_look_ahead": (_u32, [_vp]),
    "groove_set_fx_allpass_stream": (_i, [_vp, _i]),
    "groove_fx_allpass_stream": (_i, [_vp]),
    "groove_set_split_max_waves": (_i, [_vp, _u32]),
    "groove_split_max_waves": (_u32, [_vp]),
    "groove_bank_kernel_form": (C.c_char_p, [_vp, _u32, _i]),
    "groove_event_create": (_i, [_vp, _vpp]),
    "groove_event_destroy": (_i, [_vp, _vp]),
    "groove_event_record": (_i, [_vp, _vp]),
    "groove_event_elapsed_ms": (_i, [_vp, _vp, _vp, _fp]),
    "groove_block_create": (_i, [_vp, _u32, _u32, _vpp]),
    "groove_block_destroy": (_i, [_vp]),
    "groove_block_device_ptr": (_vp, [_vp]),
    "groove_block_lanes": (_u32, [_vp]),
    "groove_block_frames_cap": (_u32, [_vp]),
    "groove_block_upload": (_i, [_vp, _fp, _u32]),
    "groove_block_download": (_i, [_vp, _fp, _u32]),
    "groove_block_accumulate": (_i, [_vp, _vp, _u32, _i]),
    "groove_block_zero": (_i, [_vp]),
    "groove_welsh_create": (_i, [_vp, C.POINTER(T.WelshParams), _u32, _vpp]),
    "groove_fm_create": (_i, [_vp, C.POINTER(T.FmParams), _u32, _vpp]),
    "groove_sampler_create": (_i, [_vp, _fp, C.c_uint64, C.POINTER(T.SampleDesc), _u32, C.POINTER(T.SamplerParams), _u32, _vpp]),
    "groove_bank_destroy": (_i, [_vp]),
    "groove_bank_voices": (_u32, [_vp]),
    "groove_bank_note_events": (_i, [_vp, C.POINTER(T.NoteEvent), _u32]),
    "groove_bank_set_param": (_i, [_vp, _u32, _u32, _d]),
    "groove_bank_render": (_i, [_vp, _u32, _vp]),
    "groove_bank_render_async": (_i, [_vp, _u32, _vp]),
    "groove_block_acquire": (_i, [_vp]),
    "groove_block_release": (_i, [_vp]),
    "groove_block_wait_ready": (_i, [_vp]),
    "groove_block_wait_released": (_i, [_vp]),
    "groove_mix_deferred": (_i, [_vp, _vp, _u32, _vp, _i]),
    "groove_bank_render_mix": (_i, [_vp, _u32, _vp, _i]),
    "groove_bank_render_mix_deferred": (_i, [_vp, _u32, _vp, _i]),
    "groove_banks_render_mix_deferred": (_i, [_vp, _vpp, _u32, _u32, _vp, _i]),
    "groove_bus_flush": (_i, [_vp]),
    "groove_bank_render_mix_paced": (_i, [_vp, _u32, _vp, _i]),
    "groove_bank_reset": (_i, [_vp]),
    "groove_bank_state_words": (_u32, [_vp]),
    "groove_bank_download_state": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "groove_fx_create": (_i, [_vp, _u32, C.POINTER(T.FxParams), _u32, _vpp]),
    "groove_fx_destroy": (_i, [_vp]),
    "groove_fx_reset": (_i, [_vp]),
    "groove_fx_process": (_i, [_vp, _vp, _u32]),
    "groove_fx_chain_process": (_i, [_vp, _u32, _vp, _u32]),
    "groove_fx_chain_process_async": (_i, [_vp, _u32, _vp, _u32, C.POINTER(_u32)]),
    "groove_bank_render_chain_async": (_i, [_vp, _u32, _vp, _vp, _u32, C.POINTER(_u32)]),
    "groove_fx_set_param": (_i, [_vp, _u32, _u32, _d]),
    "groove_fx_set_params": (_i, [_vp, C.POINTER(T.FxParams), _u32]),
    "groove_mix": (_i, [_vp, _vpp, _u32, _u32, _vp, _i]),
    "groove_bus_create": (_i, [_vp, C.c_size_t, _vpp]),
    "groove_bus_destroy": (_i, [_vp, _vp]),
    "groove_bus_zero": (_i, [_vp, _vp, C.c_size_t]),
    "groove_download": (_i, [_vp, _vp, _fp, C.c_size_t]),
    "groove_upload": (_i, [_vp, _vp, _fp, C.c_size_t]),
    "groove_bus_to_i16": (_i, [_vp, _vp, C.c_size_t, C.POINTER(C.c_int16)]),
    "groove_comm_unique_id": (_i, [_vp, C.POINTER(C.c_uint8)]),
    "groove_comm_init": (_i, [_vp, C.POINTER(C.c_uint8), _i, _i]),
    "groove_comm_ranks": (_i, [_vp, C.POINTER(C.c_int)]),
    "groove_comm_destroy": (_i, [_vp]),
    "groove_bus_reduce": (_i, [_vp, _vp, C.c_size_t, _i]),
}

_LIB = None


class GrooveError(RuntimeError):
    pass


def load():
    """dlopen libgroove_hip.so and bind every declared symbol.  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise GrooveError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc, ctx=None):
    if rc != 0:
        msg = load().groove_last_error(ctx)
        raise GrooveError(msg.decode() if msg else f"groove call failed with code {rc}")
