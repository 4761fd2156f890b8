"""ctypes binding of tests/emul/libemul.so (host build of the device DSP text; test harness only)."""
import ctypes as C
import os
import subprocess
import numpy as np
from groove_amd import abi_types as T

HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)


def build():
    src = os.path.join(HERE, "emul.cpp")
    out = os.path.join(HERE, "libemul.so")
    deps = [src] + [os.path.join(HERE, "..", "..", "groove_amd", "csrc", f) for f in ("dsp_core.h", "derive.h", "welsh_tp.h", "welsh_split.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        # -ffp-contract=off: fmaf() calls stay fused, plain a*b+c stays unfused, like hipcc's default for explicit code
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src], check=True)
    return out


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, u32 = C.c_void_p, C.c_uint32
        L.emul_welsh_create.restype = vp; L.emul_welsh_create.argtypes = [C.POINTER(T.WelshParams), u32, u32]
        L.emul_fm_create.restype = vp; L.emul_fm_create.argtypes = [C.POINTER(T.FmParams), u32, u32]
        L.emul_sampler_create.restype = vp
        L.emul_sampler_create.argtypes = [_fp, C.c_uint64, C.POINTER(T.SampleDesc), u32, C.POINTER(T.SamplerParams), u32, u32]
        L.emul_bank_destroy.argtypes = [vp]
        L.emul_set_generic_lfo.argtypes = [vp, C.c_int]
        L.emul_set_f32_filter.argtypes = [C.c_int]
        L.emul_set_f32_kind.restype = u32; L.emul_set_f32_kind.argtypes = [vp, C.c_int]
        L.emul_filter_f32_error.restype = C.c_double; L.emul_filter_f32_error.argtypes = [C.POINTER(T.WelshParams), u32]
        L.emul_set_segmented.argtypes = [vp, C.c_int]
        L.emul_set_lfo_look_ahead.argtypes = [vp, C.c_int]
        L.emul_set_time_parallel.argtypes = [vp, C.c_int]
        L.emul_set_role_split.argtypes = [vp, C.c_int]
        L.emul_bank_note_events.argtypes = [vp, C.POINTER(T.NoteEvent), u32]
        L.emul_bank_render.argtypes = [vp, u32, _fp]
        L.emul_bank_min_segment.restype = u32; L.emul_bank_min_segment.argtypes = [vp]
        L.emul_segment_begin_of.restype = u32
        L.emul_segment_begin_of.argtypes = [C.POINTER(T.WelshParams), u32, C.POINTER(u32), C.POINTER(u32)]
        L.emul_bitcrush.restype = C.c_float; L.emul_bitcrush.argtypes = [C.c_float, u32]
        L.emul_lp24_coef_both.argtypes = [C.c_double, C.c_float, C.c_float, C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def segment_begin_of(params, amp, fil, sr=T.DEFAULT_SAMPLE_RATE):
    """welsh_segment_begin for a voice of patch `params` whose envelope records are (state, n, N) = amp / fil."""
    a = (C.c_uint32 * 3)(*amp)
    f = (C.c_uint32 * 3)(*fil)
    return lib().emul_segment_begin_of(C.byref(params), sr, a, f)


class Bank:
    def __init__(self, h, n):
        self.h, self.n = h, n

    @classmethod
    def welsh(cls, params, sr=T.DEFAULT_SAMPLE_RATE):
        return cls(lib().emul_welsh_create(params, len(params), sr), len(params))

    @classmethod
    def fm(cls, params, sr=T.DEFAULT_SAMPLE_RATE):
        return cls(lib().emul_fm_create(params, len(params), sr), len(params))

    @classmethod
    def sampler(cls, pcm, descs, params, sr=T.DEFAULT_SAMPLE_RATE):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        return cls(lib().emul_sampler_create(pcm.ctypes.data_as(_fp), pcm.size, descs, len(descs), params, len(params), sr), len(params))

    def note_events(self, ev):
        lib().emul_bank_note_events(self.h, ev, len(ev))

    def set_segmented(self, on):
        """True (default): boundary-free segments as in the uniform kernels; False: every frame checked (per-lane kernel)."""
        lib().emul_set_segmented(self.h, 1 if on else 0)

    def set_lfo_look_ahead(self, on):
        """True: the smooth-f64 kinds' frames take what the LFO does to the oscillators from an exact evaluation at the frame's phase
        (kernels.h "LFO look-ahead": what a wave whose voices share the LFO's phase does on the device) instead of advancing recurrences."""
        lib().emul_set_lfo_look_ahead(self.h, 1 if on else 0)

    def set_time_parallel(self, on):
        """True: Welsh voices through the time-parallel form (welsh_tp.h: 64 lanes x 4 frames, affine-map scan)."""
        lib().emul_set_time_parallel(self.h, 1 if on else 0)

    def set_role_split(self, on):
        """True: every frame role by role, with what the role-split kernel (welsh_split.h) passes between its wavefronts."""
        lib().emul_set_role_split(self.h, int(on))  # (4: the four-role form — front in two halves, coefficient quotients in role B)

    def set_f32_kind(self, on):
        """True: the per-kind kernels' arithmetic — patches measured safe by the host criterion (WF_FILTER_F32) filter in fp32.
        Returns how many voices carry the flag."""
        return lib().emul_set_f32_kind(self.h, 1 if on else 0)

    def set_generic_lfo(self, on):
        """True: exact per-frame f64 LFO (per-lane kernel); False: block-seeded recurrences where promised."""
        lib().emul_set_generic_lfo(self.h, 1 if on else 0)

    @property
    def min_segment(self):
        """Smallest frames-to-next-boundary any voice has reported at a segment start so far (the kernels rely on >= 1)."""
        return lib().emul_bank_min_segment(self.h)

    def render(self, frames):
        out = np.zeros((2, frames, self.n), dtype=np.float32)
        lib().emul_bank_render(self.h, frames, out.ctypes.data_as(_fp))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().emul_bank_destroy(self.h)
            self.h = None
