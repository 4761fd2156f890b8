"""Class proportions of the reference's Welsh patch library (assets/patches/welsh/*.json under /root/reference: DATA, read where it
lies, in the development container only — nothing of a file is copied; the output is counts).

Every file goes through the host layer's derivation (host/project.cpp welsh_params_from_patch_json = settings/src/patches.rs:87-170)
and the library's own kind rule (csrc/dsp_core.h welsh_base_kind / welsh_body_classes via tests/emul): which of the six base kinds
(LFO mode x retune) its voices run in, the LFO's waveform and routing when it moves a waveform edge, the oscillators' classes.
`groove_amd/patches.py` LIBRARY_* holds the table this prints (the workload `welsh-1m-library`: synthetic patches whose CLASS
proportions follow these counts).

    python3 tools/library_proportions.py [--json]
"""
import collections
import ctypes as C
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from groove_amd import abi_types as T  # noqa: E402
from tests.emul import emul as E  # noqa: E402

BASE = ["F32-static", "F32-retune", "smooth-static", "smooth-retune", "exact-f64-static", "exact-f64-retune"]
WAVE = ["none", "sine", "square", "pulse-width", "triangle", "sawtooth", "noise", "dbg0", "dbgmax", "dbgmin", "triangle-sine"]
ROUTE = ["none", "amplitude", "pitch", "pulse-width", "cutoff", "pitch-osc2", "pw-osc1", "pw-osc2", "resonance", "cutoff-amp"]
CLS = ["any", "pulse", "saw", "triangle", "sine", "unused"]


def classify(p, sr=44100):
    E.build()
    lib = E._lib() if hasattr(E, "_lib") else C.CDLL(os.path.join(REPO, "tests", "emul", "libemul.so"))
    lib.emul_welsh_classify.argtypes = [C.POINTER(T.WelshParams), C.c_uint32, C.POINTER(C.c_uint32)]
    out = (C.c_uint32 * 6)()
    lib.emul_welsh_classify(C.byref(p), sr, out)
    return list(out)


def main():
    host = C.CDLL(os.path.join(REPO, "groove_amd", "host", "libgroove_host.so"))
    host.gh_welsh_params_from_patch_json.argtypes = [C.c_char_p, C.POINTER(T.WelshParams), C.c_char_p, C.c_size_t]
    files = sorted(glob.glob("/root/reference/assets/patches/welsh/*.json"))
    base, edge, f32, oscs, lfo_cls, skipped = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), 0
    sync = retune_env = 0
    for f in files:
        p, err = T.WelshParams(), C.create_string_buffer(512)
        if host.gh_welsh_params_from_patch_json(open(f).read().encode(), C.byref(p), err, 512):
            skipped += 1  # (oscillator-2-track false without a note: the reference panics on these too, patches.rs:98)
            continue
        k = classify(p)
        base[BASE[k[0]]] += 1
        f32[(BASE[k[0]], bool(k[4]))] += 1
        oscs[(CLS[k[2]], CLS[k[3]])] += 1
        lfo_cls[(BASE[k[0]], CLS[k[1]])] += 1
        if p.lfo_routing in (T.LFO_PITCH, T.LFO_PULSE_WIDTH, T.LFO_PITCH_OSC2, T.LFO_PW_OSC1, T.LFO_PW_OSC2, T.LFO_RESONANCE) or k[0] >= 4:
            edge[(BASE[k[0]], ROUTE[p.lfo_routing], WAVE[p.lfo_waveform])] += 1
        sync += int(p.oscillator_2_sync != 0)
        retune_env += int(p.filter_cutoff_end != 0.0)
    n = sum(base.values())
    table = {
        "files": len(files), "derived": n, "skipped": skipped,
        "base_kind": {k: base.get(k, 0) for k in BASE},
        "base_kind_pct": {k: round(100.0 * base.get(k, 0) / n, 1) for k in BASE},
        "edge_moving_lfo": {" / ".join(k): v for k, v in sorted(edge.items())},
        "fp32_filter_ok": {f"{k[0]} / {'fp32' if k[1] else 'f64'}": v for k, v in sorted(f32.items())},
        "lfo_class": {" / ".join(k): v for k, v in sorted(lfo_cls.items())},
        "oscillator_classes": {" x ".join(k): v for k, v in sorted(oscs.items(), key=lambda kv: -kv[1])},
        "hard_sync": sync, "envelope_retune": retune_env,
    }
    if "--json" in sys.argv:
        print(json.dumps(table, indent=1))
        return
    for k, v in table.items():
        print(f"{k}: {json.dumps(v)}")


if __name__ == "__main__":
    main()
