"""The reference's own Welsh patch library (assets/patches/welsh/*.json under /root/reference: DATA, read where it lies, in this container only)
derived by the host layer (host/project.cpp, the restatement of settings/src/patches.rs:87-170) and played through the DEVICE's frame text
compiled for the CPU (tests/emul) against the f64 oracle: per patch, the worst of four voices' RMS error, f64 and fp32 filter kinds.
    python3 tools/reference_patches_emul.py"""
import ctypes as C, glob, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T
from tests.emul import emul as E
from oracle import oracle as O
O.build(ref=False)
L = C.CDLL(os.path.join(REPO, "groove_amd", "host", "libgroove_host.so"))
L.gh_welsh_params_from_patch_json.argtypes = [C.c_char_p, C.POINTER(T.WelshParams), C.c_char_p, C.c_size_t]
SR = int(os.environ.get("SR", "44100"))
keys = np.array([31, 50, 64, 86], dtype=np.uint8)
lanes = np.arange(4, dtype=np.uint32)
on, off = T.note_events_np(lanes, keys, True), T.note_events_np(lanes, keys, False)
rows = []
for f in sorted(glob.glob("/root/reference/assets/patches/welsh/*.json")):
    p = T.WelshParams(); err = C.create_string_buffer(512)
    if L.gh_welsh_params_from_patch_json(open(f).read().encode(), C.byref(p), err, 512):
        continue
    params = (T.WelshParams * 4)(p, p, p, p)
    res = []
    for kind in (False, True):
        bo, be = O.Bank.welsh(params, sr=SR), E.Bank.welsh(params, SR)
        if kind: be.set_f32_kind(True)
        o, e = [], []
        for b in range(60):
            if b == 0: bo.note_events(on); be.note_events(on)
            if b == 40: bo.note_events(off); be.note_events(off)
            o.append(bo.render(256)); e.append(be.render(256))
        o = np.concatenate(o, axis=1); e = np.concatenate(e, axis=1).astype(np.float64)
        res.append((np.sqrt(np.mean((e - o) ** 2, axis=(0, 1))), np.sqrt(np.mean(o ** 2, axis=(0, 1))), np.isfinite(e).all()))
    rows.append((os.path.basename(f), res))
    print("%-28s f64 %.2e (voice %d)  f32 kind %.2e  level %.3f  routing %d ripple %.2f cutoff %.0f" % (
        os.path.basename(f)[:28], res[0][0].max(), int(res[0][0].argmax()), res[1][0].max(), res[0][1].max(), p.lfo_routing, p.filter_passband_ripple, p.filter_cutoff_hz), flush=True)
w64 = max(r[1][0][0].max() for r in rows); w32 = max(r[1][1][0].max() for r in rows)
print(len(rows), "patches; worst f64 kind %.2e, worst f32 kind %.2e" % (w64, w32))
