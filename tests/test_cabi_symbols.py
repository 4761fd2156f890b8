"""CPU tier: libgroove_hip.so loads without a GPU and exports every symbol include/groove_hip.h
declares; groove_init fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import pytest

from groove_amd import lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "groove_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(groove_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert _declared() == sorted(lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    so = C.CDLL(lib.LIB_PATH)
    missing = [s for s in _declared() if not hasattr(so, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_a_device():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except Exception:
        pass
    from groove_amd import entities as E
    with pytest.raises(lib.GrooveError, match="no HIP device|no CPU path"):
        E.Context(0)


def test_param_struct_sizes_match_the_c_headers(tmp_path):
    """ctypes mirrors vs the C compiler's view of include/groove_types.h."""
    from groove_amd import abi_types as T
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "groove_types.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(groove_envelope_params),sizeof(groove_oscillator_params),sizeof(groove_welsh_params),'
                   'sizeof(groove_fm_params),sizeof(groove_sample_desc),sizeof(groove_sampler_params),'
                   'sizeof(groove_note_event),sizeof(groove_fx_params));return 0;}\n')
    import subprocess
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(t) for t in (T.EnvelopeParams, T.OscillatorParams, T.WelshParams, T.FmParams, T.SampleDesc,
                                  T.SamplerParams, T.NoteEvent, T.FxParams)]
    assert got == want


def test_integration_doc_binds_every_declared_symbol_and_struct():
    """INTEGRATION.md's `extern "C"` block — the reference-side binding a maintainer would add — names every function the
    header declares (and nothing else), with the same number of arguments, and spells out every parameter struct."""
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = doc[doc.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    bound = dict(re.findall(r"pub fn (groove_[a-z0-9_]+)\((.*?)\)(?: -> [^;]+)?;", block))
    assert sorted(bound) == _declared()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "groove_hip.h")).read(), flags=re.S)
    for name, args in bound.items():
        c_args = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S).group(1)
        n_c = 0 if c_args.strip() in ("", "void") else c_args.count(",") + 1
        n_rust = 0 if not args.strip() else args.count(",") + 1
        assert n_c == n_rust, (name, c_args, args)
    types = re.sub(r"/\*.*?\*/", "", open(os.path.join(REPO, "include", "groove_types.h")).read(), flags=re.S)
    for struct in re.findall(r"\}\s*(groove_[a-z_]+_(?:params|desc|event))\s*;", types):
        assert f"pub struct {struct} " in doc, struct
