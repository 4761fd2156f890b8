"""CPU tier: the host side's untrusted-input readers under AddressSanitizer + UndefinedBehaviorSanitizer, and a fuzz pass.

`groove_amd/host/json5.hpp` (a hand-written JSON5 reader) and `project.cpp` (projects, Welsh patches, WAV headers) read files a
user hands them; the reference gets memory safety from Rust (`/root/reference/settings/src/songs.rs:84-89`: `json5::from_str`) and
warns-and-continues on bad content (`songs.rs:136-139, 152-156`).  The C ABI's promise is "an error string, never a crash"
(`include/groove_hip.h`).  Checked here with the parser half built as a program of its own (`make -C groove_amd/host asan` →
`parse_check_asan`, no device library):
  (a) every project and every Welsh patch the reference ships, read in place (only when /root/reference exists: this container),
      and the committed synthetic project everywhere;
  (b) truncated, bit-flipped, deeply nested and hostile-valued variants of the committed project, of a patch and of a WAV file:
      every case answers "ok" or "error: ...", the process exits 0, the sanitizers stay silent;
  (c) the oracle's known-answer tests against `oracle/liboracle_asan.so` (a child pytest with the sanitizer runtime preloaded).
Never on the GPU build (sanitizers are not available on the GPU pool)."""
import glob
import os
import random
import struct
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(REPO, "groove_amd", "host")
EXE = os.path.join(HOST, "parse_check_asan")
SYNTHETIC = os.path.join(REPO, "tests", "data", "synthetic-kit-sweep.json5")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0:exitcode=97", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=98"}


@pytest.fixture(scope="module")
def exe():
    r = subprocess.run(["make", "-s", "-C", HOST, "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def run(exe, mode, files, assets=None, chunk=200):
    """All `files` through parse_check_asan; returns the per-file answers.  Fails on a crash or a sanitizer report."""
    answers = []
    for i in range(0, len(files), chunk):
        part = files[i:i + chunk]
        cmd = [exe, mode] + ([assets or "-"] if mode == "project" else []) + part
        r = subprocess.run(cmd, capture_output=True, text=True, errors="replace", timeout=600, env=dict(os.environ, **SAN_ENV))
        assert r.returncode == 0, (r.returncode, r.stderr[-4000:])
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
        lines = r.stdout.splitlines()
        assert len(lines) == len(part), (len(lines), len(part))
        answers += lines
    assert all(a.startswith("ok") or a.startswith("error: ") for a in answers), [a for a in answers if not a.startswith(("ok", "error: "))][:5]
    return answers


def test_committed_project_parses_clean_under_the_sanitizers(exe):
    (a,) = run(exe, "project", [SYNTHETIC])
    assert a.startswith("ok "), a
    devices, notes = int(a.split()[1]), int(a.split()[2])
    assert devices >= 2 and notes > 10


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_every_reference_project_and_patch_under_the_sanitizers(exe):
    projects = sorted(glob.glob(os.path.join(REF, "projects", "**", "*.json*"), recursive=True))
    patches = sorted(glob.glob(os.path.join(REF, "assets", "patches", "welsh", "*.json")))
    assert len(projects) >= 90 and len(patches) >= 100
    pa = run(exe, "project", projects, assets=os.path.join(REF, "assets"))
    assert sum(a.startswith("ok") for a in pa) >= 85, [(p, a) for p, a in zip(projects, pa) if not a.startswith("ok")][:5]
    ta = run(exe, "patch", patches)
    assert all(a == "ok" for a in ta), [(p, a) for p, a in zip(patches, ta) if a != "ok"][:5]
    wavs = sorted(glob.glob(os.path.join(REF, "assets", "samples", "**", "*.wav"), recursive=True))[:40]
    if wavs:
        wa = run(exe, "wav", wavs)
        assert sum(a.startswith("ok") for a in wa) >= len(wavs) // 2


def _variants(data: bytes, rng, n_trunc=120, n_flip=250, n_splice=60):
    out = []
    for k in range(n_trunc):                      # truncations spread over the whole text (and the first 40 bytes one by one)
        out.append(data[:k] if k < 40 else data[:(len(data) * k) // n_trunc])
    for _ in range(n_flip):                       # one to four flipped bits
        b = bytearray(data)
        for _ in range(rng.randint(1, 4)):
            i = rng.randrange(len(b))
            b[i] ^= 1 << rng.randrange(8)
        out.append(bytes(b))
    for _ in range(n_splice):                     # a slice duplicated, dropped, or replaced by structural characters
        i, j = sorted(rng.randrange(len(data)) for _ in range(2))
        kind = rng.randrange(3)
        out.append(data[:i] + (data[i:j] * 2 if kind == 0 else b"" if kind == 1 else bytes(rng.choice(b"{}[],:\"'\\/*-+.eExX0") for _ in range(8))) + data[j:])
    return out


HOSTILE = [
    b"", b" ", b"{", b"}", b"[", b"]", b"{]", b"[}", b"nul", b"tru", b"-", b"+", b".", b"0x", b"0xZZ", b"1e", b"1e99999", b"-Infinity", b"NaN",
    b'"', b"'", b'"\\', b'"\\u', b'"\\u12', b'"\\uD800"', b'"\\x"', b"/*", b"/* *", b"//", b"{a}", b"{a:}", b"{:1}", b"{a:1,,}", b"[,]", b"[1 2]",
    b"[" * 100000, b"{a:" * 50000, b"[" * 300 + b"]" * 300, b"[" * 200 + b"1" + b"]" * 200, b"/*" * 50000, b'"' + b"\\" * 99999,
    b"\xff\xfe\x00\x00", b"\x00" * 64, b"{" + b'"k":1,' * 20000 + b"}", b"[" + b"1e308," * 20000 + b"]",
    b'{"title": 1, "clock": [], "devices": {}, "patch-cables": 7, "patterns": "x", "tracks": null, "trips": 1.5}',
    b'{"clock": {"bpm": NaN, "time-signature": [0, 0]}}', b'{"clock": {"bpm": -1e308, "time-signature": [1e99, -4]}}',
    b'{"clock": {"bpm": 1e-300, "time-signature": [4, 4]}, "patterns": [{"id": "p", "note-value": "sixteenth", "notes": [["1","2"],[]]}], "tracks": [{"id":"t","midi-channel": 1e30, "patterns": ["p","p","nope"]}]}',
    b'{"devices": [{"instrument": ["x", {"midi-in": -5, "welsh": "../../../../etc/passwd"}]}, {"effect": ["y", {"gain": {"ceiling": "loud"}}]}, {"effect": []}, {"instrument": 3}, 17, null]}',
    b'{"devices": [{"effect": ["f", {"filter-low-pass-24db": {"cutoff": -1, "passband-ripple": 1e400}}]}, {"effect": ["b", {"bitcrusher": {"bits-to-crush": 1e20}}]}, {"effect": ["d", {"delay": {"delay": -3}}]}]}',
    b'{"patterns": [{"id": "a", "note-value": "bogus", "notes": [[' + b'"60",' * 5000 + b'"61"]]}], "tracks": [{"id": "t", "midi-channel": 0, "patterns": [' + b'"a",' * 3000 + b'"a"]}]}',
    b'{"trips": [{"id": "t", "target": {"id": "nobody", "param": "cutoff"}, "steps": [{"flat": {"value": NaN}}, {"slope": {"start": 1e308, "end": -1e308}}, {"bogus": 1}, 5], "note-value": "whole"}]}',
    b'{"trips": [{"id": "t", "target": 4, "steps": "many"}], "patch-cables": [["a"], [], [1, 2, 3], "x", ["main-mixer", "main-mixer"]]}',
]


def test_fuzzed_projects_answer_ok_or_error_never_crash(exe, tmp_path):
    rng = random.Random(20261004)
    data = open(SYNTHETIC, "rb").read()
    cases = _variants(data, rng) + HOSTILE
    files = []
    for i, c in enumerate(cases):
        p = tmp_path / f"case{i:04d}.json5"
        p.write_bytes(c)
        files.append(str(p))
    answers = run(exe, "project", files)
    n_ok = sum(a.startswith("ok") for a in answers)
    assert len(answers) == len(cases) and 20 < n_ok < len(cases) - 150     # many variants still parse, many must not
    # the reference warns and continues on bad CONTENT (unknown devices, dangling cables): those cases answer ok
    hostile_answers = answers[-len(HOSTILE):]
    assert hostile_answers[HOSTILE.index(b"[" * 100000)].startswith("error: ") and "nest" in hostile_answers[HOSTILE.index(b"[" * 100000)]
    assert all(a.startswith("error: ") for a in hostile_answers[:40])          # the malformed texts


PATCH_KEYS = b'''{"name": "Fuzz", "oscillator_1": {"waveform": "Sawtooth", "tune": {"Osc": {"octave": 0, "semi": 0, "cent": 0}}, "mix": 1.0},
 "oscillator_2": {"waveform": {"PulseWidth": 0.3}, "tune": {"Note": 60}, "mix": 0.5}, "oscillator_2_track": true, "oscillator_2_sync": false,
 "noise": 0.0, "lfo": {"routing": "Pitch", "waveform": "Sine", "frequency": 5.0, "depth": {"Pct": 0.1}}, "glide": 0.0, "unison": false, "polyphony": "Multi",
 "filter_type_24db": {"cutoff_hz": 900.0, "cutoff_pct": 0.5}, "filter_type_12db": {"cutoff_hz": 900.0, "cutoff_pct": 0.5}, "filter_resonance": 0.1, "filter_envelope_weight": 0.5,
 "filter_envelope": {"attack": 0.1, "decay": 0.2, "sustain": 0.5, "release": 0.3}, "amp_envelope": {"attack": 0.01, "decay": 0.2, "sustain": 0.8, "release": 0.2}}'''


def test_fuzzed_patches_and_wav_headers_never_crash(exe, tmp_path):
    rng = random.Random(5)
    src = PATCH_KEYS
    if os.path.isdir(REF):   # a real patch's shape when the tree is here; the constant above otherwise
        some = sorted(glob.glob(os.path.join(REF, "assets", "patches", "welsh", "*.json")))
        if some:
            src = open(some[len(some) // 2], "rb").read()
    files = []
    for i, c in enumerate(_variants(src, rng, 60, 150, 40) + HOSTILE[:46]):
        p = tmp_path / f"patch{i:04d}.json"
        p.write_bytes(c)
        files.append(str(p))
    run(exe, "patch", files)
    # WAV: a valid 16-bit stereo file, then its truncations, flipped header bits, and lying size fields
    frames = 500
    pcm = struct.pack("<%dh" % (2 * frames), *[((i * 37) % 2000) - 1000 for i in range(2 * frames)])
    def wav(fmt=1, ch=2, rate=44100, bits=16, data=pcm, riff_size=None, data_size=None, extra=b""):
        body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, fmt, ch, rate, (rate * ch * bits // 8) & 0xFFFFFFFF, (ch * bits // 8) & 0xFFFF, bits) + extra + \
               b"data" + struct.pack("<I", len(data) if data_size is None else data_size) + data
        return b"RIFF" + struct.pack("<I", len(body) if riff_size is None else riff_size) + body
    good = wav()
    cases = [good, wav(ch=0), wav(bits=0), wav(bits=7), wav(bits=64), wav(ch=65535), wav(rate=0), wav(fmt=3, bits=32), wav(fmt=0xFFFE),
             wav(data_size=0xFFFFFFFF), wav(data_size=0x7FFFFFFF), wav(riff_size=0), wav(riff_size=0xFFFFFFFF), wav(data=b""), wav(data=pcm[:3]),
             wav(extra=b"LIST" + struct.pack("<I", 0xFFFFFFF0)), wav(extra=b"junk" + struct.pack("<I", 3) + b"abc"), b"RIFF", b"RIFF\x00\x00\x00\x00WAVE", b""]
    cases += [good[:k] for k in range(0, 60)] + [good[:len(good) - k] for k in (1, 2, 3, 5, 999)]
    for _ in range(200):
        b = bytearray(good)
        for _ in range(rng.randint(1, 3)):
            b[rng.randrange(48)] ^= 1 << rng.randrange(8)     # the header
        cases.append(bytes(b))
    files = []
    for i, c in enumerate(cases):
        p = tmp_path / f"w{i:04d}.wav"
        p.write_bytes(c)
        files.append(str(p))
    answers = run(exe, "wav", files + [str(tmp_path / "does-not-exist.wav")])
    assert answers[0] == f"ok {frames} 44100" and answers[-1].startswith("error: ")


def test_oracle_known_answers_under_the_sanitizers():
    """tests/test_oracle_kat.py + test_golden.py against oracle/liboracle_asan.so, in a child pytest with the sanitizer runtimes
    preloaded (a sanitized shared object cannot be loaded into a plain python otherwise)."""
    r = subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    pre = []
    for name in ("libasan.so", "libubsan.so"):
        path = subprocess.run(["g++", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        if os.path.isabs(path) and os.path.exists(path):
            pre.append(os.path.realpath(path))
    if not pre:
        pytest.skip("no sanitizer runtime found beside g++")
    env = dict(os.environ, LD_PRELOAD=":".join(pre), GROOVE_ORACLE_LIB="liboracle_asan.so",
               ASAN_OPTIONS="detect_leaks=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", os.path.join(REPO, "tests", "test_oracle_kat.py"),
                        os.path.join(REPO, "tests", "test_golden.py")], env=env, capture_output=True, text=True, timeout=1200, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert " passed" in r.stdout
