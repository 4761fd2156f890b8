"""Host logic of the compiled loader (groove_amd/host/project.cpp): project JSON/JSON5 and Welsh
patch JSON → device-independent description.  CPU tier, no GPU calls.  Reference files are read
only when /root/reference exists (this container); a synthetic project with the same schema
features and content of its own is committed for the GPU box (tests/data/synthetic-kit-sweep.json5)."""
import ctypes as C
import glob
import json
import math
import os

import pytest

from groove_amd import abi_types as T
from groove_amd.host_binding import HOST_LIB

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(HOST_LIB):
        import __graft_entry__ as g
        g.build()
    L = C.CDLL(HOST_LIB)
    L.gh_project_describe.restype = C.c_void_p
    L.gh_project_describe.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.gh_project_describe_text.restype = C.c_void_p
    L.gh_project_describe_text.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.gh_free.argtypes = [C.c_void_p]
    L.gh_welsh_params_from_patch_json.argtypes = [C.c_char_p, C.POINTER(T.WelshParams), C.c_char_p, C.c_size_t]
    return L


def describe(L, path=None, text=None, assets=""):
    err = C.create_string_buffer(512)
    p = L.gh_project_describe(path.encode(), assets.encode(), err, 512) if path else \
        L.gh_project_describe_text(text.encode(), assets.encode(), err, 512)
    if not p:
        raise RuntimeError(err.value.decode())
    s = C.string_at(p).decode()
    L.gh_free(p)
    return json.loads(s)


def check_config1_shape(d):
    assert d["bpm"] == 128 and d["time_signature"] == [4, 4]
    kinds = {x["id"]: x for x in d["devices"]}
    assert kinds["drum-1"]["kind"] == "drumkit" and kinds["drum-1"]["midi_in"] == 10 and kinds["drum-1"]["name"] == "707"
    lp = kinds["low-pass-1"]
    assert lp["effect"] and lp["fx_kind"] == T.FX_BIQUAD_LP24 and lp["cutoff"] == 1000 and abs(lp["passband_ripple"] - 0.8) < 1e-6
    assert d["patch_cables"] == [["drum-1", "low-pass-1", "main-mixer"]]
    assert d["n_notes"] == 2 * (16 + 2 + 4) and d["end_beats"] == 8.0          # two measures of 4/4
    assert d["trips"] == [{"id": "trip-1", "target": "low-pass-1", "param": "cutoff", "steps": 1, "beats": 8.0, "first_kind": 3}]
    assert math.ceil(d["end_beats"] * 60 / d["bpm"] * 44100) == 165375


SYNTHETIC = os.path.join(REPO, "tests", "data", "synthetic-kit-sweep.json5")


def test_synthetic_json5_project(host):
    """The committed synthetic project (the one the GPU box renders end to end): the same schema features as config #1 —
    drumkit on channel 10, a 24 dB low-pass, a patch cable to the main mixer, patterns x tracks, a control trip on the
    cutoff — with content of its own."""
    d = describe(host, path=SYNTHETIC)
    assert d["bpm"] == 96 and d["time_signature"] == [4, 4] and d["warnings"] == 0
    kinds = {x["id"]: x for x in d["devices"]}
    assert kinds["kit"]["kind"] == "drumkit" and kinds["kit"]["midi_in"] == 10 and kinds["kit"]["name"] == "707"
    lp = kinds["sweep-lp"]
    assert lp["effect"] and lp["fx_kind"] == T.FX_BIQUAD_LP24 and lp["cutoff"] == 2400 and abs(lp["passband_ripple"] - 0.55) < 1e-6
    assert d["patch_cables"] == [["kit", "sweep-lp", "main-mixer"]]
    assert d["n_notes"] == 2 * (6 + 3 + 3) + 16 and d["end_beats"] == 12.0      # shuffle, fill, shuffle: three measures of 4/4
    assert d["trips"] == [{"id": "sweep", "target": "sweep-lp", "param": "cutoff", "steps": 2, "beats": 8.0, "first_kind": 2}]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_reference_config1_project_parses(host):
    d = describe(host, path=f"{REF}/projects/demos/effects/drums-filtered-24db.json", assets=f"{REF}/assets")
    check_config1_shape(d)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_reference_demo_projects_parse(host):
    """Every demo project of the current schema generation loads (older-generation files may fail
    on schema, never on syntax)."""
    ok, schema_fail = 0, []
    for f in sorted(glob.glob(f"{REF}/projects/demos/**/*.json*", recursive=True)):
        try:
            d = describe(host, path=f, assets=f"{REF}/assets")
            assert d["devices"] is not None
            ok += 1
        except RuntimeError as e:
            assert "JSON5 parse error" not in str(e), f"{f}: {e}"
            schema_fail.append((os.path.basename(f), str(e)[:80]))
    assert ok >= 30, (ok, schema_fail[:5])


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_fm_demo_projects_of_both_generations_load_their_synth(host):
    """projects/demos/instruments/fm-synthesizer*.json: one file in the current [midi, params] form, five (the beta sweep 0 - 100) with one
    object {"midi-in", "voice": {..}} — each yields its FM synth, its gain and its one note (the beta-100 file is what found the carrier's
    whole-turn wrap: tests/test_gpu_instruments.py::test_fm_index_of_the_reference_demo_projects)."""
    files = sorted(glob.glob(f"{REF}/projects/demos/instruments/fm-synthesizer*.json"))
    assert len(files) == 6
    for f in files:
        d = describe(host, path=f, assets=f"{REF}/assets")
        assert [x["kind"] for x in d["devices"]] == ["fm-synthesizer", "gain"], f
        assert d["warnings"] == 0 and d["n_notes"] >= 1, f


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_the_reference_test_projects(host):
    """projects/tests/: the invalid project "that should fail to load" fails (a device class DeviceSettings does not have, settings/src/lib.rs:
    42-46); both WAV-loading projects yield their sampler and their note (the stereo one in the one-object form of the generation before)."""
    with pytest.raises(RuntimeError, match="unknown device class 'instrumentx'"):
        describe(host, path=f"{REF}/projects/tests/invalid-project.json", assets=f"{REF}/assets")
    for name in ("load-mono-wav.json", "load-stereo-wav.json"):
        d = describe(host, path=f"{REF}/projects/tests/{name}", assets=f"{REF}/assets")
        assert [x["kind"] for x in d["devices"]] == ["sampler"] and d["n_notes"] == 1 and d["warnings"] == 0, name
        assert d["devices"][0]["name"].endswith(".wav"), d["devices"][0]


def test_json5_syntax_and_errors(host):
    d = describe(host, text="{clock:{bpm:90,'time-signature':{top:3,bottom:4}},devices:[],/*c*/tracks:[],}")
    assert d["bpm"] == 90 and d["time_signature"] == [3, 4]
    with pytest.raises(RuntimeError, match="JSON5 parse error at line 1"):
        describe(host, text="{clock: ")
    with pytest.raises(RuntimeError, match="note-value"):
        describe(host, text='{"patterns":[{"id":"p","note-value":"seventh","notes":[[60]]}]}')
    # a one-ID patch cable is ignored with a warning (songs.rs:136-139); unknown effects pass through
    with pytest.raises(RuntimeError, match="unknown device class"):
        describe(host, text='{"devices":[{"gadget":["g",{"toy":{"my-value":0.5}}]}]}')
    d = describe(host, text='{"devices":[{"effect":["e",{"toy":{"my-value":0.5}}]}],"patch-cables":[["e"]]}')
    assert d["warnings"] == 2 and d["devices"][0]["fx_kind"] == T.FX_MIXER
    d = describe(host, text='{"devices":[{"effect":["e",{"filter-band-pass-12db":{"cutoff":500,"bandwidth":30}}]}]}')
    assert d["warnings"] == 0 and d["devices"][0]["fx_kind"] == T.FX_BIQUAD_BP12
    # patterns: default note value is a quarter; a 5-note row takes two 4/4 measures
    d = describe(host, text='{"patterns":[{"id":"p","notes":[[60,0,62,64,65]]}],"tracks":[{"id":"t","midi-channel":3,"patterns":["p","p"]}]}')
    assert d["n_notes"] == 8 and d["end_beats"] == 16.0


def _patch(L, text):
    out = T.WelshParams()
    err = C.create_string_buffer(512)
    if L.gh_welsh_params_from_patch_json(text.encode(), C.byref(out), err, 512):
        raise RuntimeError(err.value.decode())
    return out


def test_welsh_patch_derivation(host):
    """derive_welsh_synth_params (settings/src/patches.rs:87-170) on a synthetic patch."""
    patch = {
        "name": "t", "oscillator-1": {"waveform": {"pulse-width": 0.25}, "tune": {"osc": {"octave": 1, "semi": -5, "cent": 10}}, "mix-pct": 0.6},
        "oscillator-2": {"waveform": "sawtooth", "tune": {"note": 60}, "mix-pct": 0.2}, "oscillator-2-track": False,
        "oscillator-2-sync": True, "noise": 0, "lfo": {"routing": "pitch", "waveform": "triangle", "frequency": 5.5, "depth": {"cents": 20}},
        "glide": 0, "unison": False, "polyphony": "multi", "filter-type-24db": {"cutoff-hz": 900, "cutoff-pct": 0.5},
        "filter-type-12db": {"cutoff-hz": 450, "cutoff-pct": 0.4}, "filter-resonance": 0.3, "filter-envelope-weight": 0.7,
        "filter-envelope": {"attack": 0.1, "decay": 2.5, "sustain": 0.4, "release": 9}, "amp-envelope": {"attack": 0.01, "decay": 1.5, "sustain": 0.8, "release": 7},
    }
    p = _patch(host, json.dumps(patch))
    assert p.oscillator_1.waveform == T.WAVE_PULSE_WIDTH and abs(p.oscillator_1.duty - 0.25) < 1e-7
    assert abs(p.oscillator_1.tune - 2 ** ((7 * 100 + 10) / 1200)) < 1e-12          # Osc{octave 1, semi -5, cent 10}
    assert p.oscillator_2.waveform == T.WAVE_SAWTOOTH and p.oscillator_2.tune == 1.0
    assert abs(p.oscillator_2.fixed_hz - 261.6255653) < 1e-6                         # untracked: note_to_frequency(60)
    assert p.oscillator_2_sync == 1 and abs(p.oscillator_mix - 0.75) < 1e-6          # 0.6 / (0.6 + 0.2)
    assert p.amp_envelope.release == p.amp_envelope.decay == 1.5                     # release := decay (quirk)
    assert p.filter_envelope.release == p.filter_envelope.decay == 2.5
    assert p.lfo_routing == T.LFO_PITCH and p.lfo_waveform == T.WAVE_TRIANGLE and p.lfo_frequency == 5.5
    assert p.lfo_depth == 0.0                                                        # Cents(+20) → Normal::new(1 - 2^(20/1200)) clamps to 0
    assert p.filter_cutoff_hz == 900 and abs(p.filter_passband_ripple - (0.3 ** 2 * 10 + 0.707)) < 1e-6
    assert abs(p.filter_cutoff_start - math.log(450 / 25) / math.log(800)) < 1e-6 and abs(p.filter_cutoff_end - 0.7) < 1e-7
    assert p.dca_gain == 1.0 and p.dca_pan == 0.0
    # mix rule: one oscillator → 1; both mixes zero → 1; none → 0
    patch["oscillator-2"]["waveform"] = "none"; patch["oscillator-2-track"] = True
    assert _patch(host, json.dumps(patch)).oscillator_mix == 1.0
    patch["oscillator-1"]["waveform"] = "none"
    assert _patch(host, json.dumps(patch)).oscillator_mix == 0.0
    # untracked oscillator 2 without a note tune is the reference's panic (patches.rs:98)
    patch["oscillator-2"] = {"waveform": "square", "tune": {"float": 1}, "mix-pct": 1}; patch["oscillator-2-track"] = False
    with pytest.raises(RuntimeError, match="oscillator-2-track"):
        _patch(host, json.dumps(patch))
    # the routings of the shipped patch library beyond LfoRoutingType's five
    patch["oscillator-2-track"] = True
    for name, want in (("none", T.LFO_NONE), ("amplitude", T.LFO_AMPLITUDE), ("pitch", T.LFO_PITCH), ("pulse-width", T.LFO_PULSE_WIDTH),
                       ("filter-cutoff", T.LFO_FILTER_CUTOFF), ("pitch-osc2", T.LFO_PITCH_OSC2), ("pw-osc1", T.LFO_PW_OSC1),
                       ("pw-osc2", T.LFO_PW_OSC2), ("resonance", T.LFO_RESONANCE), ("cutoff-amp", T.LFO_CUTOFF_AMP)):
        patch["lfo"]["routing"] = name
        assert _patch(host, json.dumps(patch)).lfo_routing == want, name
    patch["lfo"]["routing"] = "lfo-routing-and-depth---something-else"
    assert _patch(host, json.dumps(patch)).lfo_routing == T.LFO_NONE


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_all_reference_welsh_patches_derive(host):
    files = sorted(glob.glob(f"{REF}/assets/patches/welsh/*.json"))
    assert len(files) == 106
    n_ok = 0
    for f in files:
        raw = json.load(open(f))
        try:
            p = _patch(host, open(f).read())
        except RuntimeError as e:
            assert "oscillator-2-track" in str(e), f"{f}: {e}"   # the reference panics on these too
            continue
        n_ok += 1
        assert p.amp_envelope.release == p.amp_envelope.decay
        assert p.filter_cutoff_hz == pytest.approx(raw["filter-type-24db"].get("cutoff-hz", 0.0))
        assert 0.0 <= p.oscillator_mix <= 1.0 and 0.0 <= p.filter_cutoff_start <= 1.0
        want = {"none": T.LFO_NONE, "amplitude": T.LFO_AMPLITUDE, "pitch": T.LFO_PITCH, "pulse-width": T.LFO_PULSE_WIDTH,
                "filter-cutoff": T.LFO_FILTER_CUTOFF, "pitch-osc2": T.LFO_PITCH_OSC2, "pw-osc1": T.LFO_PW_OSC1,
                "pw-osc2": T.LFO_PW_OSC2, "resonance": T.LFO_RESONANCE, "cutoff-amp": T.LFO_CUTOFF_AMP}
        assert p.lfo_routing == want.get(raw["lfo"]["routing"], T.LFO_NONE), f
    assert n_ok >= 100


def test_bus_station_routing_table(host):
    """BusStation: restates the reference's unit test (src/mini/bus_station.rs:55-140) through the
    compiled host layer's C surface."""
    L = host
    L.gh_bus_station_new.restype = C.c_void_p
    for f in ("gh_bus_station_free", "gh_bus_station_add_send_route", "gh_bus_station_remove_send_route", "gh_bus_station_remove_track_sends"):
        getattr(L, f).restype = None
    L.gh_bus_station_free.argtypes = [C.c_void_p]
    L.gh_bus_station_add_send_route.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double]
    L.gh_bus_station_remove_send_route.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.gh_bus_station_remove_track_sends.argtypes = [C.c_void_p, C.c_uint32]
    L.gh_bus_station_tracks.argtypes = [C.c_void_p]; L.gh_bus_station_tracks.restype = C.c_uint32
    L.gh_bus_station_sends_for.argtypes = [C.c_void_p, C.c_uint32]
    L.gh_bus_station_send.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    bs = L.gh_bus_station_new()
    assert L.gh_bus_station_tracks(bs) == 0
    L.gh_bus_station_add_send_route(bs, 7, 13, 0.8)
    assert L.gh_bus_station_tracks(bs) == 1
    L.gh_bus_station_add_send_route(bs, 7, 13, 0.7)
    assert L.gh_bus_station_tracks(bs) == 1, "a second route of the same track adds no track"
    aux, amount = C.c_uint32(), C.c_double()
    assert L.gh_bus_station_send(bs, 7, 1, C.byref(aux), C.byref(amount)) == 0 and aux.value == 13 and amount.value == 0.7
    L.gh_bus_station_remove_send_route(bs, 7, 13)
    assert L.gh_bus_station_tracks(bs) == 1 and L.gh_bus_station_sends_for(bs, 7) == 0, "the (empty) list stays"
    L.gh_bus_station_remove_send_route(bs, 7, 13)  # removing a nonexistent route is a no-op
    L.gh_bus_station_add_send_route(bs, 7, 13, 0.8)
    L.gh_bus_station_add_send_route(bs, 7, 14, 0.8)
    assert L.gh_bus_station_tracks(bs) == 1 and L.gh_bus_station_sends_for(bs, 7) == 2
    L.gh_bus_station_remove_track_sends(bs, 7)
    assert L.gh_bus_station_sends_for(bs, 7) == 0, "removing a track's sends leaves an empty list for it"
    assert L.gh_bus_station_sends_for(bs, 99) == -1
    L.gh_bus_station_free(bs)


def test_hostile_project_text_is_rejected_not_crashed(host):
    """The loaders read user files: nesting is bounded (the parser recurses), numbers far outside
    an integer field's range are clamped, truncated text gives an error string.  (The same inputs,
    plus a few thousand random mutations, were run under ASan + UBSan with float-cast-overflow.)"""
    with pytest.raises(RuntimeError, match="nesting"):
        describe(host, text="[" * 100000)
    with pytest.raises(RuntimeError, match="nesting"):
        describe(host, text='{"a":' * 5000 + "1" + "}" * 5000)
    base = open(SYNTHETIC).read()
    for cut in (1, len(base) // 3, len(base) - 2):
        with pytest.raises(RuntimeError):
            describe(host, text=base[:cut])
    d = describe(host, text='{"clock": {"bpm": 1e308, "time-signature": [1e99, -1e99]}, "devices": [], '
                            '"patterns": [{"id": "p", "notes": [[1e300, -1e300, NaN, 60]]}], '
                            '"tracks": [{"midi-channel": 1e100, "patterns": ["p"]}]}')
    assert d["time_signature"] == [64, 1]
    wp = T.WelshParams()
    err = C.create_string_buffer(256)
    rc = host.gh_welsh_params_from_patch_json(
        b'{"name": "x", "oscillator-1": {"waveform": "sine", "tune": {"osc": {"octave": 1e300, "semi": -1e300, "cent": 0}}}}',
        C.byref(wp), err, 256)
    assert rc in (0, 1)  # accepted with clamped tuning or rejected with a message: either way no crash


def _wav(bits=16, channels=2, fmt=1, frames=10, rate=44100, data_len=None, truncate=None):
    import struct
    bps = max(bits // 8, 0)
    payload = bytes((i * 37) & 0xFF for i in range(frames * channels * max(bps, 1)))
    hdr = struct.pack("<HHIIHH", fmt, channels, rate, rate * channels * bps, channels * bps, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(hdr)) + hdr + b"data" + struct.pack("<I", len(payload) if data_len is None else data_len) + payload
    blob = b"RIFF" + struct.pack("<I", len(body)) + body
    return blob if truncate is None else blob[:truncate]


def test_wav_reader_formats_and_malformed_files(host, tmp_path):
    """Sample files are user data too: 8/16/24/32-bit PCM and float decode, anything else is refused
    with a message (a 4-bit header used to divide by zero), lying chunk lengths are clamped."""
    import numpy as np
    host.gh_read_wav_mono.restype = C.c_int64
    host.gh_read_wav_mono.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]

    def read(blob):
        path = tmp_path / "x.wav"
        path.write_bytes(blob)
        out = np.zeros(64, dtype=np.float32)
        sr = C.c_uint32(0)
        err = C.create_string_buffer(256)
        n = host.gh_read_wav_mono(str(path).encode(), out.ctypes.data_as(C.POINTER(C.c_float)), 64, C.byref(sr), err, 256)
        return n, out, sr.value, err.value.decode()

    for bits in (8, 16, 24, 32):
        n, out, sr, _ = read(_wav(bits=bits))
        assert n == 10 and sr == 44100 and np.isfinite(out).all() and np.abs(out).max() <= 1.0
    n, out, _, _ = read(_wav(bits=16, channels=1, frames=4))
    want = np.frombuffer(bytes((i * 37) & 0xFF for i in range(8)), dtype="<i2") / 32768.0
    assert n == 4 and np.allclose(out[:4], want)
    for bits in (0, 4, 12, 64):
        n, _, _, err = read(_wav(bits=bits))
        assert n == -1 and err
    assert read(_wav(channels=0))[0] == -1
    assert read(_wav(channels=1000))[0] == -1
    assert read(_wav(data_len=0xFFFFFFF0))[0] == 10          # the data chunk claims 4 GiB: clamped to the file
    for cut in (0, 11, 20, 30, 44, 50):
        n = read(_wav(truncate=cut))[0]
        assert n == -1 or 0 <= n <= 10
    assert read(b"RIFF\x00\x00\x00\x00WAVEjunk")[0] == -1


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_every_reference_sample_file_decodes_like_the_standard_library(host):
    """The 81 WAV files of the reference's assets (16- and 24-bit, mono and stereo, 44.1 / 48 / 96 kHz: the drum kits' and the samplers'
    sources) through read_wav_mono against Python's `wave` module: every frame, the mean of the channels scaled by 2^(bits-1), bit for bit."""
    import wave
    import numpy as np
    host.gh_read_wav_mono.restype = C.c_int64
    host.gh_read_wav_mono.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
    files = sorted(glob.glob(f"{REF}/assets/samples/**/*.wav", recursive=True))
    assert len(files) == 81
    seen = set()
    for f in files:
        with wave.open(f, "rb") as w:
            ch, width, rate, frames = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(frames)
        if width == 2:
            ints = np.frombuffer(raw, dtype="<i2").astype(np.int64)
        else:
            assert width == 3, (f, width)
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int64)
            ints = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            ints = np.where(ints & 0x800000, ints - (1 << 24), ints)
        want = ((ints.reshape(-1, ch) / float(1 << (8 * width - 1))).sum(axis=1) / ch).astype(np.float32)
        out = np.zeros(len(want) + 8, dtype=np.float32)
        sr, err = C.c_uint32(0), C.create_string_buffer(256)
        n = host.gh_read_wav_mono(f.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), len(out), C.byref(sr), err, 256)
        assert n == len(want) and sr.value == rate, (f, n, len(want), err.value)
        assert np.array_equal(out[:n].view(np.uint32), want.view(np.uint32)), f
        seen.add((ch, width, rate))
    assert len(seen) >= 4


# ---- round 6: every instrument kind of the current schema (settings/src/instruments.rs:26-39)
def _wav_with_chunks(frames=16, extra=b""):
    import struct
    payload = bytes((i * 37) & 0xFF for i in range(frames * 2))
    hdr = struct.pack("<HHIIHH", 1, 1, 44100, 88200, 2, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(hdr)) + hdr + b"data" + struct.pack("<I", len(payload)) + payload + extra
    return b"RIFF" + struct.pack("<I", len(body)) + body


def smpl_chunk(unity_note):
    import struct
    data = struct.pack("<9I", 0, 0, 22675, unity_note, 0, 0, 0, 0, 0)
    return b"smpl" + struct.pack("<I", len(data)) + data


def acid_chunk(root_note, flags=2):
    import struct
    data = struct.pack("<IHHfIHHf", flags, root_note, 0x8000, 0.0, 4, 4, 4, 109.7)
    return b"acid" + struct.pack("<I", len(data)) + data


def test_sample_root_note_from_the_wav_file(host, tmp_path):
    """SamplerParams{root: 0} takes the file's own root: the `smpl` chunk's MIDI unity note, else the `acid` chunk's root
    note when its flags say it is set; neither, a lying length, a note above 127 or no file: -1 (the sampler then plays at 440 Hz)."""
    host.gh_read_wav_root_note.argtypes = [C.c_char_p]

    def root(blob):
        path = tmp_path / "r.wav"
        path.write_bytes(blob)
        return host.gh_read_wav_root_note(str(path).encode())

    assert root(_wav_with_chunks()) == -1
    assert root(_wav_with_chunks(extra=smpl_chunk(62))) == 62
    assert root(_wav_with_chunks(extra=acid_chunk(57))) == 57
    assert root(_wav_with_chunks(extra=acid_chunk(57, flags=1))) == -1          # one-shot flag only: the root note is not set
    assert root(_wav_with_chunks(extra=acid_chunk(57) + smpl_chunk(60))) == 60  # the smpl chunk wins
    assert root(_wav_with_chunks(extra=smpl_chunk(400))) == -1
    assert root(_wav_with_chunks(extra=b"smpl\xff\xff\xff\x7f\x00\x00")) == -1   # a chunk that claims 2 GiB
    assert root(_wav_with_chunks(extra=b"LIST\x03\x00\x00\x00abc\x00" + smpl_chunk(48))) == 48   # odd-sized chunk before it: padded
    assert host.gh_read_wav_root_note(str(tmp_path / "missing.wav").encode()) == -1
    if os.path.exists(REF):   # the reference's own pair of files (test-data/samples): one carries an acid chunk with root note 57
        assert host.gh_read_wav_root_note(f"{REF}/test-data/samples/riff-acidized.wav".encode()) == 57
        assert host.gh_read_wav_root_note(f"{REF}/test-data/samples/riff-not-acidized.wav".encode()) == -1


RAW_AND_TOY = """{
  title: "welsh-raw and toy-instrument", clock: {bpm: 120, "time-signature": [4, 4]},
  devices: [
    {instrument: ["raw-1", {"welsh-raw": [{"midi-in": 3}, {
        voice: {"oscillator-1": {waveform: {"pulse-width": 0.3}, "frequency-tune": 1.0},
                "oscillator-2": {waveform: "sawtooth", "frequency-tune": {osc: {octave: -1, semi: 0, cent: 4}}},
                "oscillator-2-sync": false, "oscillator-mix": 0.6,
                "amp-envelope": {attack: 0.01, decay: 0.2, sustain: 0.7, release: 0.3},
                lfo: {waveform: "square", frequency: 5.13}, "lfo-routing": "pitch", "lfo-depth": 0.05,
                filter: {cutoff: 900, "passband-ripple": 1.2}, "filter-cutoff-start": 0.4, "filter-cutoff-end": 0.5,
                "filter-envelope": {attack: 0.0, decay: 0.5, sustain: 0.3, release: 0.5}},
        dca: {gain: 0.8, pan: -0.25}}]}]},
    {instrument: ["toy-1", {"toy-instrument": [{"midi-in": 4}, {"fake-value": 0.25, dca: {gain: 0.5, pan: 0.5}}]}]},
    {instrument: ["s-1", {sampler: [{"midi-in": 5}, {filename: "pluck.wav", root: 0}]}]},
  ],
  "patch-cables": [["raw-1", "main-mixer"], ["toy-1", "main-mixer"], ["s-1", "main-mixer"]],
}"""


def test_welsh_raw_toy_instrument_and_sampler_parse(host):
    """InstrumentSettings::{WelshRaw, ToyInstrument, Sampler} (instruments.rs:27-37): parsed, none skipped, no warning."""
    d = describe(host, text=RAW_AND_TOY)
    kinds = {x["id"]: x for x in d["devices"]}
    assert set(kinds) == {"raw-1", "toy-1", "s-1"} and d["warnings"] == 0
    raw = kinds["raw-1"]
    assert raw["kind"] == "welsh-raw" and raw["midi_in"] == 3 and raw["welsh_osc1"] == T.WAVE_PULSE_WIDTH
    assert abs(raw["welsh_mix"] - 0.6) < 1e-6 and raw["welsh_cutoff"] == 900 and raw["welsh_routing"] == T.LFO_PITCH and abs(raw["welsh_release"] - 0.3) < 1e-12
    assert kinds["toy-1"]["kind"] == "toy-instrument" and kinds["toy-1"]["midi_in"] == 4
    assert kinds["s-1"]["kind"] == "sampler" and kinds["s-1"]["name"] == "pluck.wav"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_the_reference_sampler_demo_parses_with_its_roots(host):
    """projects/demos/instruments/sampler.json: two samplers over one stereo file, roots 587.33 Hz and 86 Hz."""
    d = describe(host, path=f"{REF}/projects/demos/instruments/sampler.json", assets=f"{REF}/assets")
    assert [x["kind"] for x in d["devices"]] == ["sampler", "sampler"] and [x["midi_in"] for x in d["devices"]] == [0, 1]
    assert all(x["name"] == "stereo-pluck.wav" for x in d["devices"]) and d["n_notes"] >= 8 and d["warnings"] == 0
