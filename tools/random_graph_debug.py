"""tests/test_gpu_orchestrator.py::test_random_graphs_against_the_oracle_graph for one seed, with the first diverging frame and the event list.
    python3 tools/random_graph_debug.py <seed>"""
import math, os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, patches as P
from oracle import oracle as O

def run(oracle, seed_only):
    """The same kind of seeded random project as above, this time against the ORACLE: its patch graph (per-frame DFS from the main mixer, an
    effect sums all its sources before it transforms: orchestrator.rs:367-470) fed by a Python restatement of the host's control side —
    the sequencer's events block by block in time order (equal times in insertion order), first-idle voice allocation with stealing of the
    voice that started first (busy until note-off + the patch's release), a cutoff trip valued at block starts.  Welsh and FM synths and toy
    sources, through random chains with fan-in; bus RMS against the oracle's <= 1e-5 of the bus's scale."""
    import os
    from groove_amd import host_binding as H
    bpm, sr, upb, block = 128.0, 44100, 65536, 256
    fx_menu = [(T.FX_GAIN, dict(ceiling=0.6)), (T.FX_BIQUAD_LP12, dict(cutoff_hz=1200.0, q=0.9)), (T.FX_BIQUAD_LP24, dict(cutoff_hz=900.0, passband_ripple=0.8)),
               (T.FX_DELAY, dict(delay_seconds=0.004)), (T.FX_CHORUS, dict(voices=3, delay_seconds=0.006)), (T.FX_REVERB, dict(attenuation=0.7, reverb_seconds=0.4))]
    end_beats = 1.5
    total = math.ceil(end_beats * 60 / bpm * sr)

    class Alloc:   # VoiceBankInstrument::note_on / note_off (groove_amd/host/groove_host.cpp) restated
        def __init__(self, voices, release_seconds):
            self.key, self.busy, self.started, self.rel = [-1] * voices, [0] * voices, [0] * voices, math.ceil(release_seconds * sr) + 1

        def on(self, key, now):
            n = len(self.key)
            v = next((i for i in range(n) if self.key[i] < 0 and self.busy[i] <= now), None)
            if v is None:
                v = min(range(n), key=lambda i: self.started[i])
            self.key[v], self.started[v], self.busy[v] = key, now, 1 << 62
            return [(v, key, True)]

        def off(self, key, now):
            out = []
            for i in range(len(self.key)):
                if self.key[i] == key:
                    out.append((i, key, False))
                    self.key[i] = -1
                    self.busy[i] = now + self.rel
            return out

    for seed in [seed_only]:
        rng = np.random.default_rng(5000 + seed)
        o, g = H.Orchestrator(0, sr, bpm), oracle.Graph(sr)
        try:
            g.set_bpm(bpm)
            seq = o.add_sequencer()
            allocs, events, effects, filters = {}, [], [], []
            desc = []
            n_inst = int(rng.integers(2, 6))
            for ch in range(n_inst):
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    patch, voices = P.welsh_patch(int(rng.integers(0, P.N_PATCHES))), int(rng.integers(2, 7))
                    u = o.add_welsh(patch, voices=voices)
                    gu = g.add_instrument(oracle.Bank.welsh((T.WelshParams * voices)(*[patch] * voices)))
                    allocs[ch] = (gu, Alloc(voices, patch.amp_envelope.release))
                elif kind == 1:
                    patch, voices = P.fm_patch(int(rng.integers(0, 8))), int(rng.integers(2, 5))
                    u = o.add_fm(patch, voices=voices)
                    gu = g.add_instrument(oracle.Bank.fm((T.FmParams * voices)(*[patch] * voices)))
                    allocs[ch] = (gu, Alloc(voices, patch.carrier_envelope.release))
                else:
                    level = float(rng.uniform(0.01, 0.1))
                    u, gu = o.add_toy_source(level), g.add_source(level)
                desc.append((ch, kind, voices if kind != 2 else 0))
                route = rng.random()
                if route < 0.35 or (route < 0.5 and not effects):
                    assert o.patch(u, o.MAIN_MIXER) == 0 and g.patch(gu, g.MAIN_MIXER) == 0
                elif route < 0.5:
                    e, ge = effects[int(rng.integers(len(effects)))]
                    assert o.patch(u, e) == 0 and g.patch(gu, ge) == 0
                else:
                    chain, gchain = [u], [gu]
                    for _ in range(int(rng.integers(1, 4))):
                        k, kw = fx_menu[int(rng.integers(len(fx_menu)))]
                        chain.append(o.add_effect(k, T.fx_params(**kw))); gchain.append(g.add_effect(k, T.fx_params(**kw)))
                        effects.append((chain[-1], gchain[-1]))
                        if k in (T.FX_BIQUAD_LP12, T.FX_BIQUAD_LP24):
                            filters.append((chain[-1], gchain[-1]))
                    assert o.patch_chain_to_main_mixer(chain) == 0 and g.patch_chain_to_main_mixer(gchain) == 0
                if kind != 2:
                    o.connect_midi_downstream(u, ch)
                    for _ in range(int(rng.integers(3, 9))):
                        key, start, dur = int(rng.integers(40, 84)), float(rng.uniform(0.0, 1.2)), float(rng.uniform(0.1, 0.6))
                        key += key % 12 == 9                     # (no A: docs/DSP_SPEC.md section 2, exact ties)
                        o.sequencer_insert(seq, ch, key, start, dur)
                        events.append((int(start * upb + 0.5), len(events), ch, key, True))
                        events.append((int((start + dur) * upb + 0.5), len(events), ch, key, False))
            o.sequencer_set_end(seq, end_beats)
            if filters:
                e, ge = filters[int(rng.integers(len(filters)))]
                start, a, b, beats = float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.2, 0.9)), float(rng.uniform(0.3, 1.0))
                kind = int(rng.choice([H.STEP_SLOPE, H.STEP_EXPONENTIAL, H.STEP_LOGARITHMIC]))
                trip, gtrip = o.add_control_trip(e, "cutoff", start), g.add_control_trip(ge, T.CTL_FX_CUTOFF, start)
                o.control_trip_add_step(trip, kind, a, b, beats); g.trip_add_step(gtrip, kind, a, b, beats)
                print('trip: start beat', start, 'from', a, 'to', b, 'beats', beats, 'kind', kind, '-> ends at frame', (start + beats) * 60 / bpm * sr)
            got = o.run(block).astype(np.float64)
            # the oracle side: sequencer order = by time, equal times in insertion order (stable upper_bound insert)
            events.sort(key=lambda e: (e[0], e[1]))
            want, pos = [], 0
            while pos < total:
                fr = min(block, total - pos)
                t0, t1 = int(pos * bpm / 60.0 / sr * upb), int((pos + fr) * bpm / 60.0 / sr * upb)
                for at, _, ch, key, on in events:
                    if t0 <= at < t1:
                        gu, al = allocs[ch]
                        for ev in (al.on(key, pos) if on else al.off(key, pos)):
                            g.note_events(gu, T.note_events([ev]))
                want.append(g.tick(fr)); pos += fr
            want = np.concatenate(want, axis=0)
        finally:
            o.close()
        d = np.abs(got - want).max(axis=1)
        bad = np.nonzero(d > 1e-4)[0]
        print("total", total, "first bad frame", int(bad[0]) if bad.size else None, "block", int(bad[0]) // block if bad.size else None, "max diff", float(d.max()))
        print("instruments:", desc)
        for at, idx, ch, key, on in events:
            print("  event units", at, "frame", round(at / upb * 60 / bpm * sr, 1), "ch", ch, "key", key, "on" if on else "off")


run(O, int(sys.argv[1]))
