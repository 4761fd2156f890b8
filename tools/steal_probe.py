"""Two note-ons in one block on a two-voice FM synth whose voices are both still in their release: host layer against the oracle graph fed by
the Python restatement of the host's allocation (tests/test_gpu_orchestrator.py::test_random_graphs_against_the_oracle_graph, seed 8).
    python3 tools/steal_probe.py"""
import math, os, sys
REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from groove_amd import abi_types as T, patches as P, host_binding as H
from oracle import oracle as O

bpm, sr, upb, block = 128.0, 44100, 65536, 256
notes = [(41, 19256, 30810), (80, 28291, 43041), (48, 57055, 77371), (47, 57372, 67560)]   # (key, on units, off units)
total = math.ceil(1.5 * 60 / bpm * sr)
for patch_j in range(4):
    patch = P.fm_patch(patch_j)
    o, g = H.Orchestrator(0, sr, bpm), O.Graph(sr)
    g.set_bpm(bpm)
    u = o.add_fm(patch, voices=2)
    gu = g.add_instrument(O.Bank.fm((T.FmParams * 2)(*[patch] * 2)))
    assert o.patch(u, o.MAIN_MIXER) == 0 and g.patch(gu, g.MAIN_MIXER) == 0
    o.connect_midi_downstream(u, 0)
    seq = o.add_sequencer()
    events = []
    for key, a, b in notes:
        o.sequencer_insert(seq, 0, key, a / upb, (b - a) / upb)
        events.append((a, len(events), key, True)); events.append((b, len(events), key, False))
    o.sequencer_set_end(seq, 1.5)
    got = o.run(block).astype(np.float64)
    print("host: last allocated voice", o.last_allocated_voice(u))
    o.close()
    events.sort(key=lambda e: (e[0], e[1]))
    keyv, busy, started = [-1, -1], [0, 0], [0, 0]
    rel = math.ceil(patch.carrier_envelope.release * sr) + 1
    want, pos = [], 0
    while pos < total:
        fr = min(block, total - pos)
        t0, t1 = int(pos * bpm / 60.0 / sr * upb), int((pos + fr) * bpm / 60.0 / sr * upb)
        for at, _, key, on in events:
            if t0 <= at < t1:
                if on:
                    v = next((i for i in range(2) if keyv[i] < 0 and busy[i] <= pos), None)
                    if v is None:
                        v = min(range(2), key=lambda i: started[i])
                    keyv[v], started[v], busy[v] = key, pos, 1 << 62
                    print("   model: block at", pos, "key", key, "-> voice", v)
                    g.note_events(gu, T.note_events([(v, key, True)]))
                else:
                    for i in range(2):
                        if keyv[i] == key:
                            g.note_events(gu, T.note_events([(i, key, False)])); keyv[i] = -1; busy[i] = pos + rel
        want.append(g.tick(fr)); pos += fr
    want = np.concatenate(want, axis=0)
    d = np.abs(got - want).max(axis=1)
    bad = np.nonzero(d > 1e-4)[0]
    print("fm patch", patch_j, "release", patch.carrier_envelope.release, "max diff", float(d.max()), "first bad frame", int(bad[0]) if bad.size else None)
