"""The synthetic many-voice projects of SURVEY.md §8d, as functions of the PROJECT VOICE INDEX.

A project is a set of voices 0 .. V-1; what voice i is (instrument kind, patch, key, when it
starts) depends on i alone, so any subset of voices — a rank's contiguous shard
(`groove_amd.parallel.voice_range`), or a sample picked for a parity check — renders exactly the
frames those voices contribute to the whole project's bus.  `plan()` turns a workload name and an
index set into plain data (parameter arrays, block-granular note events, effect-chain parameters);
`Project` instantiates a plan on the GPU through the entity surface (`groove_amd.entities`), and
the test-side `oracle/projects.py` instantiates the same plan on the CPU oracle.

Workloads (BASELINE.json `configs`):
    welsh-1m       1,000,000 Welsh voices, config-#2 voice rule            172 blocks of 256 frames
    welsh-1m-library  the same project over the 106-slot library-proportioned patch table (patches.py)
    welsh-256      config #2                                               172 blocks
    chain-4096     config #3: + BiQuad LP12 → Chorus → Delay → Reverb      172 blocks
    sampler-16384  config #4: one-shots over the shared bank, staggered    344 blocks
    mixed-131072   config #5: 50 % Welsh / 25 % FM / 25 % sampler          172 blocks
Voice rules:
    Welsh voice w: patch w mod 32, key 36 + (7 w mod 49), note-on at block 0, note-off at block 86;
    FM voice f: patch f mod 16, same keys and timeline;
    sampler voice s: buffer s mod 60, key 69 + (s mod 25) - 12, note-on at block h(s) mod 172,
        h(s) = s * 2654435761 mod 2^32 (one-shot: never released);
    mixed project voice i: i mod 4 in {0, 1} -> Welsh voice 2 (i div 4) + (i mod 4); 2 -> FM voice
        i div 4; 3 -> sampler voice i div 4 (every contiguous range keeps the 50 / 25 / 25 mix).
The timeline loops with the project's length.
"""
import os

import numpy as np

from . import abi_types as T
from . import entities as E
from . import patches as P

FRAMES = T.BLOCK_FRAMES
PACED_SLACK = 0  # extra blocks in a paced rotation (Project).  Measured, chain-4096, two runs each in one job (ms per block): 0 extra blocks
                 # 0.0507 / 0.0505, 1: 0.0510 / 0.0513, 2: 0.0513 / 0.0513, 3: 0.0514 / 0.0521, 5: 0.0526 / 0.0526 (unpaced: 0.0535 / 0.0547) —
                 # the host keeping further ahead buys nothing, and every extra block is 8 MB more for the caches to hold
TAKE_TURNS_MAX_VOICES = int(__import__("os").environ.get("GROOVE_TAKE_TURNS_MAX_VOICES", "16384"))  # multi-bank projects up to this size: banks in turn on the ctx stream

WORKLOADS = {
    "welsh-1m": dict(voices=1_000_000, kind="welsh", bytes_per_vf=18.0, dominant_bytes=10.0, blocks=172),
    # round 6: the same project over 106 synthetic patches whose class proportions follow the reference's patch library (patches.py LIBRARY_*)
    "welsh-1m-library": dict(voices=1_000_000, kind="welsh", bytes_per_vf=18.0, dominant_bytes=10.0, blocks=172, table="library-106"),
    "welsh-256": dict(voices=256, kind="welsh", bytes_per_vf=18.0, dominant_bytes=10.0, blocks=172),
    "chain-4096": dict(voices=4096, kind="chain", bytes_per_vf=218.0, dominant_bytes=10.0, blocks=172),
    "sampler-16384": dict(voices=16384, kind="sampler", bytes_per_vf=20.25, dominant_bytes=12.25, blocks=344),
    "mixed-131072": dict(voices=131072, kind="mixed", bytes_per_vf=18.2, dominant_bytes=10.0, blocks=172),
}
NOTE_OFF_BLOCK = P.NOTE_OFF_FRAME // FRAMES  # 86


def _take(table, ctype, idx):
    """table[idx] as a ctypes array (vectorised)."""
    import ctypes as C
    size = C.sizeof(ctype)
    raw = np.frombuffer(bytes(bytearray(table)), dtype=np.uint8).reshape(len(table), size)
    out = np.ascontiguousarray(raw[np.asarray(idx, dtype=np.int64)])
    return (ctype * len(idx)).from_buffer_copy(out.tobytes())


def _keys(v):
    return (36 + (7 * np.asarray(v, dtype=np.int64)) % 49).astype(np.uint8)


def _lanes(n):
    return np.arange(n, dtype=np.uint32)


def split_kinds(workload, sel):
    """Project voice indices `sel` -> {kind: within-kind voice numbers} (ascending)."""
    sel = np.asarray(sel, dtype=np.int64)
    kind = WORKLOADS[workload]["kind"]
    if kind in ("welsh", "chain"):
        return {"welsh": sel}
    if kind == "sampler":
        return {"sampler": sel}
    r, q = sel % 4, sel // 4
    return {"welsh": 2 * q[r < 2] + r[r < 2], "fm": q[r == 2], "sampler": q[r == 3]}


def plan(workload, sel, grouped=True, bank_scale=1.0):
    """The banks a shard of the project consists of, as plain data.  Returns a list of dicts:
    kind, n, params (ctypes array in LANE order), voice (within-kind voice number of each lane),
    events {block -> ctypes NoteEvent array}, fx [(kind, params)], and for samplers pcm / descs."""
    wl = WORKLOADS[workload]
    kinds = split_kinds(workload, sel)
    out = []
    w = kinds.get("welsh")
    if w is not None and len(w):
        entries, patch_of = P.PATCH_TABLES[wl.get("table", "benchmark-32")]
        if grouped:  # synth-major lane order: all voices of patch 0, then patch 1, ... (stable)
            w = w[np.argsort(w % entries, kind="stable")]
        table = (T.WelshParams * entries)(*[patch_of(j) for j in range(entries)])
        n = len(w)
        bank = dict(kind="welsh", n=n, params=_take(table, T.WelshParams, w % entries), voice=w, fx=[],
                    events={0: T.note_events_np(_lanes(n), _keys(w), True),
                            NOTE_OFF_BLOCK: T.note_events_np(_lanes(n), _keys(w), False)})
        if wl["kind"] == "chain":
            bank["fx"] = P.chain_fx_params(n, w)
        out.append(bank)
    f = kinds.get("fm")
    if f is not None and len(f):
        table = (T.FmParams * 16)(*[P.fm_patch(j) for j in range(16)])
        n = len(f)
        out.append(dict(kind="fm", n=n, params=_take(table, T.FmParams, f % 16), voice=f, fx=[],
                        events={0: T.note_events_np(_lanes(n), _keys(f), True),
                                NOTE_OFF_BLOCK: T.note_events_np(_lanes(n), _keys(f), False)}))
    s = kinds.get("sampler")
    if s is not None and len(s):
        n = len(s)
        pcm, descs, _ = P.drum_bank(scale=bank_scale)
        raw = np.zeros(n, dtype=np.dtype([("sample_index", "<u4"), ("one_shot", "<u4"), ("gain", "<f4")]))
        raw["sample_index"] = s % P.BANK_BUFFERS
        raw["one_shot"] = 1
        raw["gain"] = 1.0
        params = (T.SamplerParams * n).from_buffer_copy(raw.tobytes())
        keys = (69 + (s % 25) - 12).astype(np.uint8)
        h = (s.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
        start = (h % np.uint64(172)).astype(np.int64)
        events = {}
        for b in np.unique(start):
            lanes = np.nonzero(start == b)[0].astype(np.uint32)
            events[int(b)] = T.note_events_np(lanes, keys[lanes], True)
        out.append(dict(kind="sampler", n=n, params=params, voice=s, fx=[], events=events, pcm=pcm, descs=descs))
    return out


class Project:
    """One shard of a synthetic project on one GPU: banks, blocks, effect chains and the block loop
    (instruments render, their chains run, the mix bus sums: Orchestrator::tick / gather_audio,
    /root/reference/orchestration/src/orchestrator.rs:856-877, 367-470)."""

    def __init__(self, ctx, workload, sel, fused=True, grouped=True, render_ahead=True, bank_scale=1.0, head_ahead=True, paced=None, one_launch=True, allpass_stream=None):
        self.ctx, self.fused, self.workload = ctx, fused, workload
        # PACED walk (instruments with an effect chain; default for them): the renders go out TWO blocks ahead into a rotation of
        # four blocks, and the host itself waits for the two events a step depends on — the release of the block the new render
        # fills (two steps old) and the render of the block the chain is about to take (two steps old as well) — so that both are
        # complete when the calls are made and neither stream carries a cross-queue wait packet (groove_block_wait_released /
        # _ready; docs/STREAMS.md item 13: 7 - 9 us of the waiting stream's timeline each).  The block's bus reduction rides in
        # the next block's chain launch (groove_mix_deferred).  paced=False: round 3's walk (one block ahead, device-side waits).
        self.paced = paced
        self.head_ahead = head_ahead      # render-ahead walk: the chain's leading IIR stages ride behind the render (groove_fx_chain_process_async)
        self.head_done = {}               # block handle -> stages of its chain already processed
        self.period = WORKLOADS[workload]["blocks"]
        self.render_ahead = render_ahead  # instruments with an effect chain: render block b+1 beside the effects of block b
        self.ahead = {}                   # instrument -> [current block, next block, spare] (rotating), created on first use
        self.primed = False               # the current block's render has been submitted
        self.block_index = 0
        self.n = int(len(sel))
        self.banks = []  # (instrument, block, [effects], events)
        for spec in plan(workload, sel, grouped, bank_scale):
            if spec["kind"] == "welsh":
                inst = E.WelshSynth(ctx, spec["params"])
            elif spec["kind"] == "fm":
                inst = E.FmSynth(ctx, spec["params"])
            else:
                inst = E.Sampler(ctx, spec["pcm"], spec["descs"], spec["params"])
            fx = [E.Effect(ctx, k, p) for k, p in spec["fx"]]
            block = ctx.block(spec["n"], FRAMES) if (fx or not fused) else None
            self.banks.append((inst, block, fx, spec["events"]))
        self.dominant = self.banks[0][0] if self.banks else None
        self.has_chain = any(fx for _, _, fx, _ in self.banks)
        # the entity-boundary (materialised) form takes the render-ahead walk too: with the renders on the side streams and
        # three blocks per instrument in rotation, the per-kind kernels of consecutive blocks follow each other on their
        # streams the way the fused path's do — one block alone costs its thinly occupied tail (0.70 against 0.54 ms at
        # 1,000,000 voices)
        self.ahead_walk = self.render_ahead and (self.has_chain or not fused)
        if self.paced is None:
            self.paced = self.ahead_walk and self.has_chain
        self.paced = bool(self.paced and self.ahead_walk)
        self.lookahead = 2 if self.paced else 1
        # The paced walk of a chain that ends in a reverb: the reverb's all-passes (the block's last kernel) go to a side stream of
        # the library, beside the next block's fused run (groove_set_fx_allpass_stream; round 5: config #3 0.0489 -> 0.044 ms per
        # block).  allpass_stream=False: on the ctx stream behind the run (A/B).  A ctx-wide knob: restored by destroy().
        if allpass_stream is None:   # (GROOVE_PROJECT_ALLPASS_STREAM=0: off, for in-job A/B runs of bench.py)
            allpass_stream = os.environ.get("GROOVE_PROJECT_ALLPASS_STREAM", "1") != "0" and self.has_chain
        self.allpass_stream = bool(allpass_stream and self.paced)
        # (the release of a block is then recorded behind its all-passes on that stream, and the host waits for the release of the block
        # the next render fills: with no slack that is the previous step's, and the host would hold back this step's run until those
        # all-passes are done — the very overlap the stream is for.  Two extra blocks in the rotation, the host waits three steps back:
        # in one job 0.0489 (ctx stream) / 0.0473 - 0.0508 (no slack) / 0.0399 - 0.0436 (one) / 0.0401 - 0.0407 (two), profiles/r05_ap_stream_ab.log.)
        self.paced_slack = int(os.environ.get("GROOVE_PROJECT_PACED_SLACK", PACED_SLACK + (2 if self.allpass_stream else 0)))
        self._allpass_stream_before = ctx.fx_allpass_stream
        if self.allpass_stream != self._allpass_stream_before:
            ctx.fx_allpass_stream = self.allpass_stream
        # A small fused project (a lone bank; or a few small banks — config #5's 16,384-voice share of a GPU) renders its banks one
        # after the other on the ctx stream, every render carrying the bus reduction of the one before it
        # (groove_bank_render_mix_deferred): one launch per bank and block, no cross-queue waits.  Bigger banks render side by side.
        total = sum(inst.n for inst, _, _, _ in self.banks)
        self.take_turns = fused and not self.has_chain and (len(self.banks) == 1 or total <= TAKE_TURNS_MAX_VOICES)
        # ... and since round 5 in ONE launch per block where the library can (groove_banks_render_mix_deferred: at most one
        # time-parallel bank of each kind): the banks' durations no longer add.  one_launch=False: the banks in turn (A/B).
        self.one_launch = self.take_turns and len(self.banks) > 1 and one_launch
        # Every other fused project — banks side by side on the library's streams, or one bank big enough for the per-kind block
        # pipeline — is PACED by default (groove_bank_render_mix_paced: the host waits for the events, every bank's bus reduction is
        # launched by its next call); paced=False: device-side waits.  Measured in one job (profiles/r04_paced_fused_ab.log):
        # 1,000,000 voices 0.4885 / 0.4916 / 0.4885 against 0.4974 / 0.4939 / 0.4925 ms per block, config #5 on one GPU 0.0914 / 0.0914
        # against 0.0946 / 0.0942; its 16,384-voice share of eight GPUs gains nothing side by side (0.055 - 0.059 against 0.056) and
        # keeps taking turns.
        pipelined = len(self.banks) == 1 and "blocks pipelined" in self.banks[0][0].kernel_form(FRAMES, True)  # (a lone bank of >= ~550,000 Welsh voices)
        if pipelined:
            self.take_turns = False  # (groove_bank_render_mix_deferred would hand such a bank to groove_bank_render_mix anyway)
        self.paced_fused = fused and not self.has_chain and (paced is not False) and not self.take_turns

    def reset(self):
        """Back to block 0 of the timeline with every voice and effect in its initial state."""
        for inst, _, fx, _ in self.banks:
            inst.reset()
            for e in fx:
                e.reset()
        self.primed = False  # (the three blocks per instrument stay)
        self.head_done = {}
        self.block_index = 0

    def _events(self, block_index):
        b = block_index % self.period
        for inst, _, _, events in self.banks:
            ev = events.get(b)
            if ev is not None:
                inst.handle_midi_events(ev)

    def _step_render_ahead(self, bus, frame0, ev_pair):
        """The same block walk, software-pipelined: the instruments' render of block b+1 goes to the
        library's side streams (groove_bank_render_async) before the effect chains of block b are
        submitted, three blocks per instrument in rotation.  Every step still submits one render, one
        pass of every effect and one mix per instrument; the first call also renders block b itself."""
        ctx = self.ctx
        if ev_pair is not None and ev_pair[0] is not None:
            ctx.record(ev_pair[0])
        L = self.lookahead
        if not self.primed:
            for inst, block, fx, _ in self.banks:
                if inst not in self.ahead:
                    # L + 1 blocks are in use at any time; the paced walk adds slack: the block a render fills was released
                    # PACED_SLACK + 1 steps ago, so the host — which waits for that release — may run that far ahead of the GPU
                    self.ahead[inst] = [block] + [ctx.block(inst.n, FRAMES) for _ in range(L + 1 + (self.paced_slack if self.paced else 0))]
            for d in range(L):  # blocks b .. b + L - 1
                self._events(self.block_index + d)
                for inst, _, fx, _ in self.banks:
                    self._render_ahead(inst, fx, self.ahead[inst][d])
            self.primed = True
        self._events(self.block_index + L)
        self.block_index += 1
        for inst, _, fx, _ in self.banks:
            # the block this render fills was released L steps ago: no cross-queue wait (groove_block_release) — and in the paced walk
            # the host has SEEN that release complete, so the side stream carries no wait packet at all
            nxt = self.ahead[inst][L]
            if self.paced:
                nxt.wait_released()
            self._render_ahead(inst, fx, nxt)
        first = True
        for inst, _, fx, _ in self.banks:
            cur = self.ahead[inst][0]
            if self.paced:
                cur.wait_ready()
            ctx.transform_chain(fx[self.head_done.pop(id(cur), 0):], cur, FRAMES)
            if self.paced:
                ctx.mix_deferred(cur, FRAMES, E._Slice(bus, frame0), accumulate=not first)
            else:
                ctx.mix([cur], FRAMES, E._Slice(bus, frame0), accumulate=not first)
            cur.release()
            self.ahead[inst] = self.ahead[inst][1:] + [cur]
            first = False
        if ev_pair is not None and ev_pair[1] is not None:
            ctx.record(ev_pair[1])

    def _render_ahead(self, inst, fx, block):
        """The instrument's next block on the side streams; with head_ahead, the chain's leading IIR stages ride behind it
        (head_ahead == "unfused": as a launch of their own; otherwise fused into the render kernel where the library can)."""
        if fx and self.head_ahead and self.head_ahead != "unfused":
            self.head_done[id(block)] = inst.generate_batch_values_chain_async(block, fx, FRAMES)
            return
        inst.generate_batch_values_async(block, FRAMES)
        if fx and self.head_ahead:
            self.head_done[id(block)] = self.ctx.transform_chain_async(fx, block, FRAMES)

    def step(self, bus, frame0, ev_pair=None):
        """One block: every instrument renders, its effect chain runs, the mix bus sums."""
        ctx = self.ctx
        if self.ahead_walk:
            return self._step_render_ahead(bus, frame0, ev_pair)
        self._events(self.block_index)
        self.block_index += 1
        if self.one_launch:
            if ev_pair is not None and ev_pair[0] is not None:
                ctx.record(ev_pair[0])
            ctx.render_mix_banks_deferred([inst for inst, _, _, _ in self.banks], bus, FRAMES, accumulate=False, at_frame=frame0)
            if ev_pair is not None and ev_pair[1] is not None:
                ctx.record(ev_pair[1])
            return
        first = True
        for inst, block, fx, _ in self.banks:
            if ev_pair is not None and ev_pair[0] is not None and inst is self.dominant:
                ctx.record(ev_pair[0])
            if self.fused and not fx:
                if self.take_turns:  # small banks, one after the other on the ctx stream: a render carries the reduction of the one before it
                    inst.render_mix_deferred(bus, FRAMES, accumulate=not first, at_frame=frame0)
                elif self.paced_fused:
                    inst.render_mix_paced(bus, FRAMES, accumulate=not first, at_frame=frame0)
                else:
                    inst.render_mix(bus, FRAMES, accumulate=not first, at_frame=frame0)
                if ev_pair is not None and ev_pair[1] is not None and inst is self.banks[-1][0]:
                    ctx.record(ev_pair[1])  # fused steps are bracketed whole: after the last bank's bus sum
            else:
                inst.generate_batch_values(block, FRAMES)
                if ev_pair is not None and ev_pair[1] is not None and inst is self.dominant:
                    ctx.record(ev_pair[1])
                ctx.transform_chain(fx, block, FRAMES)
                ctx.mix([block], FRAMES, E._Slice(bus, frame0), accumulate=not first)
            first = False

    def destroy(self):
        for inst, block, fx, _ in self.banks:
            for e in fx:
                e.destroy()
            inst.destroy()
            for b in self.ahead.get(inst, [block]):  # the rotation holds the bank's own block too
                if b is not None:
                    b.destroy()
        self.ahead = {}
        self.banks = []
        if self.ctx.fx_allpass_stream != self._allpass_stream_before:
            self.ctx.fx_allpass_stream = self._allpass_stream_before
