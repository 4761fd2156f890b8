#!/usr/bin/env python3
"""bench.py — offline-render throughput of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload welsh-1m]

A "step" is one pass of the hot path over one 256-frame block of the whole project: every
Welsh voice ticks 256 frames (welsh_render_kernel) and the mix bus sums the voice blocks
(Orchestrator::gather_audio).  `value` = stereo bus frames rendered per second, whole job,
with all inputs (patch parameters, voice state) resident in HBM before the timed region.

Workloads:
    welsh-1m      1,000,000 Welsh voices (config-#2 voice rule) — north-star target, default
    welsh-256     config #2   (256 voices; 4 wavefronts: a latency config)
    chain-4096    config #3   (4,096 voices + BiQuad→Chorus→Delay→Reverb per voice)
    sampler-16384 config #4   (16,384 one-shot sampler voices over a shared bank)
    mixed-131072  config #5   (50 % Welsh / 25 % FM / 25 % sampler)

Multi-GPU: the project's voices are cut into contiguous index ranges, one range per rank (one
process per GPU, no data-path collective), and the per-rank buses are summed with ONE RCCL reduce
over the whole timed region's frames (K*256 frames * 8 B), inside the timed region.
  default (weak scaling): every rank holds the workload's voice count, so the project grows with
      N (N x 1,000,000 voices); `value` = stereo frames rendered by all ranks per second (each
      rank renders the K*256 bus frames of its shard), `project_frames_per_s` = the merged
      project's frames per second (flat when scaling is ideal), `voice_frames_per_s` the rate in
      the unit that does not depend on how the voices are grouped.
  --strong: the workload's voice count is fixed and split N ways; `value` = project frames/s.
      (1,000,000 voices are small for eight MI355X: DESIGN.md §6 has the latency ceiling.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from groove_amd import entities as E, patches as P, abi_types as T  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAMES = T.BLOCK_FRAMES
SR = T.DEFAULT_SAMPLE_RATE

WORKLOADS = {
    "welsh-1m": dict(voices=1_000_000, kind="welsh", bytes_per_vf=18.0, dominant_bytes=10.0),
    "welsh-256": dict(voices=256, kind="welsh", bytes_per_vf=18.0, dominant_bytes=10.0),
    "chain-4096": dict(voices=4096, kind="chain", bytes_per_vf=218.0, dominant_bytes=10.0),
    "sampler-16384": dict(voices=16384, kind="sampler", bytes_per_vf=20.25, dominant_bytes=12.25),
    "mixed-131072": dict(voices=131072, kind="mixed", bytes_per_vf=18.2, dominant_bytes=10.0),
}


class Project:
    """The synthetic many-voice project shard owned by one rank."""

    def __init__(self, ctx, workload, first_voice, n_voices, fused, grouped=True, render_ahead=True):
        self.ctx, self.n, self.fused = ctx, n_voices, fused
        self.render_ahead = render_ahead  # instruments with an effect chain: render block b+1 beside the effects of block b
        self.ahead = {}                   # instrument -> [current block, next block] once primed
        kind = WORKLOADS[workload]["kind"]
        self.banks = []   # (instrument, block, [effects])
        self.timeline = []  # Welsh banks that follow the config-#2 note timeline
        self.on_ev = self.off_ev = None
        self.block_index = 0
        if kind in ("welsh", "chain"):
            if grouped:
                params, idx = P.welsh_voices_grouped(n_voices, first_voice)
                synth = E.WelshSynth(ctx, params)
                self.on_ev, self.off_ev = P.grouped_note_events(idx, True), P.grouped_note_events(idx, False)
            else:
                idx = np.arange(first_voice, first_voice + n_voices)
                synth = E.WelshSynth(ctx, P.welsh_voices(n_voices, first_voice))
                self.on_ev, self.off_ev = P.note_on_all(n_voices, first_voice), P.note_off_all(n_voices, first_voice)
            self.timeline.append(synth)
            fx = []
            if kind == "chain":  # per-voice chain parameters follow the voice into its lane
                fx = [E.Effect(ctx, k, p) for k, p in P.chain_fx_params(n_voices, idx)]
            self.banks.append((synth, ctx.block(n_voices, FRAMES), fx))
        elif kind == "sampler":
            pcm, descs, _ = P.drum_bank()
            s = E.Sampler(ctx, pcm, descs, P.sampler_voices(n_voices))
            s.handle_midi_events(T.note_events_np(np.arange(n_voices, dtype=np.uint32), P.sampler_keys(n_voices), True))
            self.banks.append((s, ctx.block(n_voices, FRAMES), []))
        elif kind == "mixed":
            nw, nf = n_voices // 2, n_voices // 4
            ns = n_voices - nw - nf
            wp, widx = P.welsh_voices_grouped(nw, first_voice)
            w = E.WelshSynth(ctx, wp)
            self.on_ev, self.off_ev = P.grouped_note_events(widx, True), P.grouped_note_events(widx, False)
            self.timeline.append(w)
            f = E.FmSynth(ctx, P.fm_voices(nf, first_voice))
            f.handle_midi_events(P.note_on_all(nf, first_voice))
            pcm, descs, _ = P.drum_bank()
            s = E.Sampler(ctx, pcm, descs, P.sampler_voices(ns))
            s.handle_midi_events(T.note_events_np(np.arange(ns, dtype=np.uint32), P.sampler_keys(ns), True))
            for inst, n in ((w, nw), (f, nf), (s, ns)):
                self.banks.append((inst, ctx.block(n, FRAMES), []))
        self.dominant = self.banks[0][0]

    def _timeline_events(self, block_index):
        # config-#2 timeline, looped: note-on at block 0, note-off at block 86 of every 172 blocks
        b = block_index % P.RENDER_BLOCKS
        for synth in self.timeline:
            if b == 0:
                synth.handle_midi_events(self.on_ev)
            elif b == P.NOTE_OFF_FRAME // FRAMES:
                synth.handle_midi_events(self.off_ev)

    def _step_render_ahead(self, bus, frame0, ev_pair):
        """The same block walk, software-pipelined: the instruments' render of block b+1 goes to the
        library's side streams (groove_bank_render_async) before the effect chains of block b are
        submitted, two blocks per instrument alternating.  Every step still submits one render, one
        pass of every effect and one mix per instrument; the first call also renders block b itself."""
        ctx = self.ctx
        if ev_pair is not None and ev_pair[0] is not None:
            ctx.record(ev_pair[0])
        if not self.ahead:  # three blocks per instrument in rotation: [current, next, spare]
            self._timeline_events(self.block_index)
            for inst, block, fx in self.banks:
                self.ahead[inst] = [block, ctx.block(inst.n, FRAMES), ctx.block(inst.n, FRAMES)]
                inst.generate_batch_values_async(block, FRAMES)
        self._timeline_events(self.block_index + 1)
        self.block_index += 1
        for inst, _, fx in self.banks:
            # the block this render fills was released a whole step ago: no cross-queue wait (groove_block_release)
            inst.generate_batch_values_async(self.ahead[inst][1], FRAMES)
        first = True
        for inst, _, fx in self.banks:
            cur = self.ahead[inst][0]
            for e in fx:
                e.transform_audio(cur, FRAMES)
            ctx.mix([cur], FRAMES, E._Slice(bus, frame0), accumulate=not first)
            cur.release()
            self.ahead[inst] = self.ahead[inst][1:] + [cur]
            first = False
        if ev_pair is not None and ev_pair[1] is not None:
            ctx.record(ev_pair[1])

    def step(self, bus, frame0, ev_pair=None):
        """One block: every instrument renders, its effect chain runs, the mix bus sums."""
        ctx = self.ctx
        if self.render_ahead and any(fx for _, _, fx in self.banks):
            return self._step_render_ahead(bus, frame0, ev_pair)
        self._timeline_events(self.block_index)
        self.block_index += 1
        first = True
        for inst, block, fx in self.banks:
            if self.fused and not fx:
                if ev_pair is not None and ev_pair[0] is not None and inst is self.dominant:
                    ctx.record(ev_pair[0])
                inst.render_mix(bus, FRAMES, accumulate=not first, at_frame=frame0)
                if ev_pair is not None and ev_pair[1] is not None and inst is self.banks[-1][0]:
                    ctx.record(ev_pair[1])  # fused steps are bracketed whole: after the last bank's bus sum
            else:
                if ev_pair is not None and ev_pair[0] is not None and inst is self.dominant:
                    ctx.record(ev_pair[0])
                inst.generate_batch_values(block, FRAMES)
                if ev_pair is not None and ev_pair[1] is not None and inst is self.dominant:
                    ctx.record(ev_pair[1])
                for e in fx:
                    e.transform_audio(block, FRAMES)
                ctx.mix([block], FRAMES, E._Slice(bus, frame0), accumulate=not first)
            first = False


def committed_traffic(workload, world):
    """HBM bytes per step from the committed PMC passes (profiles/*_summary.json, FETCH_SIZE doubled per
    MI355X_MICROARCH.md §HBM); only meaningful for the default single-GPU workload it was collected on."""
    if workload != "welsh-1m" or world != 1:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        t = json.load(open(files[-1])).get("hbm_traffic_bytes_per_step", {})
        return {"bytes_per_step": t.get("total_corrected"), "source": os.path.basename(files[-1])}
    except Exception:
        return None


def cpu_baseline(workload, seconds_target=15.0):
    """The f64 scalar oracle ("port"; the reference Rust path cannot be built here) timed on
    this host, rank 0 only, on a bounded sample of the same workload, single thread (mode A)."""
    from oracle import oracle as O
    L = O.lib()
    kind = WORKLOADS[workload]["kind"]
    V = WORKLOADS[workload]["voices"]
    if kind not in ("welsh", "chain", "mixed"):
        kind = "sampler"
    sample_voices = min(V, 1024)
    blocks = 8
    if kind == "sampler":
        pcm, descs, _ = P.drum_bank()
        bank = O.Bank.sampler(pcm, descs, P.sampler_voices(sample_voices))
        bank.note_events(T.note_events_np(np.arange(sample_voices, dtype=np.uint32), P.sampler_keys(sample_voices), True))
    else:
        bank = O.Bank.welsh(P.welsh_voices(sample_voices))
        bank.note_events(P.note_on_all(sample_voices))
    chain = []
    if WORKLOADS[workload]["kind"] == "chain":  # config #3: the per-voice effect chain is part of the path
        chain = [O.Fx(k, p) for k, p in P.chain_fx_params(sample_voices)]

    def one_block():
        if chain:
            blk = bank.render(FRAMES)
            for fx in chain:
                fx.process(blk)
            O.mix(blk)
        else:
            bank.render_bus(FRAMES)

    # calibrate, then run ~seconds_target of CPU work
    t0 = time.perf_counter()
    one_block()
    dt = max(time.perf_counter() - t0, 1e-6)
    blocks = int(max(4, min(4096, seconds_target / dt)))
    t0 = time.perf_counter()
    for _ in range(blocks):
        one_block()
    el = time.perf_counter() - t0
    vf_per_s = sample_voices * FRAMES * blocks / el
    out = {
        "value": vf_per_s / V, "unit": "stereo frames/s", "cores": 1, "kind": "port",
        "sample": f"{sample_voices} of {V} voices x {blocks} blocks of {FRAMES} frames, f64 scalar oracle -O2, "
                  f"1 thread; frames/s scaled by {sample_voices}/{V} (measured {vf_per_s:.3e} voice-frames/s)",
    }
    # mode B: all host cores, on a sample large enough to keep every thread busy
    cores = int(L.oracle_hardware_concurrency()) or 1
    if kind != "sampler" and not chain:
        mt_voices = min(V, 64 * cores)
        mbank = O.Bank.welsh(P.welsh_voices(mt_voices))
        mbank.note_events(P.note_on_all(mt_voices))
        mt_blocks = int(max(2, min(512, 0.3 * seconds_target * vf_per_s * min(cores, 16) / (mt_voices * FRAMES))))
        t0 = time.perf_counter()
        for _ in range(mt_blocks):
            mbank.render_bus(FRAMES, threads=cores)
        el = time.perf_counter() - t0
        out["all_cores"] = {"value": mt_voices * FRAMES * mt_blocks / el / V, "cores": cores,
                            "sample": f"{mt_voices} voices x {mt_blocks} blocks, {cores} threads"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=172, help="default: one full 172-block project (44,032 frames)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="welsh-1m", choices=sorted(WORKLOADS))
    ap.add_argument("--voices", type=int, default=0, help="override the workload's total voice count")
    ap.add_argument("--materialise", action="store_true",
                    help="entity-boundary form: write every voice block to HBM, then run the separate mix kernels "
                         "(default: fused render+mix, no materialised voice blocks)")
    ap.add_argument("--interleaved", action="store_true", help="voice i uses patch i mod 32 inside every wavefront (generic per-lane kernel)")
    ap.add_argument("--strong", action="store_true",
                    help="multi-GPU: split the workload's voices over the ranks (default: weak scaling, every rank "
                         "holds the workload's voice count and the project grows with --gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render-ahead", action="store_true",
                    help="workloads with effect chains: render block b, then its effects (default: the render of block b+1 "
                         "is submitted to the side streams before the effects of block b)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    use_dist = world > 1 or os.environ.get("GROOVE_BENCH_FORCE_DIST") == "1"  # the latter: exercise the N>1 code path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    wl = dict(WORKLOADS[args.workload])
    V = args.voices or wl["voices"]
    weak = not args.strong
    V_total = V * world if weak else V   # voices of the whole project
    lo = V_total * rank // world
    hi = V_total * (rank + 1) // world
    ctx = E.Context(local_rank if use_dist else 0)
    if use_dist:
        # The library's own communicator (RCCL, dlopen'ed): one ncclReduce of the bus per render.  If it cannot
        # be set up on this node the reduce falls back to the launcher's process group (same RCCL collective
        # through torch, plus two staging copies of the bus, once per render) and the line says so.
        reduce_via = "groove_bus_reduce (RCCL ncclReduce on the ctx stream)"
        try:
            uid = [ctx.comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            uid = [None]
            reduce_via = f"torch.distributed.reduce (library communicator unavailable: {e})"
        dist.broadcast_object_list(uid, src=0)
        ok = [1]
        if uid[0] is not None:
            try:
                if os.environ.get("GROOVE_BENCH_BREAK_COMM") == "1":  # exercise the fallback below
                    raise RuntimeError("GROOVE_BENCH_BREAK_COMM=1")
                ctx.comm_init(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001
                ok = [0]
                reduce_via = f"torch.distributed.reduce (groove_comm_init failed: {e})"
        else:
            ok = [0]
        import torch
        flag = torch.tensor(ok, dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # every rank takes the same path
        own_comm = bool(int(flag.item()))
        if not own_comm and reduce_via.startswith("groove_bus_reduce"):
            reduce_via = "torch.distributed.reduce (another rank could not set up the library communicator)"

    fused = not args.materialise
    proj = Project(ctx, args.workload, lo, hi - lo, fused, grouped=not args.interleaved, render_ahead=not args.no_render_ahead)
    K, W = args.steps, args.warmup
    bus = ctx.bus((K + W) * FRAMES)

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            ctx.synchronize()

    for s in range(W):
        proj.step(bus, s * FRAMES)
    sync_all()
    # Where the event pair brackets the whole (pipelined) step, consecutive pairs would tile the timed
    # region: only its two ends are recorded (every record is a packet on the ctx stream, ~5 us of its
    # timeline each) and the average step is their span divided by the steps.
    span_mode = (fused and wl["kind"] != "chain") or (wl["kind"] == "chain" and not args.no_render_ahead)
    if span_mode:
        first_ev, last_ev = ctx.event(), ctx.event()
        pairs = [(first_ev if s == 0 else None, last_ev if s == K - 1 else None) for s in range(K)]
    else:
        pairs = [(ctx.event(), ctx.event()) for _ in range(K)]
    t0 = time.perf_counter()
    for s in range(K):
        proj.step(bus, (W + s) * FRAMES, pairs[s])
    if use_dist:
        if own_comm:
            ctx.bus_reduce(E._Slice(bus, W * FRAMES), K * FRAMES, 0)
        else:
            import torch
            host = bus.download()
            t = torch.from_numpy(host[W * FRAMES:(W + K) * FRAMES].copy()).cuda()
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                host[W * FRAMES:(W + K) * FRAMES] = t.cpu().numpy()
                bus.upload(host)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if span_mode:
        kern_ms = ctx.elapsed_ms(pairs[0][0], pairs[-1][1]) / K
    else:
        kern_ms = float(np.mean([ctx.elapsed_ms(a, b) for a, b in pairs]))
    if rank == 0:
        out_bus = bus.download()[W * FRAMES:]
        finite = bool(np.isfinite(out_bus).all())
        peak = float(np.abs(out_bus).max() / V_total)
        frames_total = K * FRAMES
        project_fps = frames_total / elapsed                    # frames of the merged project per second
        value = project_fps * (world if weak else 1)            # weak: every rank rendered frames_total bus frames of its shard
        n_local = hi - lo
        whole_step = span_mode and wl["kind"] in ("chain", "mixed")  # the events bracket the step, not one kernel
        dom_bytes = wl["bytes_per_vf"] if span_mode else wl["dominant_bytes"]
        achieved = dom_bytes * n_local * FRAMES / (kern_ms * 1e-3) / 1e9
        traffic = committed_traffic(args.workload, world)
        line = {
            "metric": "stereo frames/sec rendered (offline)", "value": value, "unit": "stereo frames/s",
            "x_realtime_44k1": project_fps / SR, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32 (f64 IIR state, u64 phase)", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {V_total} voices total, {FRAMES}-frame blocks, {SR} Hz, "
                                   f"{'fused render+mix' if fused else 'materialised blocks + mix kernels'}",
                       "voices_total": V_total, "voices_per_gpu": n_local,
                       "bus_reduce": (reduce_via if use_dist else "none (one rank)"),
                       "parallelism": (f"voices sharded x{world} ({'weak: ' + str(V) + ' voices per GPU, project grows with N' if weak else 'strong: fixed project split N ways'}), "
                                       "no data-path collective, 1 RCCL bus reduce per render")},
            "project_frames_per_s": project_fps,
            "voice_frames_per_s": project_fps * V_total,
            "path_effective_GBs": wl["bytes_per_vf"] * project_fps * V_total / 1e9,
            "roofline": {"bound": "hbm",
                         "kernel": ("welsh_render_uniform_kernel<fused, LFO mode, retune> (one kernel per base kind, run "
                                    "concurrently; class-specialised block bodies) + partial_rows/final" if fused and wl["kind"] == "welsh"
                                    else "whole step: render of block b+1 (side streams) beside the effect chain + mix of block b" if whole_step and wl["kind"] == "chain"
                                    else "whole step: the banks' fused render kernels side by side + their bus reductions" if whole_step
                                    else "render kernel of the first bank"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_voice_frame": dom_bytes, "algorithmic_bytes_per_step": dom_bytes * n_local * FRAMES,
                         "kernel_ms": kern_ms,
                         "traffic": (traffic or {}).get("bytes_per_step"), "traffic_unit": "bytes per step (PMC, committed pass)",
                         "traffic_source": (traffic or {}).get("source")},
            "output_check": {"finite": finite, "peak_abs_bus_over_V": peak},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        # The JSON line goes out last: RCCL prints a version banner through C stdio, which would otherwise
        # be flushed at exit, after this line.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
