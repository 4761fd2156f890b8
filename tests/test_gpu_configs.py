"""GPU tier: BASELINE.json configs #3, #4 and #5 at their FULL sizes, through the C ABI.

  config #5  mixed-131072   the three banks into one bus exactly as bench.py builds them (groove_amd.projects):
                            the whole project on one GPU == the sum of the eight 16,384-voice shards
                            voice_range(131072, r, 8) (the multi-GPU partition), and a voice sample of every
                            kind against the oracle over the whole 172-block timeline;
  config #3  chain-4096     4,096 lanes with the real per-voice chain parameters, 100 blocks (the 11,025-frame
                            chorus line and the 4,410-frame delay line both wrap), 64 sampled lanes against
                            the oracle, and the render-ahead walk of bench.py against the plain walk;
  config #4  sampler-16384  the full-size bank, starts staggered by the hash rule, 344 blocks: sampled lanes
                            bit-exact, fused bus against the oracle's sum of all 16,384 voices.
"""
import ctypes as C

import numpy as np
import pytest

from groove_amd import abi_types as T, projects as PJ
from groove_amd.parallel import voice_range

pytestmark = pytest.mark.gpu
FR = 256


def _render(ctx, workload, sel, blocks, **kw):
    proj = PJ.Project(ctx, workload, sel, **kw)
    bus = ctx.bus(blocks * FR)
    for b in range(blocks):
        proj.step(bus, b * FR)
    out = bus.download().astype(np.float64)
    proj.destroy()
    bus.destroy()
    return out


def test_config5_mixed_131072_shards_and_sampled_parity(gpu_ctx, oracle):
    from oracle.projects import OracleProject
    V, blocks = 131072, 172
    whole = _render(gpu_ctx, "mixed-131072", np.arange(V), blocks)
    assert np.isfinite(whole).all() and np.abs(whole).max() / V > 1e-3
    acc = np.zeros_like(whole)
    for r in range(8):
        lo, hi = voice_range(V, r, 8)
        assert hi - lo == 16384
        acc += _render(gpu_ctx, "mixed-131072", np.arange(lo, hi), blocks)
    assert np.max(np.abs(acc - whole)) / V <= 1e-6, "sum of the eight shards differs from the single-GPU project"
    # a sample spread over the whole index range, every kind in it (stride 1021 is odd: i mod 4 cycles)
    sel = np.arange(0, V, 1021)[:128]
    kinds = PJ.split_kinds("mixed-131072", sel)
    assert min(len(kinds["welsh"]), len(kinds["fm"]), len(kinds["sampler"])) >= 24
    got = _render(gpu_ctx, "mixed-131072", sel, blocks) / len(sel)
    want = OracleProject("mixed-131072", sel).render(blocks) / len(sel)
    rms = np.sqrt(np.mean((got - want) ** 2))
    assert np.sqrt(np.mean(want ** 2)) > 1e-3 and rms <= 1e-5, rms
    # and per kind (a kind's error must not hide behind the others' signal)
    for kind_sel in (sel[sel % 4 < 2], sel[sel % 4 == 2], sel[sel % 4 == 3]):
        g = _render(gpu_ctx, "mixed-131072", kind_sel, 60) / len(kind_sel)
        w = OracleProject("mixed-131072", kind_sel).render(60) / len(kind_sel)
        assert np.sqrt(np.mean((g - w) ** 2)) <= 1e-5


def test_config3_chain_4096_full_size(gpu_ctx, oracle):
    """The chain has gain (four chorus taps, four recirculating combs: the reverb's output of a sustained
    voice reaches several times full scale), and it is linear, so it scales a voice's error like its signal.
    Three bars, 64 sampled lanes of the 4,096, per block:
      A  instrument output vs the oracle voice: per-lane RMS <= 1e-5 (voice level: full scale 1);
      B  the chain alone — the oracle chain fed the SAME input (the GPU's instrument block): per-lane RMS
         <= 2e-6 x max(1, lane peak): the chain's own arithmetic (f64 biquad, fp32 rings);
      C  end to end, oracle voices through the oracle chain, summed over the sample and divided by 64 (the
         bus normalisation of SURVEY 8d): RMS <= 1e-5 x max(1, peak of that bus)."""
    from groove_amd import entities as E
    V, blocks = 4096, 100   # 25,600 frames: the chorus line (11,025) and the delay line (4,410) wrap
    spec = PJ.plan("chain-4096", np.arange(V))[0]
    synth = E.WelshSynth(gpu_ctx, spec["params"])
    fx = [E.Effect(gpu_ctx, k, p) for k, p in spec["fx"]]
    block = gpu_ctx.block(V, FR)
    bus = gpu_ctx.bus(blocks * FR)
    lanes = np.arange(64) * 64 + 17   # 64 sampled lanes across the patch-major lane order
    sub = (T.WelshParams * 64)(*[spec["params"][int(i)] for i in lanes])
    ob = oracle.Bank.welsh(sub)
    chain_same, chain_e2e = [], []
    for k, p in spec["fx"]:
        sp = (T.FxParams * 64)(*[p[int(i)] for i in lanes])
        chain_same.append(oracle.Fx(k, sp))
        chain_e2e.append(oracle.Fx(k, sp))
    keys = np.array([e.key for e in spec["events"][0]], dtype=np.uint8)
    worst = {"A": 0.0, "B": 0.0, "C": 0.0}
    peak_out = 0.0
    for b in range(blocks):
        for blk, on in ((0, True), (PJ.NOTE_OFF_BLOCK, False)):
            if b == blk:
                synth.handle_midi_events(spec["events"][blk])
                ob.note_events(T.note_events_np(np.arange(64, dtype=np.uint32), keys[lanes], on))
        synth.generate_batch_values(block, FR)
        dry = block.download(FR)[:, :, lanes].astype(np.float64)
        want_dry = ob.render(FR)
        worst["A"] = max(worst["A"], float(np.sqrt(np.mean((dry - want_dry) ** 2, axis=(0, 1))).max()))
        for e in fx:
            e.transform_audio(block, FR)
        gpu_ctx.mix([block], FR, E._Slice(bus, b * FR))
        got = block.download(FR)[:, :, lanes].astype(np.float64)
        same = np.ascontiguousarray(dry)
        for e in chain_same:
            e.process(same)
        scale = np.maximum(1.0, np.abs(same).max(axis=(0, 1)))
        worst["B"] = max(worst["B"], float((np.sqrt(np.mean((got - same) ** 2, axis=(0, 1))) / scale).max()))
        e2e = want_dry
        for e in chain_e2e:
            e.process(e2e)
        gb, wb = got.sum(axis=2) / 64.0, e2e.sum(axis=2) / 64.0
        worst["C"] = max(worst["C"], float(np.sqrt(np.mean((gb - wb) ** 2)) / max(1.0, np.abs(wb).max())))
        peak_out = max(peak_out, float(np.abs(e2e).max()))
        # the bus is the lane sum of the block (fixed-order fp32 reduction)
        if b in (0, 43, 44, 99):
            full = block.download(FR).astype(np.float64)
            scale_bus = max(1.0, float(np.abs(full).max()))
            assert np.max(np.abs(bus.download()[b * FR:(b + 1) * FR].astype(np.float64) - full.sum(axis=2).T)) / V <= 1e-6 * scale_bus
    assert peak_out > 1.5, peak_out   # the chain does amplify: the bars above are the meaningful ones
    assert worst["A"] <= 1e-5 and worst["B"] <= 2e-6 and worst["C"] <= 1e-5, worst
    plain = bus.download().astype(np.float64)
    assert np.abs(plain[12000:]).max() > np.abs(plain[:256]).max() * 0.01  # the delayed taps do sound
    for e in fx:
        e.destroy()
    synth.destroy(); block.destroy(); bus.destroy()
    # bench.py's software-pipelined walks give the same bus, bit for bit: the paced one (renders two blocks ahead, the host
    # waits for the events itself, the bus reduction deferred into the next chain launch — four lane-sum rows, summed in the same
    # order by either path) and round 3's (one block ahead, device-side waits, a reduction launch per block)
    # ... and, round 5, the paced walk with the reverb's all-passes on the library's all-pass stream (the default for it: the
    # all-passes of block b beside the run of block b+1, the lane sums summed by the NEXT all-pass launch) or behind the run
    for paced, ap in ((True, True), (True, False), (False, False)):
        assert not gpu_ctx.fx_allpass_stream
        ahead = _render(gpu_ctx, "chain-4096", np.arange(V), blocks, paced=paced, allpass_stream=ap)
        assert np.array_equal(ahead.astype(np.float32).view(np.uint32), plain.astype(np.float32).view(np.uint32)), (paced, ap)
    assert not gpu_ctx.fx_allpass_stream   # (a ctx-wide knob the project restores)
    assert gpu_ctx.debug_info()["zero_segments"] == 0


def test_config4_sampler_16384_full_size(gpu_ctx, oracle):
    from groove_amd import entities as E
    from oracle.projects import OracleProject
    V, blocks = 16384, 344
    spec = PJ.plan("sampler-16384", np.arange(V))[0]
    assert sum(d.length for d in spec["descs"]) > 2_700_000   # the real bank's size class
    assert len(spec["events"]) == 172
    # fused form (what bench.py times) against the oracle's sum of ALL voices
    fused = _render(gpu_ctx, "sampler-16384", np.arange(V), blocks)
    op = OracleProject("sampler-16384", np.arange(V))
    want = np.concatenate([op.step(threads=8) for _ in range(blocks)], axis=0)
    assert np.abs(want).max() > 10.0
    assert np.max(np.abs(fused - want)) / V <= 1e-6 and np.sqrt(np.mean((fused - want) ** 2)) / V <= 1e-7
    # materialised form: sampled lanes bit-exact (the fetch is exact, the gain is 1)
    s = E.Sampler(gpu_ctx, spec["pcm"], spec["descs"], spec["params"])
    block = gpu_ctx.block(V, FR)
    lanes = (np.arange(96) * 167 + 5) % V
    ob = oracle.Bank.sampler(spec["pcm"], spec["descs"], (T.SamplerParams * 96)(*[spec["params"][int(i)] for i in lanes]))
    pos = {int(l): k for k, l in enumerate(lanes)}
    checked = 0
    for b in range(blocks):
        ev = spec["events"].get(b % 172) if b < 172 else None
        if ev is not None:
            s.handle_midi_events(ev)
            mine = [(pos[int(e.voice)], int(e.key), True) for e in ev if int(e.voice) in pos]
            if mine:
                ob.note_events(T.note_events(mine))
        s.generate_batch_values(block, FR)
        ref = ob.render(FR)
        if b % 7 == 0 or b < 8:
            got = block.download(FR)[:, :, lanes]
            assert np.array_equal(got, ref.astype(np.float32)), b
            checked += 1
    assert checked > 50
    s.destroy(); block.destroy()


def test_bench_n_gpu_code_path_on_one_gpu():
    """GROOVE_BENCH_FORCE_DIST=1: the whole N > 1 path of bench.py — gloo rendezvous, the library's RCCL communicator, the bus
    reduce inside the timed region — with ONE rank on this box's one GPU, so that it cannot rot between the rounds in which a
    multi-GPU node is available.  The line carries the three sections of one run (strong, weak, config #5), each with the
    communicator's size, the ranks' clocks and the reduce timed alone."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(), "bench_detail.json")
    env = dict(os.environ, GROOVE_BENCH_FORCE_DIST="1", MASTER_PORT="29541", GROOVE_BENCH_DETAIL=detail)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "1", "--voices", "200000",
                        "--no-configs", "--no-shard-curve", "--no-cpu-baseline", "--no-parity", "--no-watchdog"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    json_lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(json_lines) == 1 and p.stdout.strip().splitlines()[-1] == json_lines[0]
    sys.path.insert(0, os.path.join(repo, "tests"))
    from test_bench_line import _check_compact
    line = _check_compact(json_lines[0])          # <= 4,096 bytes, strict JSON, the contract keys, the roofline block
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["zero_segments"] == 0 and "tainted" not in line
    assert line["config"]["bus_reduce"].startswith("groove_bus_reduce")
    sec = line["sections"]
    assert list(sec) == ["weak", "strong", "mixed-131072"]     # the curve the design defends first
    for name, s in sec.items():
        assert s["rccl_ranks"] == 1 and s["bus_reduce_alone_ms"] > 0.0 and s["value"] > 0.0, (name, s)
        assert s["rank_ms_max"] >= s["rank_ms_min"] > 0.0
    assert sec["strong"]["voices_total"] == sec["weak"]["voices_total"] == 200000 and sec["mixed-131072"]["voices_per_gpu"] == 131072
    full = json.load(open(detail))                 # everything else is in the detail file
    assert sorted(full["sections"]) == ["mixed-131072", "strong", "weak"] and "streams" in full and "timed_region" in full
    assert full["sections"]["strong"]["ms_per_step_by_rank"]["max"] >= full["sections"]["strong"]["ms_per_step_by_rank"]["min"] > 0.0


def test_bench_default_line_is_compact_and_carries_this_runs_bound():
    """The driver's own command, through the watchdog parent: ONE JSON line on stdout, <= 4,096 bytes, strict JSON, with `roofline`
    (incl. the issue bound measured on THIS box by the parent's mix_bound child: `bound_source: "this run"`) and `cpu_baseline`."""
    import json
    import os
    import subprocess
    import sys
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(repo, "tools", "micro", "mix_bound")):
        pytest.fail("tools/micro/mix_bound is not built (__graft_entry__.build())")
    detail = os.path.join(tempfile.mkdtemp(), "bench_detail.json")
    env = dict(os.environ, GROOVE_BENCH_DETAIL=detail)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GROOVE_BENCH_CHILD"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "3",
                        "--no-configs", "--no-shard-curve"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    json_lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(json_lines) == 1 and p.stdout.strip().splitlines()[-1] == json_lines[0]
    sys.path.insert(0, os.path.join(repo, "tests"))
    from test_bench_line import _check_compact
    line = _check_compact(json_lines[0])
    assert line["config"]["voices_total"] == 1_000_000 and line["steps"] == 20 and line["warmup"] == 5
    r = line["roofline"]
    assert r["bound_source"] == "this run" and 0.5 < r["frac_of_measured_bound"] <= 1.0, r
    assert r["bound"] == "valu-issue" and 0.3 < r["valu_achieved_frac"] < 0.9 and 0.05 < r["hbm_physical_frac"] < 0.3
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["parity_rms"] <= 1e-5
    assert line["watchdog"]["attempts"] == 1 and line["watchdog"]["tainted"] is False and line["zero_segments"] == 0
    full = json.load(open(detail))
    assert full["roofline"]["valu"]["measured_bound"]["source"] == "this run"
