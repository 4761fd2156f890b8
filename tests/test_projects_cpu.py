"""CPU tier: the synthetic projects as data (groove_amd/projects.py `plan`) and bench.py's launcher.

  * every project voice index lands in exactly one bank lane, whatever the shard cut;
  * a shard's plan is the restriction of the whole project's plan (same patch / key / start block per voice);
  * the oracle instantiation of two shards sums to the oracle render of the whole (what the N > 1 path relies on);
  * `bench.py --gpus 2 --dry-launch` starts two ranks that rendezvous (gloo) without touching a GPU.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from groove_amd import projects as PJ, patches as P, abi_types as T
from groove_amd.parallel import voice_range

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workload_table_matches_baseline_configs():
    assert PJ.WORKLOADS["welsh-256"]["voices"] == 256
    assert PJ.WORKLOADS["chain-4096"]["voices"] == 4096
    assert PJ.WORKLOADS["sampler-16384"]["voices"] == 16384 and PJ.WORKLOADS["sampler-16384"]["blocks"] == 344
    assert PJ.WORKLOADS["mixed-131072"]["voices"] == 131072
    assert PJ.WORKLOADS["welsh-1m"]["voices"] == 1_000_000


def test_mixed_split_keeps_the_mix_in_every_contiguous_range():
    V = 131072
    for r in range(8):
        lo, hi = voice_range(V, r, 8)
        k = PJ.split_kinds("mixed-131072", np.arange(lo, hi))
        assert (len(k["welsh"]), len(k["fm"]), len(k["sampler"])) == (8192, 4096, 4096)
    whole = PJ.split_kinds("mixed-131072", np.arange(V))
    parts = [PJ.split_kinds("mixed-131072", np.arange(*voice_range(V, r, 8))) for r in range(8)]
    for kind in ("welsh", "fm", "sampler"):
        assert np.array_equal(np.sort(np.concatenate([p[kind] for p in parts])), whole[kind])
        assert np.array_equal(whole[kind], np.arange(len(whole[kind])))  # within-kind voice numbers are dense


def _voice_table(spec):
    """{within-kind voice number: (parameter bytes, events as (block, key, on))}"""
    import ctypes as C
    size = C.sizeof(spec["params"]._type_)
    raw = bytes(memoryview(spec["params"]))
    ev = {}
    for b, arr in spec["events"].items():
        for e in arr:
            ev.setdefault(int(e.voice), []).append((b, int(e.key), int(e.on)))
    return {int(v): (raw[i * size:(i + 1) * size], tuple(sorted(ev.get(i, [])))) for i, v in enumerate(spec["voice"])}


@pytest.mark.parametrize("workload,V", [("welsh-256", 256), ("sampler-16384", 2048), ("mixed-131072", 4096)])
def test_shard_plans_are_restrictions_of_the_whole(workload, V):
    whole = {s["kind"]: _voice_table(s) for s in PJ.plan(workload, np.arange(V), bank_scale=0.01)}
    seen = {k: {} for k in whole}
    for r in range(3):
        lo, hi = voice_range(V, r, 3)
        for s in PJ.plan(workload, np.arange(lo, hi), bank_scale=0.01):
            t = _voice_table(s)
            assert not (set(t) & set(seen[s["kind"]]))
            seen[s["kind"]].update(t)
    assert seen == whole


def test_sampler_start_blocks_follow_the_hash_rule():
    spec = PJ.plan("sampler-16384", np.arange(512), bank_scale=0.01)[0]
    start = P.sampler_start_block(512)
    for b, arr in spec["events"].items():
        lanes = sorted(int(e.voice) for e in arr)
        assert lanes == sorted(np.nonzero(start == b)[0].tolist())
        assert all(e.on == 1 and e.key == 69 + (int(e.voice) % 25) - 12 for e in arr)
    assert sum(len(a) for a in spec["events"].values()) == 512


def test_oracle_shards_sum_to_the_whole_project(oracle):
    from oracle.projects import OracleProject
    V, blocks = 96, 3
    whole = OracleProject("mixed-131072", np.arange(V), bank_scale=0.02).render(blocks)
    acc = np.zeros_like(whole)
    for r in range(2):
        acc += OracleProject("mixed-131072", np.arange(*voice_range(V, r, 2)), bank_scale=0.02).render(blocks)
    assert np.abs(whole).max() > 0.1
    assert np.max(np.abs(acc - whole)) <= 1e-9


def test_bench_dry_launch_starts_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dry_launch"] is True and line["ranks"] == 2 and line["world"] == 2
    assert line["voice_ranges"] == [[0, 500000], [500000, 1000000]]
    # the three measurements one `bench.py --gpus N` run makes (bench.py section_plan), with every rank in the communicator
    sec = line["sections"]
    assert list(sec) == ["weak", "strong", "mixed-131072"]     # the compact N > 1 line's order: the curve the design defends first
    compact_keys = {"scaling", "value", "ms_per_step", "voices_total", "voices_per_gpu", "voice_frames_per_s", "rccl_ranks", "rank_ms_min",
                    "rank_ms_max", "bus_reduce_alone_ms"}   # == bench.compact_line's section entries (tests/test_bench_line.py)
    assert all(set(s) == compact_keys | {"workload", "ranges"} for s in sec.values())
    assert sec["weak"]["scaling"] == "weak" and sec["strong"]["scaling"] == "strong" and sec["weak"]["voices_per_gpu"] == 1_000_000
    assert all(s["rccl_ranks"] == line["world"] for s in sec.values())
    assert sec["strong"]["voices_total"] == 1_000_000 and sec["strong"]["ranges"] == [[0, 500000], [500000, 1000000]]
    assert sec["weak"]["voices_total"] == 2_000_000 and sec["weak"]["ranges"] == [[0, 1000000], [1000000, 2000000]]
    assert sec["mixed-131072"]["voices_total"] == 131072 and sec["mixed-131072"]["ranges"] == [[0, 65536], [65536, 131072]]


def test_bench_has_one_collective_path_and_refuses_without_it():
    """A rank whose library communicator cannot be set up ends the job: every rank leaves with a non-zero code and says why, nothing is
    printed as a result — the torch.distributed-over-nccl fallback of rounds 3 - 5 is gone, so a line that
    exists measured groove_bus_reduce.  Here through --dry-launch (the rendezvous without a GPU; GROOVE_BENCH_BREAK_COMM names the
    rank that fails); the launcher does not retry a failure that is not a stall."""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "new_group(backend=\"nccl\")" not in src and "dist.reduce(" not in src      # no second collective backend in the file
    env = _clean_env(GROOVE_BENCH_BREAK_COMM="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0, p.stdout[-500:]
    assert "no fallback" in p.stderr and "GROOVE_BENCH_BREAK_COMM" in p.stderr and "another rank could not set up" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{") and "dry_launch" in ln]


def test_section_plan_is_config_5_at_eight_gpus():
    """BASELINE.json config #5: 131,072 mixed voices sharded over 8 GPUs = 16,384 contiguous voices per rank."""
    sys.path.insert(0, REPO)
    import bench
    for r in range(8):
        pl = bench.section_plan(8, r)
        assert pl["mixed-131072"]["range"] == [16384 * r, 16384 * (r + 1)]
        assert pl["strong"]["range"] == [125000 * r, 125000 * (r + 1)]
        assert pl["weak"]["range"] == [1000000 * r, 1000000 * (r + 1)] and pl["weak"]["voices_total"] == 8_000_000


def test_bench_refuses_a_world_size_that_is_not_the_gpu_count():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_bench_watchdog_kills_a_stalled_child_and_retries(tmp_path):
    """bench.py on one GPU runs its measurement in a child process; a child that does not finish in time is killed and
    the run starts again.  Here the child is a stand-in that hangs the first time and answers the second (no GPU)."""
    marker = tmp_path / "stalled_once"
    env = dict(os.environ, GROOVE_BENCH_FAKE_STALL_ONCE=str(marker))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--watchdog-seconds", "3"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["metric"] == "fake" and d["watchdog"] == {"attempts": 2, "killed": 1, "seconds_allowed": 3.0, "tainted": True}
    assert "killed" in r.stderr


def test_bench_watchdog_restarts_a_child_that_reports_a_stalled_stream(tmp_path):
    """... and a child whose library reported the stalled stream itself (groove_synchronize's deadline -> exit code 75) is
    replaced at once, without waiting for the watchdog's own limit."""
    import time
    marker = tmp_path / "stalled_once"
    env = dict(os.environ, GROOVE_BENCH_FAKE_STALL_ONCE=str(marker), GROOVE_BENCH_FAKE_STALL_EXIT="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--watchdog-seconds", "60"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["watchdog"]["attempts"] == 2 and d["watchdog"]["killed"] == 1 and d["watchdog"]["tainted"] is True
    assert "stalled stream" in r.stderr and time.time() - t0 < 50


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "GROOVE_BENCH_CHILD")}
    env.update(extra)
    return env


@pytest.mark.parametrize("mode", ["hang", "exit75"])
def test_bench_rank_launcher_has_a_deadline_and_restarts_all_ranks(tmp_path, mode):
    """bench.py --gpus 2 started plain launches its ranks itself: when one rank stalls (here a stand-in that hangs, or
    that reports a stalled stream, the first time), the whole group is killed at the deadline — exactly the PIDs started —
    and two FRESH ranks are started; the line says so."""
    marker = tmp_path / "stalled_once"
    env = _clean_env(GROOVE_BENCH_FAKE_STALL_ONCE=str(marker), GROOVE_BENCH_RANKS_DEADLINE_S="6")
    if mode == "exit75":
        env.update(GROOVE_BENCH_FAKE_STALL_EXIT="1", GROOVE_BENCH_RANKS_DEADLINE_S="120")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dry_launch"] is True and line["ranks"] == 2
    assert line["watchdog"]["attempts"] == 2 and line["watchdog"]["killed"] == 1 and line["watchdog"]["ranks"] == 2
    assert ("had not finished" in r.stderr) if mode == "hang" else ("stalled stream" in r.stderr)


def test_bench_ranks_under_an_external_launcher_are_supervised(tmp_path):
    """Started the way the driver starts it for N > 1 (torch.distributed.run: RANK / WORLD_SIZE / MASTER_* in the
    environment), every rank process only supervises a child; a child that reports a stalled stream makes ALL supervisors
    replace their children, and rank 0 prints the line of the second attempt."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    marker = tmp_path / "stalled_once"
    procs = []
    for rank in range(2):
        env = _clean_env(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                         GROOVE_BENCH_FAKE_STALL_ONCE=str(marker), GROOVE_BENCH_FAKE_STALL_EXIT="1", TORCHELASTIC_RUN_ID="test")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    line = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert line["dry_launch"] is True and line["ranks"] == 2
    assert line["watchdog"]["supervised_under_launcher"] is True and line["watchdog"]["attempts"] == 2 and line["watchdog"]["killed"] == 1
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # only rank 0 prints


def test_bench_line_reads_the_profile_of_its_own_window():
    """bench.py's roofline block: for the driver's command (--steps 20 --warmup 5) the physical figures come from the PMC
    summary collected with that very command (profiles/rNN_welsh-1m-window_summary.json), otherwise from the whole-timeline
    summary; the effective fraction, both physical fractions and the bound are all on the block."""
    sys.path.insert(0, REPO)
    import bench
    win = bench.committed_profile("welsh-1m", window=(20, 5))
    whole = bench.committed_profile("welsh-1m", window=(172, 4))
    assert win and win["source"].endswith("welsh-1m-window_summary.json") and win["same_window_as_this_run"] is True
    assert whole and whole["source"].endswith("_welsh-1m_summary.json") and whole["window_steps_warmup"] == [172, 4]
    assert win["valu_per_step"] > whole["valu_per_step"] > 1e8        # the window is the timeline's most expensive stretch
    assert win["traffic"] and win["valu_simd_ns_per_step"]["other_all_fast"] < win["valu_simd_ns_per_step"]["other_all_normal"]
    ms = win["valu_per_step"] / 0.6 / bench.VALU_ISSUE_PER_S * 1e3        # a step at 0.6 of the spec issue rate of THIS profile's instruction count
    r = bench.roofline_block("welsh-1m", 1_000_000, ms, True, True, window=(20, 5))
    assert r["bound"] == "valu-issue" and r["traffic_same_window"] is True
    assert 0.9 < r["frac"] < 2.6 and 0.05 < r["hbm_physical_frac"] < 0.35           # effective (can exceed 1: the fewer instructions a step needs, the higher) vs physical
    assert 0.55 < r["valu"]["achieved_frac"] < 0.65 and 0.7 < r["valu"]["cost_weighted_frac"]["low"] <= r["valu"]["cost_weighted_frac"]["high"] < 1.3
    lat = bench.roofline_block("sampler-16384", 16384, 0.0166, True, True)
    assert lat["bound"].startswith("latency") and lat["hbm_physical_frac"] < 0.25
    hbm = bench.roofline_block("chain-4096", 4096, 0.0647, True, True)
    assert hbm["bound"] == "hbm" and 0.3 < hbm["hbm_physical_frac"] < 0.5 and 0.4 < hbm["frac"] < 0.5
    # a shard of the workload scales the counted traffic and instructions with its voices
    half = bench.roofline_block("welsh-1m", 500_000, 0.6 * ms, True, True, window=(20, 5))
    assert abs(half["traffic"] / r["traffic"] - 0.5) < 1e-9
    sel = bench.spread_sample(1_000_000, 256)
    assert len(sel) == 256 and len(set((sel % 32).tolist())) == 32 and sel.max() < 1_000_000


def test_bench_under_torch_distributed_run_the_way_the_driver_starts_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`
    (the driver's command for N > 1; here with --dry-launch: the rendezvous only, no GPU): every rank supervises a child,
    rank 0 prints one line."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["ranks"] == 2 and line["voice_ranges"] == [[0, 500000], [500000, 1000000]]
    assert line["watchdog"]["supervised_under_launcher"] is True and line["watchdog"]["attempts"] == 1


def test_parity_sample_is_forced_to_the_timed_kernel_form():
    """bench.py timed_kernel_form: which tuning knobs make a few dozen sample voices run the Welsh kernel form the full-size
    workload's timed region ran (the library picks the form by bank size)."""
    import bench

    class Ctx:  # the library's defaults
        time_parallel_max_voices = 16384
        time_parallel_pair_min_voices = 3073
        split_max_waves = 1024
        pipeline_min_waves = 3800

    def forced(workload, v):
        t = bench.timed_kernel_form(Ctx(), workload, v)
        return t.form, t.set

    assert forced("welsh-256", 256)[1] == {}
    assert forced("sampler-16384", 16384)[1] == {}
    f, s = forced("chain-4096", 4096)
    assert "two voices per wavefront" in f and s == {"time_parallel_pair_min_voices": 1}
    f, s = forced("mixed-131072", 131072)   # 65,536 Welsh voices: role-split; the FM / sampler banks keep their time-parallel kernels
    assert "role-split" in f and s["time_parallel_max_voices"] == 1 and s["split_max_waves"] >= 1024
    f, s = forced("mixed-131072", 16384)    # 8,192 Welsh voices: two per wavefront
    assert "two voices per wavefront" in f
    f, s = forced("welsh-1m", 200_000)      # (under ~243,000 voices: the all-kinds kernel)
    assert "all kinds" in f and s["split_max_waves"] == 0
    f, s = forced("welsh-1m", 1_000_000)
    assert "per base kind" in f and s["pipeline_min_waves"] == 1 and s["time_parallel_max_voices"] == 1
    with bench.timed_kernel_form(Ctx(), "chain-4096", 4096) as t:
        assert t.ctx.time_parallel_pair_min_voices == 1
    assert t.ctx.time_parallel_pair_min_voices == 3073
