"""The line the driver parses: bench.py's LAST stdout line is a compact object — the contract keys,
the headline's roofline and cpu_baseline, one row per other workload — gated here to 4,096 bytes and strict JSON; everything
else goes to bench_detail.json.  No GPU: the watchdog path is fed a stand-in child that emits round 4's whole measurement
(profiles/r04_bench_line_driver_command.json, 20 KB — the line the driver could not parse) through bench.emit()."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R4_LINE = os.path.join(REPO, "profiles", "r04_bench_line_driver_command.json")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")


def _strict(text):
    def refuse(name):
        raise ValueError(f"non-standard JSON constant {name}")
    return json.loads(text, parse_constant=refuse)


def _check_compact(text, n_gpus=1):
    assert len(text.encode()) <= 4096, len(text.encode())
    d = _strict(text)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert isinstance(d["value"], float) and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["dtype"].startswith("f32")
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["voices_total"] >= d["config"]["voices_per_gpu"]
    r = d["roofline"]
    for k in ("bound", "frac", "frac_is", "achieved", "peak", "unit", "algorithmic_bytes_per_step", "kernel_ms", "traffic",
              "hbm_physical_frac", "valu_achieved_frac", "frac_of_measured_bound", "bound_source", "traffic_source"):
        assert k in r, k
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    return d


def test_watchdog_path_prints_one_compact_strict_json_line(tmp_path):
    detail = tmp_path / "bench_detail.json"
    env = dict(os.environ, GROOVE_BENCH_FAKE_STALL_ONCE=str(tmp_path / "stalled_once"), GROOVE_BENCH_FAKE_STALL_EXIT="1",
               GROOVE_BENCH_FAKE_LINE=R4_LINE, GROOVE_BENCH_DETAIL=str(detail))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--watchdog-seconds", "60"],
                       env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    json_lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(json_lines) == 1 and r.stdout.strip().splitlines()[-1] == json_lines[0]   # one JSON line, and it is the last line
    d = _check_compact(json_lines[0])
    assert d["steps"] == 20 and d["warmup"] == 5 and abs(d["ms_per_step"] - 0.487207) < 1e-6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["sample"]
    assert d["watchdog"]["attempts"] == 2 and d["watchdog"]["killed"] == 1 and d["watchdog"]["tainted"] is True
    assert d["zero_segments"] == 0 and d["parity_rms"] < 1e-5
    assert [c["workload"] for c in d["configs"]][:4] == ["welsh-256", "chain-4096", "sampler-16384", "mixed-131072"]
    assert all(set(c) == {"workload", "ms_per_step", "frac", "hbm_physical_frac", "bus_rms_err"} for c in d["configs"])
    # the whole measurement is in the detail file, with the parent's watchdog record merged in
    full = json.load(open(detail))
    assert d["detail"] == "bench_detail.json" and full["timed_region"]["repeats"] == 7 and len(full["configs"]) == 6
    assert full["watchdog"]["attempts"] == 2 and "shard_curve" in full and "streams" in full


def test_compact_line_of_an_n_gpu_run_leads_with_weak_scaling():
    sys.path.insert(0, REPO)
    import bench
    line = json.load(open(R4_LINE))
    sec = lambda scaling, vt, vpg, ms: {"workload": "welsh-1m", "scaling": scaling, "voices_total": vt, "voices_per_gpu": vpg,  # noqa: E731
                                        "value": 256 / ms * 1e3, "unit": "stereo frames/s", "voice_frames_per_s": 256 / ms * 1e3 * vt, "ms_per_step": ms,
                                        "ms_per_step_repeats": [ms] * 7, "ms_per_step_by_rank": {"max": ms, "min": ms * 0.98, "all": [ms] * 8, "note": "x" * 300},
                                        "kernel_ms_rank0": ms, "rccl_ranks": 8, "bus_reduce": "groove_bus_reduce (RCCL ncclReduce on the ctx stream)" + "y" * 200,
                                        "bus_reduce_alone_ms": 0.2, "kernel_form": ["z" * 200]}
    line.update(n_gpus=8, rccl_ranks=8, sections={"strong": sec("strong", 1_000_000, 125_000, 0.13), "weak": sec("weak", 8_000_000, 1_000_000, 0.49),
                                                 "mixed-131072": sec("strong", 131072, 16384, 0.04)})
    for k in ("configs", "shard_curve", "cpu_baseline", "parity_vs_oracle"):
        line.pop(k, None)
    text = json.dumps(bench.compact_line(line), allow_nan=False)
    d = _check_compact(text, n_gpus=8)
    assert list(d["sections"]) == ["weak", "strong", "mixed-131072"]
    for s in d["sections"].values():
        assert set(s) == {"scaling", "value", "ms_per_step", "voices_total", "voices_per_gpu", "voice_frames_per_s", "rccl_ranks", "rank_ms_min", "rank_ms_max", "bus_reduce_alone_ms"}
        assert s["rccl_ranks"] == 8 and s["rank_ms_min"] <= s["rank_ms_max"]
    assert d["rccl_ranks"] == 8


def test_compact_line_never_carries_nan_or_infinity():
    sys.path.insert(0, REPO)
    import bench
    line = json.load(open(R4_LINE))
    line["roofline"]["traffic"] = float("nan")
    line["roofline"]["hbm_physical_frac"] = float("inf")
    line["configs"][0]["frac"] = float("-inf")
    d = _strict(json.dumps(bench.compact_line(line), allow_nan=False))
    assert d["roofline"]["traffic"] is None and d["roofline"]["hbm_physical_frac"] is None and d["configs"][0]["frac"] is None


def test_emit_drops_optional_tables_rather_than_exceed_the_limit(tmp_path, capsys, monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "d.json"))
    line = json.load(open(R4_LINE))
    line["configs"] = line["configs"] * 12   # 72 rows: too many for the line
    bench.emit(line)
    text = capsys.readouterr().out.strip()
    d = _check_compact(text)
    assert "configs" not in d and len(json.load(open(tmp_path / "d.json"))["configs"]) == 72


def test_measured_bound_prefers_this_runs_figure(monkeypatch):
    """roofline.frac_of_measured_bound: the issue bound measured on this box by the parent process (tools/micro/mix_bound, handed
    down in the environment) is used when present — `bound_source: "this run"` — else the committed one, labelled as such; the
    spec-rate fraction is a separate field and does not change with it."""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.delenv(bench.MIX_BOUND_ENV, raising=False)
    # (step times taken relative to the committed window profile's own instruction count, so that the test follows the profiles)
    prof, mb0 = bench.committed_profile("welsh-1m", window=(20, 5)), bench.measured_mix_bound()
    bound0 = prof["valu_per_step"] * mb0["ns_at_5_waves"] * 1e-9 / 1024.0 * 1e3
    slow, fast = bound0 / 0.88, bound0 * 0.9
    a = bench.roofline_block("welsh-1m", 1_000_000, slow, True, True, window=(20, 5))
    assert a["valu"]["bound_source"].startswith("committed: r") and 0.8 < a["valu"]["frac_of_measured_bound"] < 1.0
    run5 = round(mb0["ns_at_5_waves"] * 1.07, 4)   # (this run's box a little slower than the committed one: still under the step)
    monkeypatch.setenv(bench.MIX_BOUND_ENV, json.dumps({"ns_at_5_waves": run5, "ns_at_4_waves": round(run5 * 1.03, 4), "source": "this run"}))
    b = bench.roofline_block("welsh-1m", 1_000_000, slow, True, True, window=(20, 5))
    assert b["valu"]["bound_source"] == "this run" and b["valu"]["frac_of_measured_bound"] > a["valu"]["frac_of_measured_bound"]
    assert b["valu"]["achieved_frac"] == a["valu"]["achieved_frac"] and b["physical"]["valu_frac"] == b["valu"]["achieved_frac"]
    mbd = b["valu"]["measured_bound"]
    assert mbd["share_of_instructions_at_4_waves"] == 0.0         # (round 5: every per-kind kernel is budgeted for five waves per SIMD)
    assert mbd["ns_per_wave_instruction"]["weighted"] == run5
    assert "frac_of_measured_bound_flags" not in b["valu"] or all("above 1" not in f for f in b["valu"]["frac_of_measured_bound_flags"])
    c = bench.roofline_block("welsh-1m", 1_000_000, fast, True, True, window=(20, 5))   # a step faster than the bound is flagged, not hidden
    assert c["valu"]["frac_of_measured_bound"] > 1.0 and any("above 1" in f for f in c["valu"]["frac_of_measured_bound_flags"])
    d = bench.roofline_block("welsh-1m", 1_000_000, slow, True, True, window=(172, 4))
    assert any("different window" in f for f in d["valu"].get("frac_of_measured_bound_flags", [])) or d["valu"]["measured_bound"]["same_window"]


def test_round6_profiles_and_the_committed_line_describe_the_same_code():
    """every profile summary carries the identity of the library it was taken with — the hash of the library's
    sources as the library itself reports it (groove_debug_info) and the commit of the build — and the committed line's physical fields
    (traffic, instruction counts) are marked stale when they come from other code.  The round's summaries were all taken at ONE hash, the
    committed lines ran that very library, and their roofline says `stale_profile: false`."""
    import glob
    pdir = os.path.join(REPO, "profiles")
    summaries = sorted(glob.glob(os.path.join(pdir, "r06_*_summary.json")))
    names = {os.path.basename(f)[len("r06_"):-len("_summary.json")] for f in summaries}
    assert {"welsh-1m", "welsh-1m-window", "welsh-1m-library-window", "welsh-1m-materialised-window", "welsh-256", "chain-4096", "sampler-16384",
            "mixed-131072"} <= names
    ids = {f: json.load(open(f)).get("library") for f in summaries}
    assert all(i and i.get("source_hash") and i.get("git_head") for i in ids.values()), ids
    hashes = {i["source_hash"] for i in ids.values()}
    assert len(hashes) == 1, ids
    for name in ("r06_bench_line_driver_command.json", "r06_bench_line_default.json"):
        line = _strict(open(os.path.join(pdir, name)).read().strip().splitlines()[-1])
        assert line["library"]["source_hash"] in hashes, (name, line["library"], hashes)
        assert line["roofline"]["stale_profile"] is False and line["roofline"]["traffic_source"].startswith("r06_"), (name, line["roofline"])
        assert line["zero_segments"] == 0 and not line.get("tainted")
    # ... and a summary of other code is called stale: the same block against a different running hash
    sys.path.insert(0, REPO)
    import bench
    r = bench.roofline_block("welsh-1m", 1_000_000, 0.36, True, True, window=(20, 5), source_hash="0000000000000000")
    assert r["stale_profile"] is True and r["traffic_source_hash"] in hashes
    r = bench.roofline_block("welsh-1m", 1_000_000, 0.36, True, True, window=(20, 5), source_hash=next(iter(hashes)))
    assert r["stale_profile"] is False
