#!/usr/bin/env python3
"""bench.py — offline-render throughput of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload welsh-1m]

A "step" is one pass of the hot path over one 256-frame block of the whole project: every voice
ticks 256 frames (render kernels), its effect chain runs, and the mix bus sums the voice blocks
(Orchestrator::gather_audio).  `value` = stereo bus frames of the project rendered per second,
with all inputs (patch parameters, voice state, sample bank) resident in HBM before the timed
region.  The timed region is repeated (--repeats; default 7 for windows of up to 40 steps, 3 otherwise) from a reset state — W warm-up steps,
then exactly K timed steps, i.e. blocks W .. W+K-1 of the project's timeline each time — and
`value` is the median repeat; every repeat is on the line.

Workloads (groove_amd/projects.py):
    welsh-1m      1,000,000 Welsh voices (config-#2 voice rule) — north-star target, default
    welsh-256     config #2   (256 voices; 4 wavefronts: a latency config)
    chain-4096    config #3   (4,096 voices + BiQuad→Chorus→Delay→Reverb per voice)
    sampler-16384 config #4   (16,384 one-shot sampler voices over a shared bank, staggered starts)
    mixed-131072  config #5   (50 % Welsh / 25 % FM / 25 % sampler)
With the default workload on one GPU the line also carries `configs`: the other four workloads,
each timed over its WHOLE project timeline (172 blocks; 344 for the sampler), with the bus RMS error
of a voice sample against the CPU oracle (computed outside the timed region).

On one GPU (and not under a launcher) the measurement runs in a CHILD process under a watchdog: a child that has not
printed its line in time is killed and the measurement starts again (DESIGN.md section 7 says why); the line carries
`watchdog: {attempts, killed, seconds_allowed}`.  `--no-watchdog` runs in this process.

Multi-GPU (`--gpus N`): one process per GPU.  Started without WORLD_SIZE in the environment this
script launches the N rank processes itself (before anything touches a GPU); started under
`torch.distributed.run` it is one of the ranks.  The project's voices are cut into contiguous index
ranges, one per rank, no data-path collective, and the per-rank buses are summed with ONE RCCL
reduce over the timed region's frames, inside the timed region.
  default (strong scaling): the workload's voices are split N ways; `value` = project frames/s.
  --weak: every rank holds the workload's voice count, the project grows with N; `value` is still
      the merged project's frames per second (flat when scaling is ideal) and
      `voice_frames_per_s` is the unit that scales.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from groove_amd import entities as E, abi_types as T, projects as PJ  # noqa: E402  (no GPU call at import)
from groove_amd.parallel import voice_range  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PER_S = 1024 * 2.4e9 / 2.0  # 1,024 SIMD-32s, one wave64 VALU instruction per 2 cycles at 2.4 GHz (same guide)
FRAMES = T.BLOCK_FRAMES
SR = T.DEFAULT_SAMPLE_RATE
WORKLOADS = PJ.WORKLOADS


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


RANK_STALL_EXIT = 75   # EX_TEMPFAIL: a rank whose library reported a stalled stream (groove_synchronize's deadline) — worth a fresh attempt


def _stop(procs):
    """Terminate exactly the processes in `procs` (started by this script), then kill what is left."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def launch_ranks(n, argv, deadline_s=None, attempts=3):
    """Start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), pass rank 0's output through and
    return non-zero unless every rank exits cleanly.  Nothing here touches a GPU.  Every attempt has a deadline for the
    whole group: when it passes (or a rank reports a stalled stream, exit code 75), exactly the rank processes started
    here are killed and n FRESH ones are started, up to `attempts` times; rank 0's JSON line gets a `watchdog` entry."""
    deadline_s = deadline_s or float(os.environ.get("GROOVE_BENCH_RANKS_DEADLINE_S", "420"))
    killed = 0
    base_env = dict(os.environ)
    if "--dry-launch" not in argv and not os.environ.get("GROOVE_BENCH_FAKE_STALL_ONCE"):
        mix_bound_env(base_env)   # GPU 0's measured issue bound, before any rank starts
    for attempt in range(1, attempts + 1):
        port = str(_free_port())  # a fresh rendezvous per attempt
        procs = []
        for r in range(n):
            env = dict(base_env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", GROOVE_BENCH_CHILD="1",
                       GROOVE_BENCH_ATTEMPT=str(attempt))
            out = subprocess.PIPE if r == 0 else sys.stderr  # only rank 0 prints the JSON line
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
        failed = None
        out0 = b""
        pending = set(range(n))
        t0 = time.monotonic()
        timed_out = False
        while pending and failed is None and not timed_out:
            for r in sorted(pending):
                p = procs[r]
                try:
                    if r == 0:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    else:
                        p.wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if p.returncode != 0:
                    failed = (r, p.returncode)
                    break
            timed_out = bool(pending) and time.monotonic() - t0 > deadline_s
        if failed is None and not timed_out:
            text = out0.decode(errors="replace")
            lines = text.splitlines()
            for i in range(len(lines) - 1, -1, -1):
                if lines[i].startswith("{"):
                    try:
                        lines[i] = add_watchdog(lines[i], {"attempts": attempt, "killed": killed, "seconds_allowed": deadline_s, "ranks": n,
                                                           "tainted": killed > 0})
                    except Exception:
                        pass
                    break
            sys.stdout.write("\n".join(lines) + ("\n" if lines else ""))
            sys.stdout.flush()
            return 0
        still = sorted(pending)
        _stop([procs[r] for r in still])
        if timed_out:
            killed += 1
            sys.stderr.write(f"bench.py: attempt {attempt}: ranks {still} had not finished after {deadline_s:.0f} s; killed, "
                             f"{'starting fresh ranks' if attempt < attempts else 'giving up'}\n")
            continue
        if failed[1] == RANK_STALL_EXIT:
            killed += 1
            sys.stderr.write(f"bench.py: attempt {attempt}: rank {failed[0]} reported a stalled stream; "
                             f"{'starting fresh ranks' if attempt < attempts else 'giving up'}\n")
            continue
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}; {n} ranks were asked for\n")
        sys.stdout.write(out0.decode(errors="replace"))
        sys.stdout.flush()
        return 1
    print(json.dumps({"error": f"bench: {attempts} attempts of {n} ranks did not finish", "watchdog": {"attempts": attempts, "killed": killed, "ranks": n}}), flush=True)
    return 3


def supervise_rank(argv, rank, world, deadline_s=None, attempts=3):
    """Under an external launcher (torch.distributed.run: RANK / WORLD_SIZE already set) this process IS a rank.  It then
    only SUPERVISES: the measurement of the rank runs in a child process (this one never touches a GPU), the supervisors
    talk to each other over gloo on the launcher's rendezvous, and when a child stalls or the group's deadline passes
    every supervisor kills its own child (exactly that PID) and all start fresh children on a fresh port — what
    launch_ranks does when this script starts the ranks itself.  Rank 0 prints its child's line with `watchdog` added."""
    import torch
    import torch.distributed as dist
    deadline_s = deadline_s or float(os.environ.get("GROOVE_BENCH_RANKS_DEADLINE_S", "420"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    killed = 0
    rc_final = 3
    mb = [None]
    if rank == 0 and "--dry-launch" not in argv and not os.environ.get("GROOVE_BENCH_FAKE_STALL_ONCE"):
        mb = [mix_bound_env({}, int(os.environ.get("LOCAL_RANK", "0"))).get(MIX_BOUND_ENV)]  # rank 0's GPU, before any child starts
    dist.broadcast_object_list(mb, src=0)
    if mb[0]:
        os.environ[MIX_BOUND_ENV] = mb[0]
    for attempt in range(1, attempts + 1):
        port = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(port, src=0)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}  # the children host their own store
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]), GROOVE_BENCH_CHILD="1", GROOVE_BENCH_ATTEMPT=str(attempt),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                 stdout=subprocess.PIPE if rank == 0 else sys.stderr)
        out0 = []
        if rank == 0:  # drain the pipe while waiting
            import threading
            th = threading.Thread(target=lambda: out0.append(child.stdout.read()), daemon=True)
            th.start()
        t0 = time.monotonic()
        RUNNING, OK, STALL, FAIL = 0, 1, 2, 3
        while True:
            rc = child.poll()
            mine = RUNNING if rc is None else OK if rc == 0 else STALL if rc == RANK_STALL_EXIT else FAIL
            if mine == RUNNING and time.monotonic() - t0 > deadline_s:
                mine = STALL
            t = torch.tensor([mine, -mine], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst, best = int(t[0]), -int(t[1])
            if worst >= STALL or best == OK:
                break
            time.sleep(0.5)
        if worst < STALL:  # every child finished cleanly
            if rank == 0:
                th.join(timeout=10)
                text = (out0[0] if out0 else b"").decode(errors="replace")
                lines = text.splitlines()
                for i in range(len(lines) - 1, -1, -1):
                    if lines[i].startswith("{"):
                        try:
                            lines[i] = add_watchdog(lines[i], {"attempts": attempt, "killed": killed, "seconds_allowed": deadline_s, "ranks": world,
                                                               "tainted": killed > 0, "supervised_under_launcher": True})
                        except Exception:
                            pass
                        break
                sys.stdout.write("\n".join(lines) + ("\n" if lines else ""))
                sys.stdout.flush()
            rc_final = 0
            break
        _stop([child])
        killed += 1
        if worst == FAIL:
            if rank == 0:
                sys.stderr.write(f"bench.py: attempt {attempt}: a rank failed (not a stall); giving up\n")
            rc_final = 1
            break
        if rank == 0:
            sys.stderr.write(f"bench.py: attempt {attempt}: a rank stalled or the {deadline_s:.0f} s deadline passed; every rank's child was "
                             f"killed, {'starting fresh ones' if attempt < attempts else 'giving up'}\n")
    if rc_final == 3 and rank == 0:
        print(json.dumps({"error": f"bench: {attempts} attempts of {world} supervised ranks did not finish",
                          "watchdog": {"attempts": attempts, "killed": killed, "ranks": world}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return rc_final


def leave_unless_every_rank_has_its_communicator(dist, torch, rank, why, ctx=None):
    """ONE collective path (round 6): `why` is None on a rank whose library communicator stands.  The ranks agree over the host-side
    group; if any of them failed, every rank says so and ends with code 4 — there is no second backend, so a line that exists
    measured groove_bus_reduce.  (A rank that has touched the GPU is never re-exec'ed: the process ends, the supervisor decides.)"""
    flag = torch.tensor([0 if why else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()):
        return
    sys.stderr.write(f"bench.py rank {rank}: {why or 'another rank could not set up the library communicator'}; "
                     "there is one collective path (groove_bus_reduce) and no fallback: leaving with code 4\n")
    sys.stderr.flush()
    if ctx is not None:
        ctx.close()
    dist.destroy_process_group()
    sys.exit(4)


def dry_launch(world, rank):
    """--dry-launch: the rendezvous of the N>1 path without a GPU (gloo); rank 0 prints how many ranks joined."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    # the real run's one way out when a rank's communicator cannot be set up (Dist.make_context), without a GPU:
    # GROOVE_BENCH_BREAK_COMM=<rank> makes that rank fail, every rank learns it and leaves with code 4 — no line, no fallback
    broken = os.environ.get("GROOVE_BENCH_BREAK_COMM")
    if broken is not None:
        leave_unless_every_rank_has_its_communicator(dist, torch, rank, "GROOVE_BENCH_BREAK_COMM" if str(rank) == broken or broken == "all" else None)
    lo, hi = voice_range(WORKLOADS["welsh-1m"]["voices"], rank, world)
    spans = [None] * world
    dist.all_gather_object(spans, (lo, hi))
    plans = [None] * world
    dist.all_gather_object(plans, section_plan(world, rank))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # the real run's `sections`, as far as they exist without a GPU: every rank's voice range per section, and the number of
        # ranks that joined in the place of the communicator's size (`rccl_ranks`: ncclCommCount in the real run)
        # ... in the compact line's own shape and order (compact_line: the weak-scaling section first), the measured fields null
        sections = {}
        for name in ("weak", "strong", "mixed-131072"):
            rng = [pl[name]["range"] for pl in plans]
            sections[name] = {"scaling": "weak" if name == "weak" else "strong", "value": None, "ms_per_step": None,
                              "voices_total": plans[0][name]["voices_total"], "voices_per_gpu": rng[0][1] - rng[0][0],
                              "voice_frames_per_s": None, "rccl_ranks": int(t.item()), "rank_ms_min": None, "rank_ms_max": None,
                              "bus_reduce_alone_ms": None, "workload": plans[0][name]["workload"], "ranges": rng}
        print(json.dumps({"dry_launch": True, "ranks": int(t.item()), "world": world, "voice_ranges": spans, "sections": sections}), flush=True)
    return 0 if int(t.item()) == world else 1


# ------------------------------------------------------------------------------------------ the line
DETAIL_PATH = os.environ.get("GROOVE_BENCH_DETAIL", os.path.join(REPO, "bench_detail.json"))
COMPACT_LIMIT = 4096   # bytes; the final stdout line is gated to this (tests/test_bench_line.py): the driver parses that line


def _r(x, digits=6):
    """Round a float to `digits` significant digits for the compact line (the detail file keeps every digit)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    return x


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def compact_line(line):
    """The line the driver parses: the contract keys, the headline's roofline and cpu_baseline, one row per other workload —
    and nothing else.  Every repeat, instruction mix, note and stream table of the run is in bench_detail.json (`detail`).
    Model: the reference's own perf print, two figures on two lines (/root/reference/src/bin/groove-cli.rs:123-139)."""
    if "error" in line and "metric" not in line:
        return line
    cfg, roof = line.get("config") or {}, line.get("roofline") or {}
    valu = roof.get("valu") or {}
    forms = cfg.get("kernel_form") or []
    out = {k: _r(line.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                       "vs_baseline", "dtype", "data", "x_realtime_44k1")}
    out["config"] = {"workload": _short(cfg.get("workload"), 120), "voices_total": cfg.get("voices_total"), "voices_per_gpu": cfg.get("voices_per_gpu"),
                     "kernel_form": _short("; ".join(forms) if isinstance(forms, list) else forms, 140), "bus_reduce": _short(cfg.get("bus_reduce"), 80)}
    out["roofline"] = {"bound": _short(roof.get("bound"), 24), "frac": _r(roof.get("frac")),
                       "frac_is": "effective: SURVEY 8d algorithmic bytes / time (fused kernels do not move them)",
                       "achieved": _r(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"),
                       "algorithmic_bytes_per_step": _r(roof.get("algorithmic_bytes_per_step")), "kernel_ms": _r(roof.get("kernel_ms")),
                       "traffic": _r(roof.get("traffic")), "traffic_source": roof.get("traffic_source"),
                       "traffic_source_head": (roof.get("traffic_source_head") or "")[:12] or None, "stale_profile": roof.get("stale_profile"),
                       "hbm_physical_frac": _r(roof.get("hbm_physical_frac"), 4), "valu_achieved_frac": _r(valu.get("achieved_frac"), 4),
                       "frac_of_measured_bound": _r(valu.get("frac_of_measured_bound"), 4), "bound_source": valu.get("bound_source"),
                       "measured_bound_ms": _r((valu.get("measured_bound") or {}).get("bound_ms"), 4)}
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": _short(cb.get("sample"), 110),
                               "all_cores_value": _r((cb.get("all_cores") or {}).get("value")), "all_cores": (cb.get("all_cores") or {}).get("cores"),
                               "all_cores_efficiency": _r((cb.get("all_cores") or {}).get("parallel_efficiency"), 3)}
    par = line.get("parity_vs_oracle")
    if par:
        out["parity_rms"] = _r(par.get("bus_rms_err"), 4)
        out["parity_sample"] = f"{par.get('voices_sampled')} voices x {par.get('blocks')} blocks, same kernel form, bus / voices"
    out["zero_segments"] = line.get("zero_segments")
    lib_id = line.get("library") or {}
    out["library"] = {"source_hash": lib_id.get("source_hash"), "git_head": (lib_id.get("git_head") or "")[:12] or None, "dirty": lib_id.get("dirty")}
    if line.get("tainted"):
        out["tainted"] = True
    if line.get("configs"):
        out["configs"] = [{"workload": c.get("workload"), "ms_per_step": _r(c.get("ms_per_step"), 4), "frac": _r(c.get("frac"), 4),
                           "hbm_physical_frac": _r(c.get("hbm_physical_frac"), 3),
                           "bus_rms_err": _r((c.get("parity_vs_oracle") or {}).get("bus_rms_err"), 3)} for c in line["configs"]]
    sc = line.get("shard_curve")
    if sc:
        out["shard_curve_ms"] = {str(r_["voices_per_gpu"]): _r(r_["ms_per_step"], 4) for r_ in sc.get("welsh-1m", [])}
        if "mixed-131072_shard_of_8" in sc:
            out["shard_curve_ms"]["mixed_16384"] = _r(sc["mixed-131072_shard_of_8"]["ms_per_step"], 4)
    if line.get("sections"):
        # N > 1: the curve this design defends first (weak: the workload's voice count on every rank), then strong and config #5
        order = [k for k in ("weak", "strong", "mixed-131072") if k in line["sections"]]
        out["sections"] = {k: {"scaling": v.get("scaling"), "value": _r(v.get("value")), "ms_per_step": _r(v.get("ms_per_step"), 5),
                               "voices_total": v.get("voices_total"), "voices_per_gpu": v.get("voices_per_gpu"),
                               "voice_frames_per_s": _r(v.get("voice_frames_per_s")), "rccl_ranks": v.get("rccl_ranks"),
                               "rank_ms_min": _r((v.get("ms_per_step_by_rank") or {}).get("min"), 5),
                               "rank_ms_max": _r((v.get("ms_per_step_by_rank") or {}).get("max"), 5),
                               "bus_reduce_alone_ms": _r(v.get("bus_reduce_alone_ms"), 4)}
                           for k, v in ((k, line["sections"][k]) for k in order)}
        out["rccl_ranks"] = line.get("rccl_ranks")
    if line.get("watchdog"):
        out["watchdog"] = line["watchdog"]
    out["detail"] = os.path.basename(DETAIL_PATH)
    return out


def emit(line):
    """Write the whole measurement to bench_detail.json, then print the compact line — the LAST line of stdout, the only
    JSON line on it."""
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(line, f, indent=1)
            f.write("\n")
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {DETAIL_PATH}: {e}\n")
    text = json.dumps(compact_line(line), allow_nan=False, separators=(", ", ": "))
    if len(text.encode()) > COMPACT_LIMIT:   # never print a line the driver cannot take: drop the optional tables, keep the contract
        c = compact_line(line)
        for k in ("shard_curve_ms", "configs", "parity_sample"):
            c.pop(k, None)
            text = json.dumps(c, allow_nan=False, separators=(", ", ": "))
            if len(text.encode()) <= COMPACT_LIMIT:
                break
    print(text, flush=True)


def add_watchdog(text, watchdog):
    """A parent that supervised the measurement adds its `watchdog` record to the child's compact line (and to the detail file)."""
    d = json.loads(text)
    d["watchdog"] = watchdog
    if d.get("detail"):
        try:
            full = json.load(open(DETAIL_PATH))
            full["watchdog"] = watchdog
            with open(DETAIL_PATH, "w") as f:
                json.dump(full, f, indent=1)
                f.write("\n")
        except Exception:  # noqa: BLE001
            pass
    return json.dumps(d, allow_nan=False, separators=(", ", ": "))



# ------------------------------------------------------------------------------------------ evidence
def library_identity(debug_info=None):
    """Which code this is: the hash of the library's sources as the loaded library itself reports it (groove_debug_info; compiled in by
    groove_amd/Makefile) and the commit the in-tree binaries were built at (groove_amd/build_id.json, written by __graft_entry__.build()
    where .git exists).  On the line, and in every profile summary (tools/summarize_prof.py): a summary whose hash differs from the
    running library's was taken with other device code (`roofline.stale_profile`)."""
    out = {"source_hash": (debug_info or {}).get("source_hash")}
    try:
        b = json.load(open(os.path.join(REPO, "groove_amd", "build_id.json")))
        out.update(git_head=b.get("git_head"), dirty=b.get("dirty"), libgroove_hip_sha256_16=b.get("libgroove_hip_sha256_16"))
    except Exception:  # noqa: BLE001
        out.update(git_head=None)
    return out


def committed_profile(workload, window=None):
    """The committed rocprofv3 summary of a workload (profiles/rNN_<workload>_summary.json, or round 1's single
    summary for welsh-1m): HBM traffic per step (FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM) and the VALU / SALU
    wave-instruction counts per step of the PMC passes.  `window` = (steps, warmup): a summary collected with exactly
    that command (profiles/rNN_<workload>-window_summary.json, e.g. the driver's --steps 20 --warmup 5) is preferred when
    the run being described times the same blocks; otherwise the whole-timeline summary is used and the line says so."""
    import re
    pdir = os.path.join(REPO, "profiles")
    names = os.listdir(pdir) if os.path.isdir(pdir) else []

    def newest(pattern):
        found = sorted((int(m.group(1)), n) for n in names for m in [re.match(pattern, n)] if m)
        for _, n in reversed(found):
            try:
                return n, json.load(open(os.path.join(pdir, n)))
            except Exception:
                continue
        return None, None

    name, d = None, None
    if window is not None:
        name, d = newest(rf"^r(\d+)_{re.escape(workload)}-window_summary\.json$")
        if d is not None and [d.get("steps"), d.get("warmup")] != list(window):
            name, d = None, None
    if d is None:
        name, d = newest(rf"^r(\d+)_{re.escape(workload)}_summary\.json$")
    if d is None and workload == "welsh-1m":
        name, d = newest(r"^r(\d+)_summary\.json$")  # round 1's layout: one summary, of the default workload
    if d is None:
        return None
    out = {"source": name, "library": d.get("library"), "window_steps_warmup": [d.get("steps"), d.get("warmup")],
           "same_window_as_this_run": window is not None and [d.get("steps"), d.get("warmup")] == list(window),
           "traffic": (d.get("hbm_traffic_bytes_per_step") or {}).get("total_corrected")}
    ins = d.get("instructions_per_step") or {}
    valu, salu = ins.get("valu_wave_insts") or 0.0, ins.get("salu_wave_insts") or 0.0
    if not valu:  # round-1 layout: per-dispatch means of the (once-per-step) kernels
        for k, v in (d.get("sq") or {}).items():
            if "render" in k or "partial_" in k or "mix_" in k or "fx_" in k:
                m = v.get("mean_per_dispatch", {})
                valu += m.get("SQ_INSTS_VALU", 0.0)
                salu += m.get("SQ_INSTS_SALU", 0.0)
    out["valu_per_step"], out["salu_per_step"] = (valu or None), (salu or None)
    # VALU wave-instructions per step of each kernel of the step (the weights of the measured issue bound below)
    steps_per_run = d.get("steps_per_run") or 0.0
    by_kernel = {}
    if steps_per_run:
        for k, v in (d.get("sq") or {}).items():
            n = (v.get("mean_per_dispatch") or {}).get("SQ_INSTS_VALU", 0.0) * (v.get("dispatches") or 0) / steps_per_run
            if n and ("render" in k or "partial_" in k):
                by_kernel[k] = n
    out["valu_by_kernel"] = by_kernel or None
    out["valu_mix_per_step"] = d.get("valu_mix_per_step")   # measured classes (f64 / conversions / transcendental / ...), if the pass was run
    out["valu_simd_ns_per_step"] = (d.get("valu_mix_per_step") or {}).get("cost_weighted_simd_ns")
    out["clock_ghz"] = (d.get("clock_in_counter_pass") or {}).get("ghz_weighted_long_kernels")
    return out


MIX_BOUND_BIN = os.path.join(REPO, "tools", "micro", "mix_bound")
MIX_BOUND_ENV = "GROOVE_BENCH_MIX_BOUND"


def run_mix_bound(device=0, timeout=60.0):
    """Run tools/micro/mix_bound (built by __graft_entry__.build()) as a child process on GPU `device` and return its figures
    ({"ns_at_5_waves", "ns_at_4_waves", "source": "this run"}), or None.  Called by the watchdog PARENT / the launcher /
    the rank-0 supervisor — processes that never touch a GPU themselves — BEFORE the measurement children start, so that the
    issue bound and the step it is compared with come from the same box, the same minute and the same clocks."""
    if not os.path.exists(MIX_BOUND_BIN) or os.environ.get("GROOVE_BENCH_NO_MIX_BOUND") == "1":
        return None
    try:
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(device))
        out = subprocess.run([MIX_BOUND_BIN], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, text=True)
        if out.returncode != 0:
            return None
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])["ns_per_wave_instruction"]
        r = {"ns_at_5_waves": min(d["5"].values()), "ns_at_4_waves": min(d["4"].values()), "source": "this run"}
        return r if 0.3 < r["ns_at_5_waves"] < 10.0 and 0.3 < r["ns_at_4_waves"] < 10.0 else None
    except Exception:  # noqa: BLE001  (no binary for this host, no GPU, a garbled line: the committed figure is used, and the line says so)
        return None


def mix_bound_env(env, device=0):
    """Put this box's measured issue bound into a child's environment (the children of run_under_watchdog / launch_ranks /
    supervise_rank read it in measured_mix_bound)."""
    if MIX_BOUND_ENV not in env:
        mb = run_mix_bound(device)
        if mb:
            env[MIX_BOUND_ENV] = json.dumps(mb)
    return env


def measured_mix_bound():
    """ns of SIMD time per wave-instruction of the million-voice window's VALU class mix (tools/micro/mix_bound), the best
    variant at 5 and at 4 waves per SIMD: measured on THIS box by the parent of this process just before it started
    (`source: "this run"`), else the newest committed profiles/rNN_mix_bound.json (`source: "committed: <file>"`)."""
    import re
    try:
        d = json.loads(os.environ.get(MIX_BOUND_ENV, ""))
        if d.get("source") == "this run":
            return {"ns_at_5_waves": float(d["ns_at_5_waves"]), "ns_at_4_waves": float(d["ns_at_4_waves"]), "source": "this run"}
    except Exception:  # noqa: BLE001
        pass
    pdir = os.path.join(REPO, "profiles")
    found = sorted((int(m.group(1)), n) for n in (os.listdir(pdir) if os.path.isdir(pdir) else []) for m in [re.match(r"^r(\d+)_mix_bound\.json$", n)] if m)
    for _, n in reversed(found):
        try:
            d = json.load(open(os.path.join(pdir, n)))["ns_per_wave_instruction"]
            return {"ns_at_5_waves": min(d["5"].values()), "ns_at_4_waves": min(d["4"].values()), "source": "committed: " + n}
        except Exception:
            continue
    return None


def spread_sample(v_total, count):
    """`count` project voice indices spread over [0, v_total) with an ODD stride, so that the sample meets every
    residue of the voice rules' moduli (32 patches, 16 FM patches, 4 kinds, 60 buffers ...)."""
    count = min(count, v_total)
    stride = max(1, v_total // count)
    stride -= 1 - (stride & 1) if stride > 1 else 0
    return np.unique((np.arange(count, dtype=np.int64) * stride + stride // 2) % v_total)


def under_profiler():
    """rocprofv3 preloads its tool library into the program after `--`: nothing may start a child process from here then
    (the child would be profiled instead, and a re-launch from under --pmc is the forbidden exec of DESIGN.md section 7)."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_")) for k in os.environ)


class timed_kernel_form:
    """Make a SMALL sample project run the Welsh kernel form the full-size workload's timed region runs.  The library picks the
    form by bank size: time-parallel with one voice per wavefront (up to 3,072 voices) or two (up to
    16,384), role-split (up to 65,536), all kinds in one serial launch (up to ~243,000), one launch per base kind with
    pipelined blocks above — and a sample of a few dozen voices would always take the first.  The tuning knobs of the C ABI
    force the timed bank's form for the duration.  (A Welsh limit of ONE voice, not zero: zero would switch the FM and sampler
    banks of a mixed project to their serial kernels too.)"""

    def __init__(self, ctx, workload, v_total):
        self.ctx = ctx
        kind = WORKLOADS[workload]["kind"]
        vw = {"welsh": v_total, "chain": v_total, "mixed": v_total // 2}.get(kind, 0)  # Welsh voices of the timed bank
        tp_max, pair_min = ctx.time_parallel_max_voices, ctx.time_parallel_pair_min_voices
        split_max, pipe_min = ctx.split_max_waves * 64, ctx.pipeline_min_waves * 64
        self.set = {}
        if vw == 0 or (vw <= tp_max and (pair_min == 0 or vw < pair_min)):
            self.form = "default (time-parallel, one voice per wavefront)"
        elif vw <= tp_max + tp_max * 3 // 8 and pair_min:  # (the paired form's limit: groove_hip.hip use_tp; the projects' banks are laid out synth by synth)
            self.form, self.set = "time-parallel, two voices per wavefront", {"time_parallel_pair_min_voices": 1}
        elif vw <= split_max:
            self.form, self.set = "role-split (four wavefronts per 64 voices)", {"time_parallel_max_voices": 1, "split_max_waves": max(ctx.split_max_waves, 4096)}
        elif vw < pipe_min:
            self.form, self.set = "serial, all kinds in one launch", {"time_parallel_max_voices": 1, "split_max_waves": 0}
        else:
            self.form, self.set = "serial, one launch per base kind, blocks pipelined", {"time_parallel_max_voices": 1, "pipeline_min_waves": 1}
        self.force = bool(self.set)

    def __enter__(self):
        self.old = {k: getattr(self.ctx, k) for k in self.set}
        for k, v in self.set.items():
            setattr(self.ctx, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            setattr(self.ctx, k, v)


def sampled_parity(ctx, workload, v_total, blocks, sample=64, fused=True, grouped=True):
    """Bus RMS error (normalised by the sample size) of `sample` voices of the project, spread over its
    whole index range, rendered by the product path — by the SAME kernel form the timed region ran — and by the CPU
    oracle over the first `blocks` blocks of the timeline.  Outside every timed region."""
    from oracle.projects import OracleProject
    phase(f"parity sample of {workload} ({sample} voices, {blocks} blocks)")
    sel = spread_sample(v_total, sample)
    with timed_kernel_form(ctx, workload, v_total) as forced:
        proj = PJ.Project(ctx, workload, sel, fused=fused, grouped=grouped)
        forms = sorted({inst.kernel_form(FRAMES, fused and not fx) for inst, _, fx, _ in proj.banks})
        bus = ctx.bus(blocks * FRAMES)
        for b in range(blocks):
            proj.step(bus, b * FRAMES)
        got = bus.download().astype(np.float64) / len(sel)
        proj.destroy()
        bus.destroy()
    want = OracleProject(workload, sel, grouped=grouped).render(blocks) / len(sel)
    rms = float(np.sqrt(np.mean((got - want) ** 2)))
    peak = float(np.abs(want).max())
    # the worst single block, so that a window (e.g. the release after the note-off) cannot hide in the average
    per_block = np.sqrt(np.mean((got - want).reshape(blocks, -1) ** 2, axis=1))
    return {"voices_sampled": int(len(sel)), "blocks": blocks, "timeline_blocks": f"0..{blocks - 1}",
            "normalisation": "bus / voices sampled",
            "kernel_form": forms, "kernel_form_forced_to_match_timed_region": bool(forced.force), "kernel_form_forced_to": forced.form,
            "bus_rms_err": rms, "bus_rms_err_worst_block": float(per_block.max()), "worst_block": int(per_block.argmax()),
            "signal_rms": float(np.sqrt(np.mean(want ** 2))), "signal_peak": peak,
            # an effect chain with gain (config #3: chorus taps + recirculating combs) lifts signal and error alike
            "bus_rms_err_rel_full_scale": rms / max(1.0, peak),
            "max_abs_err": float(np.max(np.abs(got - want)))}


def usable_cpus(hardware_concurrency):
    """How many CPUs this process can actually keep busy: its affinity mask and its cgroup's CPU quota (cpu.max: the GPU boxes of this
    pool show 256 hardware threads and allow 16 CPUs of time — measured, round 6: the f64 oracle scales 7.0x on 8 threads, 11.3x on 16,
    and then flat at the quota however many threads run).  Returns (threads to use, why)."""
    n, why = hardware_concurrency, "std::thread::hardware_concurrency()"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, why = aff, f"the process's affinity mask ({aff} of {hardware_concurrency} hardware threads)"
    except Exception:  # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                q = max(1, int(math.ceil(float(quota) / float(period))))
                if q < n:
                    n, why = q, f"the cgroup's CPU quota ({path}: {quota} / {period} = {q} CPUs of {hardware_concurrency} hardware threads)"
        except Exception:  # noqa: BLE001
            pass
    try:   # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            q = max(1, -(-quota // period))
            if q < n:
                n, why = q, f"the cgroup's CPU quota (cfs_quota_us / cfs_period_us = {q} CPUs of {hardware_concurrency} hardware threads)"
    except Exception:  # noqa: BLE001
        pass
    return n, why


def cpu_baseline(workload, seconds_target=15.0):
    """The f64 scalar oracle ("port"; the reference Rust path cannot be built here) timed on this host,
    rank 0 only, on a bounded sample of the same workload: mode A single thread, mode B all host cores."""
    from oracle import oracle as O
    from oracle.projects import OracleProject
    V = WORKLOADS[workload]["voices"]
    sample_voices = min(V, 1024)
    sel = spread_sample(V, sample_voices)
    sample_voices = len(sel)
    op = OracleProject(workload, sel)
    t0 = time.perf_counter()
    op.step()
    dt = max(time.perf_counter() - t0, 1e-6)
    blocks = int(max(4, min(4096, seconds_target / dt)))
    t0 = time.perf_counter()
    for _ in range(blocks):
        op.step()
    el = time.perf_counter() - t0
    vf_per_s = sample_voices * FRAMES * blocks / el
    out = {
        "value": vf_per_s / V, "unit": "stereo frames/s", "cores": 1, "kind": "port",
        "sample": f"{sample_voices} of {V} voices (spread over the project, odd stride) x {blocks} blocks of {FRAMES} frames from block 1 of the "
                  f"timeline, f64 scalar oracle -O2, 1 thread; frames/s scaled by {sample_voices}/{V} "
                  f"(measured {vf_per_s:.3e} voice-frames/s)",
    }
    # second figure: the same scalar port compiled -O3 -march=native ON THIS HOST (the .so is
    # rebuilt here: one built in another container may use instructions this CPU lacks), same sample, single thread
    try:
        import subprocess as sp
        odir = os.path.join(REPO, "oracle")
        sp.run(["make", "-s", "-B", "-C", odir, "native"], check=True, timeout=120)
        Ln = O.lib(native=True)
        opn = OracleProject(workload, sel, lib_=Ln)
        opn.step()
        nb = max(4, blocks // 3)
        t0 = time.perf_counter()
        for _ in range(nb):
            opn.step()
        eln = time.perf_counter() - t0
        out["native"] = {"value": sample_voices * FRAMES * nb / eln / V, "cores": 1, "flags": "-O3 -march=native (built on this host)",
                         "sample": f"{sample_voices} voices x {nb} blocks"}
    except Exception as e:  # noqa: BLE001
        out["native"] = {"error": str(e)[:200]}
    hw = int(O.lib().oracle_hardware_concurrency()) or 1
    cores, why_cores = usable_cpus(hw)
    if WORKLOADS[workload]["kind"] in ("welsh", "mixed", "sampler"):  # mode B: voices sharded over all host cores
        # PERSISTENT workers (round 6): every thread owns >= 64 voices for >= 16 consecutive blocks between one spawn and one join
        # (oracle_bank_render_bus_blocks_mt; the stretch ends where the timeline has note events).  Round 5 spawned and joined 256 threads
        # for every block of 64 voices each: that figure (10.5 x one core on 256 cores) timed the threads, not the voices.
        mt_voices = min(V, max(64 * cores, 4096))
        mp_ = OracleProject(workload, spread_sample(V, mt_voices))
        mt_voices = mp_.n
        mp_.run_mt(1, cores)                                 # block 0: the note-ons, first touch of every page by its thread
        budget_s = 0.3 * seconds_target
        mt_blocks = int(max(16, min(80, budget_s * vf_per_s * max(1.0, 0.5 * min(cores, 64)) / (mt_voices * FRAMES))))
        t0 = time.perf_counter()
        _, stretches = mp_.run_mt(mt_blocks, cores)
        el = time.perf_counter() - t0
        mt_vf = mt_voices * FRAMES * mt_blocks / el
        eff = mt_vf / (cores * vf_per_s)
        out["all_cores"] = {"value": mt_vf / V, "cores": cores, "hardware_concurrency": hw, "cores_is": why_cores,
                            "speedup_over_one_core": mt_vf / vf_per_s, "parallel_efficiency": eff,
                            "sample": f"{mt_voices} voices ({mt_voices // max(cores, 1)} per thread) x {mt_blocks} blocks from block 1 of the timeline, {cores} persistent "
                                      f"threads, {stretches} spawn / join stretch(es)",
                            "note": ("hardware_concurrency counts SMT siblings: two threads of a core share its FP units, so ~0.5 of the thread count is what a "
                                     "scalar f64 loop can gain; a box's other tenants and NUMA placement take the rest") if eff < 0.5 else None}
    return out


# ------------------------------------------------------------------------------------------ timing
class Dist:
    """The ranks' rendezvous and the bus reduce.  The data-path collective is the library's own RCCL communicator (one
    ncclReduce of the bus on the ctx stream, groove_bus_reduce); the launcher-side process group only carries the
    128-byte unique id, the barriers and the max of the ranks' clocks, so it runs over gloo on the host and torch
    never opens a GPU context of its own in a rank: its streams and a second RCCL instance would share the device's
    few hardware queues with the render's per-kind streams (measured on one GPU: 0.58 -> 0.69 ms per block).  If the
    library communicator cannot be set up the rank leaves with a non-zero code: there is no second collective backend."""

    def __init__(self, rank, world, local_rank):
        # torch is imported (and the host-side group formed) BEFORE libgroove_hip.so is loaded: a process has one HIP
        # runtime and one RCCL, whichever library asks first, and the wheel's bundled pair only works as a pair
        # (libgroove_hip.so first, then torch: ncclCommInitRank fails with "unhandled cuda error").
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world, self.local_rank = torch, dist, rank, world, local_rank
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        self.ctx = None

    def make_context(self):
        """The rank's library context and its RCCL communicator (one ncclReduce of the bus per render).  ONE collective path: a rank
        whose communicator cannot be set up says so and ends with a non-zero code (every rank learns it through the gloo group and
        leaves the same way) — there is no second backend to fall back on, so a line that exists measured groove_bus_reduce
        (round 6; the torch.distributed-over-nccl fallback of rounds 3 - 5 is gone).  The supervisor (supervise_rank) starts a fresh
        child; a rank that has touched the GPU is never re-exec'ed."""
        torch, dist, rank, world, local_rank = self.torch, self.dist, self.rank, self.world, self.local_rank
        self.reduce_via = "groove_bus_reduce (RCCL ncclReduce on the ctx stream; communicator created " + ("before" if os.environ.get("GROOVE_COMM_BEFORE_STREAMS") == "1" else "after") + " the library's streams)"
        why = None
        uid = [None]
        try:
            uid = [E.Context.new_comm_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            why = f"ncclGetUniqueId failed: {e}"
        dist.broadcast_object_list(uid, src=0)
        ctx = None
        if uid[0] is None:
            why = why or "rank 0 could not create the communicator's id"
        else:
            try:
                if os.environ.get("GROOVE_BENCH_BREAK_COMM") == "1":  # exercise the refusal below (tests/test_projects_cpu.py)
                    raise RuntimeError("GROOVE_BENCH_BREAK_COMM=1")
                # The library's streams FIRST, the communicator after them.  (groove_init_comm — the communicator before the
                # streams, so that RCCL's own streams cannot land between the library's — was round 3's first answer to the
                # queue-mapping question of DESIGN.md section 7, and measured it costs every rank a fifth of its speed: 1,000,000
                # voices on one GPU through this path 0.630 / 0.635 / 0.630 ms per block against 0.517 / 0.522 / 0.517 in this
                # order and 0.534 / 0.533 / 0.531 with no communicator at all, profiles/r03_ab_logs_second_half.txt (r3_dist_ab).
                # GROOVE_COMM_BEFORE_STREAMS=1 selects it.)
                if os.environ.get("GROOVE_COMM_BEFORE_STREAMS") == "1":
                    ctx = E.Context(local_rank, comm=(uid[0], rank, world))
                else:
                    ctx = E.Context(local_rank)
                    ctx.comm_init(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001
                why = f"the library communicator could not be set up: {e}"
        leave_unless_every_rank_has_its_communicator(dist, torch, rank, why, ctx)   # every rank takes the same way out
        self.ctx = ctx
        self.rccl_ranks = ctx.comm_ranks()
        return ctx

    def sync(self):
        self.ctx.synchronize()      # every stream of the library on this device
        self.dist.barrier()
        self.ctx.synchronize()

    def reduce_bus(self, bus, frame0, frames):
        self.ctx.bus_reduce(E._Slice(bus, frame0), frames, 0)

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def max_over_ranks(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def time_project(ctx, proj, bus, K, W, repeats, span_mode, dist=None):
    """`repeats` x (reset, W warm-up steps, K timed steps bracketed by a synchronisation — and a barrier
    across ranks — on both sides).  Returns per-repeat wall seconds (max over ranks) and HIP-event ms per step."""
    walls, kerns, own = [], [], []
    for _ in range(repeats):
        proj.reset()
        for s in range(W):
            proj.step(bus, s * FRAMES)
        dist.sync() if dist else ctx.synchronize()
        # Where the event pair brackets the whole (pipelined) step, consecutive pairs would tile the timed
        # region: only its two ends are recorded (every record is a packet on the ctx stream, ~5 us of its
        # timeline each) and the average step is their span divided by the steps.
        if span_mode:
            first_ev, last_ev = ctx.event(), ctx.event()
            pairs = [(first_ev if s == 0 else None, last_ev if s == K - 1 else None) for s in range(K)]
        else:
            pairs = [(ctx.event(), ctx.event()) for _ in range(K)]
        t0 = time.perf_counter()
        for s in range(K):
            proj.step(bus, (W + s) * FRAMES, pairs[s])
        if dist:
            dist.reduce_bus(bus, W * FRAMES, K * FRAMES)
            dist.sync()
        else:
            ctx.synchronize()
        el = time.perf_counter() - t0
        own.append(el)
        walls.append(dist.max_over_ranks(el) if dist else el)
        if span_mode:
            kerns.append(ctx.elapsed_ms(pairs[0][0], pairs[-1][1]) / K)
        else:
            kerns.append(float(np.mean([ctx.elapsed_ms(a, b) for a, b in pairs])))
        for a, b in pairs:
            for e in (a, b):
                if e is not None:
                    ctx.L.groove_event_destroy(ctx.h, e)
    extra = {"own_walls": own}
    if dist:  # outside every timed region: what the ONE collective of a render costs by itself (barrier, reduce, sync; max over ranks)
        alone = []
        scratch = ctx.bus(K * FRAMES)  # (a bus of the same size: the timed region's own bus keeps what the render left in it)
        for _ in range(3):
            dist.sync()
            t0 = time.perf_counter()
            dist.reduce_bus(scratch, 0, K * FRAMES)
            dist.sync()
            alone.append(dist.max_over_ranks(time.perf_counter() - t0))
        scratch.destroy()
        extra["reduce_alone_ms"] = sorted(alone)[1] * 1e3
        extra["walls_by_rank"] = dist.gather(own)  # [rank][repeat]
    return walls, kerns, extra


def roofline_block(workload, n_local, kern_ms, span_mode, fused, window=None, source_hash=None):
    """Two views of one step.  EFFECTIVE (`achieved` / `frac`): the entity-boundary algorithmic bytes of
    the step divided by its duration — what an implementation that materialised every voice block would have to move;
    a fused kernel does not move them and the figure can exceed 1.  PHYSICAL: what the PMC passes of the committed
    profile counted for the same command — HBM bytes (`traffic`, `hbm_physical_frac`) and VALU wave-instructions
    (`valu.achieved_frac` of the spec issue rate; `valu.frac_of_measured_bound` against the issue time of the same class
    mix measured by tools/micro/mix_bound, a separate labelled figure).  `bound` names the larger of the two physical
    fractions (both on spec peaks), or "latency" when both are under 0.25 (a step of a few short, dependent launches)."""
    wl = WORKLOADS[workload]
    whole = span_mode
    dom_bytes = wl["bytes_per_vf"] if whole else wl["dominant_bytes"]
    achieved = dom_bytes * n_local * FRAMES / (kern_ms * 1e-3) / 1e9
    prof = committed_profile(workload, window) or {}
    kernel = ("welsh_render_uniform_mix_kernel<fused> (three launches per block, a third of every kind's workgroups each; class-specialised block bodies; "
              "welsh_render_uniform_kernel per exact-f64 kind) + partial_rows/final" if fused and wl["kind"] == "welsh"
              else "whole step: render of block b+1 + the chain's IIR head (side stream) beside the rest of the effect chain + mix of block b" if whole and wl["kind"] == "chain"
              else "whole step: the banks' fused render kernels side by side + their bus reductions" if whole
              else "render kernel of the first bank")
    r = {"bound": None,
         "kernel": kernel,
         # SURVEY §8d's entity-boundary bytes ÷ time: an EFFECTIVE rate (the fused kernels never move these
         # bytes; it can exceed the HBM peak).  `traffic` is what the PMC pass saw actually move.
         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "frac_is": "effective (algorithmic entity-boundary bytes / time, SURVEY §8d); the physical fractions are hbm_physical_frac and valu.achieved_frac",
         "effective": {"note": "algorithmic (entity-boundary) bytes / time, SURVEY §8d; not a physical bandwidth",
                       "GBs": achieved, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS},
         "algorithmic_bytes_per_voice_frame": dom_bytes, "algorithmic_bytes_per_step": dom_bytes * n_local * FRAMES,
         "kernel_ms": kern_ms,
         "traffic": None, "traffic_unit": "HBM bytes per step (PMC FETCH_SIZE x2 + WRITE_SIZE, committed pass)",
         "traffic_source": prof.get("source"), "traffic_same_window": prof.get("same_window_as_this_run"),
         "profile_window_steps_warmup": prof.get("window_steps_warmup"),
         # which code the committed counters were taken with (tools/summarize_prof.py writes the library's identity into every summary
         # since round 6) against the library that is running: the physical fields below describe THIS code only when the two agree
         "traffic_source_head": (prof.get("library") or {}).get("git_head"), "traffic_source_hash": (prof.get("library") or {}).get("source_hash"),
         "running_source_hash": source_hash,
         "stale_profile": (None if not prof else True if not (prof.get("library") or {}).get("source_hash") or not source_hash
                           else (prof["library"]["source_hash"] != source_hash))}
    scale = n_local / WORKLOADS[workload]["voices"]   # the committed pass counted the workload's full-size step
    hbm_frac = valu_frac = None
    if prof.get("traffic"):
        r["traffic"] = prof["traffic"] * scale
        hbm_frac = r["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        r["hbm_physical_frac"] = hbm_frac
        r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes_per_step"] if r["algorithmic_bytes_per_step"] else None
    if prof.get("valu_per_step"):
        wi = prof["valu_per_step"] * scale
        valu_frac = wi / (kern_ms * 1e-3) / VALU_ISSUE_PER_S
        r["valu"] = {"wave_insts_per_step": wi, "salu_wave_insts_per_step": (prof.get("salu_per_step") or 0.0) * scale,
                     "valu_per_voice_frame": wi * 64.0 / (n_local * FRAMES) if n_local else None,
                     "spec_issue_rate": VALU_ISSUE_PER_S, "spec_issue_rate_unit": "wave64 VALU instructions/s (1,024 SIMD-32 x 2.4 GHz / 2 clk)",
                     "achieved_frac": valu_frac,
                     # the clock the render kernels held in the committed GRBM counter pass, where they run one at a time
                     # (GRBM_GUI_ACTIVE / 8 XCDs / dispatch time; MI355X_MICROARCH.md "DVFS give-back"): the spec clock holds
                     "clock_ghz_in_counter_pass": prof.get("clock_ghz"),
                     "source": prof.get("source"), "same_window": prof.get("same_window_as_this_run")}
        if prof.get("valu_simd_ns_per_step"):
            # the measured instruction classes at their measured issue costs (tools/summarize_prof.py, docs/VALU_COSTS.md): the SIMD
            # time the step's VALU instructions need, against the SIMD time the step had (1,024 SIMDs x its duration)
            cw = prof["valu_simd_ns_per_step"]
            have = 1024.0 * kern_ms * 1e6
            lo, hi = cw["other_all_fast"] * scale / have, cw["other_all_normal"] * scale / have
            r["valu"]["cost_weighted_frac"] = {"low": lo, "high": hi,
                                               "note": "SIMD time of the step's VALU instructions at their measured issue costs / SIMD time available; "
                                                       "low / high: the unclassified instructions (moves, compares, selects, DPP) all fast / all normal"}
            r["valu"]["mix_shares"] = (prof.get("valu_mix_per_step") or {}).get("shares")
        mb = measured_mix_bound()
        if mb and wl["kind"] == "welsh" and WORKLOADS[workload]["voices"] >= 500_000:
            # The issue bound of THIS instruction stream, measured (tools/micro/mix_bound.hip: the window's class mix issued as
            # independent chains at the render kernels' occupancies): the time below which no schedule of the step's VALU
            # instructions can finish, and the step's duration against it, priced at the render kernels' occupancy (five waves per
            # SIMD: kernels.h WavesBudget).  A separate, labelled field: `valu.achieved_frac` stays on the spec issue rate.
            # (since round 5 all four class-specialised per-kind kernels are budgeted for FIVE waves per SIMD — kernels.h WavesBudget; until
            # then the two LFO_F64_SMOOTH kernels ran at four and their share of the step's instructions was priced at that rate)
            share4 = 0.0
            ns = mb["ns_at_5_waves"] * (1.0 - share4) + mb["ns_at_4_waves"] * share4
            bound_ms = wi * ns * 1e-9 / 1024.0 * 1e3
            fmb = bound_ms / kern_ms
            r["valu"]["measured_bound"] = {"ns_per_wave_instruction": {"5_waves_per_simd": mb["ns_at_5_waves"], "4_waves_per_simd": mb["ns_at_4_waves"], "weighted": ns},
                                           "share_of_instructions_at_4_waves": share4, "bound_ms": bound_ms, "source": mb["source"],
                                           "instruction_count_source": prof.get("source"), "same_window": prof.get("same_window_as_this_run")}
            r["valu"]["frac_of_measured_bound"] = fmb
            r["valu"]["bound_source"] = mb["source"]
            flags = []
            if fmb > 1.0:
                flags.append("above 1: the step beat the synthetic stream's issue time (instruction count from another window, or clocks moved between the two measurements)")
            if not prof.get("same_window_as_this_run"):
                flags.append("instruction count taken from a profile of a different window")
            if flags:
                r["valu"]["frac_of_measured_bound_flags"] = flags
    fr = {"hbm": hbm_frac or 0.0, "valu-issue": valu_frac or 0.0}
    top = max(fr, key=fr.get)
    r["bound"] = (top if fr[top] >= 0.25 else "latency (one or two short launches per block: neither HBM nor VALU issue is a quarter busy)") if (hbm_frac is not None or valu_frac is not None) \
        else ("valu-issue" if wl["kind"] in ("welsh", "mixed") else "hbm")
    r["physical"] = {"hbm_frac": hbm_frac, "valu_frac": valu_frac}
    return r


SHORT_WINDOW_REPEATS = 7  # timed regions of a short window (see main)
FORM_REPEATS = 5          # ... of the materialised million-voice forms (secondary entries)
PHASE = {"now": "start"}  # what the measurement was doing, for the message of a stall


def phase(text):
    PHASE["now"] = text


def bench_workload(ctx, workload, sel, K, W, repeats, fused=True, grouped=True, render_ahead=True, dist=None, head_ahead=True, paced=None):
    phase(f"timing {workload}, {len(sel)} voices, fused={fused}, grouped={grouped}, blocks {W}..{W + K - 1} x {repeats}")
    """Build the shard `sel` of a workload, time it, return the measurements (no parity, no JSON)."""
    wl = WORKLOADS[workload]
    proj = PJ.Project(ctx, workload, sel, fused=fused, grouped=grouped, render_ahead=render_ahead, head_ahead=head_ahead, paced=paced)
    forms = sorted({inst.kernel_form(FRAMES, fused and not fx) for inst, _, fx, _ in proj.banks})
    walk = ("paced: renders two blocks ahead, the host waits for the events itself, the bus reduction deferred into the next chain launch"
            if proj.paced else "render-ahead (one block, device-side waits)" if proj.ahead_walk
            else "fused render + mix on the library's streams, paced by the host (groove_bank_render_mix_paced)" if proj.paced_fused
            else "banks in turn on the ctx stream (groove_bank_render_mix_deferred)" if proj.take_turns else "block by block")
    bus = ctx.bus((K + W) * FRAMES)
    span_mode = (fused and wl["kind"] != "chain") or (render_ahead and (wl["kind"] == "chain" or not fused))
    walls, kerns, extra = time_project(ctx, proj, bus, K, W, repeats, span_mode, dist)
    out_bus = bus.download()[W * FRAMES:] if (dist is None or dist.rank == 0) else None
    proj.destroy()
    bus.destroy()
    order = np.argsort(walls)
    med = int(order[len(order) // 2])
    return {"walls": walls, "kerns": kerns, "median": med, "span_mode": span_mode, "bus": out_bus, "kernel_form": forms, "walk": walk, **extra}


def section_plan(world, rank, workload="welsh-1m", voices=0):
    """The three measurements ONE `bench.py --gpus N` run makes (N > 1), as voice ranges of this rank — shared by the real run and
    by --dry-launch (tests/test_projects_cpu.py):
      strong        the workload's voices split N ways (contiguous index ranges): the line's `value`
      weak          the workload's voice count on EVERY rank: the project grows with N
      mixed-131072  config #5 as BASELINE.json writes it: 131,072 mixed voices split N ways (16,384 per GPU at N = 8)."""
    V = voices or WORKLOADS[workload]["voices"]
    Vm = WORKLOADS["mixed-131072"]["voices"]
    return {"strong": {"workload": workload, "voices_total": V, "range": list(voice_range(V, rank, world))},
            "weak": {"workload": workload, "voices_total": V * world, "range": list(voice_range(V * world, rank, world))},
            "mixed-131072": {"workload": "mixed-131072", "voices_total": Vm, "range": list(voice_range(Vm, rank, world))}}


def section_summary(m, K, plan, dist, scaling):
    """One entry of the line's `sections` (rank 0): the whole job's frames/s, the ranks' own clocks, the communicator, the reduce."""
    i = m["median"]
    per_rank = [w[i] / K * 1e3 for w in m.get("walls_by_rank") or [m["own_walls"]]]
    fps = K * FRAMES / m["walls"][i]
    return {"workload": plan["workload"], "scaling": scaling, "voices_total": plan["voices_total"], "voices_per_gpu": plan["range"][1] - plan["range"][0],
            "value": fps, "unit": "stereo frames/s", "x_realtime_44k1": fps / SR, "voice_frames_per_s": fps * plan["voices_total"],
            "ms_per_step": m["walls"][i] / K * 1e3, "ms_per_step_repeats": [w / K * 1e3 for w in m["walls"]],
            "ms_per_step_by_rank": {"max": max(per_rank), "min": min(per_rank), "all": per_rank,
                                    "note": "each rank's own clock over the same region (the region ends with the bus reduce and a barrier, so the ranks' clocks differ by what they waited at its start)"},
            "kernel_ms_rank0": m["kerns"][i], "rccl_ranks": dist.rccl_ranks, "bus_reduce": dist.reduce_via,
            "bus_reduce_alone_ms": m.get("reduce_alone_ms"), "kernel_form": m["kernel_form"]}


def config_entry(ctx, workload, repeats, parity_voices=64):
    """One sub-entry of `configs`: the workload's whole project timeline timed on this GPU + sampled parity."""
    wl = WORKLOADS[workload]
    V, K = wl["voices"], wl["blocks"]
    m = bench_workload(ctx, workload, np.arange(V, dtype=np.int64), K, 0, repeats)
    ms = [w / K * 1e3 for w in m["walls"]]
    i = m["median"]
    fps = K * FRAMES / m["walls"][i]
    roof = roofline_block(workload, V, m["kerns"][i], m["span_mode"], True, source_hash=ctx.debug_info().get("source_hash"))
    par = sampled_parity(ctx, workload, V, min(K, 172), sample=parity_voices)
    return {"workload": workload, "voices": V, "blocks_timed": f"0..{K - 1} (the whole project, {K * FRAMES} frames)",
            "ms_per_step": ms[i], "ms_per_step_min": min(ms), "ms_per_step_repeats": ms,
            "value": fps, "unit": "stereo frames/s", "x_realtime_44k1": fps / SR,
            "kernel_ms": m["kerns"][i], "kernel_form": m["kernel_form"], "walk": m["walk"],
            "frac": roof["frac"], "frac_is": "effective (SURVEY §8d algorithmic bytes / time)", "bound": roof["bound"],
            "hbm_physical_frac": roof.get("hbm_physical_frac"), "valu_achieved_frac": (roof.get("valu") or {}).get("achieved_frac"),
            "algorithmic_bytes_per_voice_frame": roof["algorithmic_bytes_per_voice_frame"],
            "traffic": roof.get("traffic"), "traffic_source": roof.get("traffic_source"), "valu": roof.get("valu"),
            "parity_vs_oracle": par,
            "library_counters_after": {k: v for k, v in ctx.debug_info().items() if k in ("host_waits", "host_waits_blocked", "host_wait_ms", "zero_segments", "fast_table_misses")}}


def form_entry(ctx, label, K, W, fused, grouped, note):
    """welsh-1m in one of the entity-boundary forms (voice blocks written to HBM, then mixed), timed over the same
    window as the headline: what a host that calls Generates::generate_batch_values and then gather_audio gets
    (/root/reference/entities/src/instruments/metronome.rs:23-35, orchestration/src/orchestrator.rs:397-410)."""
    V = WORKLOADS["welsh-1m"]["voices"]
    m = bench_workload(ctx, "welsh-1m", np.arange(V, dtype=np.int64), K, W, FORM_REPEATS, fused=fused, grouped=grouped)
    wall = m["walls"][m["median"]]
    ms = wall / K * 1e3
    byts = WORKLOADS["welsh-1m"]["bytes_per_vf"] * V * FRAMES
    fps = K * FRAMES / wall
    return {"workload": label, "voices": V, "blocks_timed": f"{W}..{W + K - 1}", "note": note + "; renders submitted with groove_bank_render_async, three blocks in rotation",
            "kernel_form": m["kernel_form"], "ms_per_step_repeats": [w / K * 1e3 for w in m["walls"]], "statistic": "median region",
            "ms_per_step": ms, "value": fps, "unit": "stereo frames/s", "x_realtime_44k1": fps / SR,
            "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_is": "algorithmic 18 B per voice-frame / time; in this form 10 of the 18 bytes are really moved (8 B of block written + 2 B of state in and out; the mix reads the render's row sums, not the block)",
            "hbm_physical_frac": 10.0 * V * FRAMES / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def window_entry(ctx, workload, K, W, ms_headline, note):
    """A million-voice workload other than the headline's over the headline's window (blocks W..W+K-1, fused render + mix): timed,
    priced against the same algorithmic bytes, with its own sampled parity.  `welsh-1m-library`: the 106-slot table whose class
    proportions follow the reference's patch library (groove_amd/patches.py LIBRARY_*)."""
    wl = WORKLOADS[workload]
    V = wl["voices"]
    m = bench_workload(ctx, workload, np.arange(V, dtype=np.int64), K, W, SHORT_WINDOW_REPEATS)
    i = m["median"]
    ms = m["walls"][i] / K * 1e3
    fps = K * FRAMES / m["walls"][i]
    byts = wl["bytes_per_vf"] * V * FRAMES
    return {"workload": workload, "voices": V, "blocks_timed": f"{W}..{W + K - 1}", "note": note, "kernel_form": m["kernel_form"], "walk": m["walk"],
            "ms_per_step": ms, "ms_per_step_repeats": [w / K * 1e3 for w in m["walls"]], "statistic": "median region",
            "kernel_ms": m["kerns"][i], "value": fps, "unit": "stereo frames/s", "x_realtime_44k1": fps / SR,
            "vs_headline_step": ms / ms_headline if ms_headline else None,
            "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_is": "effective (SURVEY §8d algorithmic bytes / time)",
            "parity_vs_oracle": sampled_parity(ctx, workload, V, min(K + W, 48), sample=212)}


def shard_curve(ctx, K, W, ms_full):
    """What strong scaling can reach, measured on this one GPU: the per-GPU shard of welsh-1m on 1 / 2 / 4 / 8 GPUs (V, V/2,
    V/4, V/8 voices: the first rank's contiguous range) over the same window, and the 16,384-voice shard of
    mixed-131072 (config #5's per-GPU share) over its whole timeline.  efficiency = t(V) / (N x t(V / N)): the bus reduce
    (352 KB once per render) is not in it."""
    V = WORKLOADS["welsh-1m"]["voices"]
    rows = [{"gpus": 1, "voices_per_gpu": V, "ms_per_step": ms_full, "implied_speedup": 1.0, "implied_efficiency": 1.0}]
    for n in (2, 4, 8):
        lo, hi = voice_range(V, 0, n)
        m = bench_workload(ctx, "welsh-1m", np.arange(lo, hi, dtype=np.int64), K, W, SHORT_WINDOW_REPEATS)
        ms = m["walls"][m["median"]] / K * 1e3
        rows.append({"gpus": n, "voices_per_gpu": hi - lo, "ms_per_step": ms, "ms_per_step_repeats": [w / K * 1e3 for w in m["walls"]],
                     "kernel_form": m["kernel_form"], "implied_speedup": ms_full / ms, "implied_efficiency": ms_full / ms / n})
    out = {"welsh-1m": rows, "window_blocks": f"{W}..{W + K - 1}", "repeats": SHORT_WINDOW_REPEATS, "statistic": "median region"}
    Vm, Km = WORKLOADS["mixed-131072"]["voices"], WORKLOADS["mixed-131072"]["blocks"]
    lo, hi = voice_range(Vm, 0, 8)
    m = bench_workload(ctx, "mixed-131072", np.arange(lo, hi, dtype=np.int64), Km, 0, 3)  # (three small banks on three streams: the noisiest entry of the line, so three regions and their median)
    out["mixed-131072_shard_of_8"] = {"voices_per_gpu": hi - lo, "ms_per_step": m["walls"][m["median"]] / Km * 1e3,
                                      "ms_per_step_repeats": [w / Km * 1e3 for w in m["walls"]], "kernel_form": m["kernel_form"],
                                      "blocks_timed": f"0..{Km - 1}", "note": "compare with configs[mixed-131072].ms_per_step (the whole project on one GPU)"}
    return out


def run_under_watchdog(argv, args, attempts=3):
    """Run this script's measurement in a child process; kill and restart a child that exceeds its time or whose library
    reported a stalled stream (exit code 75).  The parent never touches the GPU.  Prints the child's JSON line with
    `watchdog` added; returns the exit code."""
    limit = args.watchdog_seconds or (240.0 if args.steps <= 50 else 420.0)  # a healthy run takes 40-70 s / 100-140 s
    env = dict(os.environ, GROOVE_BENCH_CHILD="1")
    if not os.environ.get("GROOVE_BENCH_FAKE_STALL_ONCE"):
        mix_bound_env(env)   # this box's issue bound, measured by a fresh child BEFORE the measurement child (this process never touches the GPU)
    killed = 0
    for attempt in range(1, attempts + 1):
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=subprocess.PIPE, text=True)
        try:
            out, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            killed += 1
            print(f"[bench] attempt {attempt}: no result after {limit:.0f} s, child {p.pid} killed", file=sys.stderr, flush=True)
            time.sleep(2.0)
            continue
        if p.returncode == RANK_STALL_EXIT:
            killed += 1
            print(f"[bench] attempt {attempt}: the library reported a stalled stream (child {p.pid} exited {RANK_STALL_EXIT}); starting again", file=sys.stderr, flush=True)
            time.sleep(2.0)
            continue
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            try:
                print(add_watchdog(lines[-1], {"attempts": attempt, "killed": killed, "seconds_allowed": limit,
                                               "tainted": killed > 0}),  # a restarted run: quote it as such
                      flush=True)
            except Exception:
                print(lines[-1], flush=True)
            return 0
        sys.stdout.write(out)
        return p.returncode or 1
    print(json.dumps({"error": f"bench: {attempts} attempts exceeded {limit:.0f} s each", "watchdog": {"attempts": attempts, "killed": killed}}), flush=True)
    return 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=172, help="default: one full 172-block project (44,032 frames)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions, each from a reset state (value = the median); default: 7 for windows of up to 40 steps, 3 otherwise")
    ap.add_argument("--workload", default="welsh-1m", choices=sorted(WORKLOADS))
    ap.add_argument("--voices", type=int, default=0, help="override the workload's total voice count")
    ap.add_argument("--materialise", action="store_true",
                    help="entity-boundary form: write every voice block to HBM, then run the separate mix kernels "
                         "(default: fused render+mix, no materialised voice blocks)")
    ap.add_argument("--interleaved", action="store_true", help="voice i uses patch i mod 32 inside every wavefront (the library regroups or runs the per-lane kernel)")
    ap.add_argument("--weak", action="store_true",
                    help="multi-GPU: every rank holds the workload's voice count and the project grows with --gpus "
                         "(default: strong scaling, the workload's voices are split over the ranks)")
    ap.add_argument("--strong", action="store_true", help="(the default; kept for explicit command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` entries (the other four workloads and the entity-boundary forms)")
    ap.add_argument("--no-shard-curve", action="store_true", help="skip `shard_curve` (the per-GPU shards of 2 / 4 / 8 GPUs timed on this one)")
    ap.add_argument("--no-parity", action="store_true", help="skip the sampled oracle comparison")
    ap.add_argument("--no-render-ahead", action="store_true",
                    help="workloads with effect chains: render block b, then its effects (default: the render of block b+1 "
                         "is submitted to the side streams before the effects of block b)")
    ap.add_argument("--no-pacing", action="store_true", help="round 3's walks: device-side event waits instead of the host waiting for the events itself (effect chains: renders one block ahead and a reduction launch per block; fused projects: groove_bank_render_mix) (A/B)")
    ap.add_argument("--no-head-ahead", action="store_true", help="render-ahead walk: keep the chain's leading IIR stage on the ctx stream (A/B)")
    ap.add_argument("--head-unfused", action="store_true", help="render-ahead walk: the IIR head behind the render as its own launch, not fused into the render kernel (A/B)")
    ap.add_argument("--dry-launch", action="store_true", help="rendezvous of the ranks over gloo only (no GPU): launcher test")
    ap.add_argument("--no-sections", action="store_true", help="N > 1: only the measurement the flags ask for (default: strong, weak and mixed-131072 from one run)")
    ap.add_argument("--no-watchdog", action="store_true", help="run the measurement in this process (default on one GPU: in a child process that is killed and restarted if it crawls)")
    ap.add_argument("--watchdog-seconds", type=float, default=0.0, help="time allowed per attempt (default: 240 s for up to 50 steps, 420 s otherwise)")
    args = ap.parse_args()

    is_child = os.environ.get("GROOVE_BENCH_CHILD") == "1"
    supervise = not args.no_watchdog and not under_profiler()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not under a launcher: start the N ranks ourselves, before anything initialises a GPU in this process.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: fewer (or more) ranks than GPUs asked for")
    # test hooks of the watchdogs (tests/test_projects_cpu.py): no GPU involved
    fake = os.environ.get("GROOVE_BENCH_FAKE_STALL_ONCE")
    if fake and is_child and (world == 1 or rank == world - 1):
        if not os.path.exists(fake):
            open(fake, "w").close()
            if os.environ.get("GROOVE_BENCH_FAKE_STALL_EXIT") == "1":
                sys.exit(RANK_STALL_EXIT)
            time.sleep(3600)
    if world > 1 and not is_child and supervise:
        # Under an external launcher (torch.distributed.run): this process supervises, a child of it is the rank.
        sys.exit(supervise_rank(sys.argv[1:], rank, world))
    if args.dry_launch:
        sys.exit(dry_launch(world, rank))
    if fake and is_child and world == 1:
        # (GROOVE_BENCH_FAKE_LINE: a whole measurement as a real child would have assembled it — tests/test_bench_line.py feeds
        # round 4's 20 KB line through emit() this way)
        src = os.environ.get("GROOVE_BENCH_FAKE_LINE")
        if src:
            emit(json.load(open(src)))
        else:
            print(json.dumps({"metric": "fake", "value": 1.0}), flush=True)
        sys.exit(0)

    # Watchdog (one GPU, not under a launcher).  Rounds 2 and 3 saw the million-voice path stall in some processes; round 4
    # found the cause (an endless loop in one workgroup of a render kernel) and removed it, and the
    # line's `zero_segments` counts its trigger.  The watchdog stays as what it always was — a harness that cannot hang:
    # the measurement runs in a child process; a child that has not finished in time — or whose library reported a
    # stalled stream itself — is killed (its exact PID) and the run starts again, and the line says so (`tainted`).
    # Never from under a profiler.
    if world == 1 and "WORLD_SIZE" not in os.environ and supervise and not is_child:
        sys.exit(run_under_watchdog(sys.argv[1:], args))
    from groove_amd.lib import GrooveError
    try:
        measure(args, world, rank, local_rank)
    except GrooveError as e:
        if "not complete after" in str(e):   # groove_synchronize's deadline: a stalled stream.  The ctx cannot be torn down
            sys.stderr.write(f"bench.py rank {rank}: {e}\n[bench] the stall was seen while: {PHASE['now']}\n")  # (hipFree would wait for the stalled kernel): leave at once
            sys.stderr.flush()
            os._exit(RANK_STALL_EXIT)
        raise


def measure(args, world, rank, local_rank):
    use_dist = world > 1 or os.environ.get("GROOVE_BENCH_FORCE_DIST") == "1"  # the latter: exercise the N>1 code path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    # a stalled stream comes back as an error after this long (the library's default is 60 s; a healthy sync here takes < 1 s)
    os.environ.setdefault("GROOVE_SYNC_TIMEOUT_MS", "30000" if not under_profiler() else "0")

    wl = dict(WORKLOADS[args.workload])
    V = args.voices or wl["voices"]
    weak = args.weak and not args.strong
    V_total = V * world if weak else V   # voices of the whole project
    lo, hi = voice_range(V_total, rank, world)
    dist = Dist(rank, world, local_rank) if use_dist else None
    ctx = dist.make_context() if dist is not None else E.Context(0)
    if dist is not None and dist.rccl_ranks != world:
        sys.stderr.write(f"bench.py: the communicator has {dist.rccl_ranks} ranks, {world} GPUs were asked for\n")
        ctx.close()
        sys.exit(3)

    fused = not args.materialise
    # A short window is 10 - 15 ms of GPU work and the first regions of a process (or after the seconds the host spends in the
    # oracle between phases) run 5 - 12 % slower than the ones after them — 1,000,000 voices, nine regions of 20 steps:
    # 0.548 0.534 0.514 0.488 0.483 0.486 0.486 0.485 0.486 ms per block (profiles/r03_timed_regions.log): three regions put the
    # median on the slope, seven put it on what the device sustains.  Every region is on the line.
    K, W = args.steps, args.warmup
    R = max(1, args.repeats) if args.repeats else (SHORT_WINDOW_REPEATS if K <= 40 else 3)
    sel = np.arange(lo, hi, dtype=np.int64)
    m = bench_workload(ctx, args.workload, sel, K, W, R, fused=fused, grouped=not args.interleaved,
                       render_ahead=not args.no_render_ahead, dist=dist,
                       head_ahead=False if args.no_head_ahead else ("unfused" if args.head_unfused else True),
                       paced=False if args.no_pacing else None)
    line = None
    if rank == 0:
        i = m["median"]
        elapsed, kern_ms = m["walls"][i], m["kerns"][i]
        out_bus = m["bus"]
        frames_total = K * FRAMES
        project_fps = frames_total / elapsed   # frames of the (merged) project per second: the metric, under either scaling
        n_local = hi - lo
        period = wl["blocks"]
        line = {
            "metric": "stereo frames/sec rendered (offline)", "value": project_fps, "unit": "stereo frames/s",
            "x_realtime_44k1": project_fps / SR, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32 (f64 IIR state, u64 phase)", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {V_total} voices total, {FRAMES}-frame blocks, {SR} Hz, "
                                   f"{'fused render+mix' if fused else 'materialised blocks + mix kernels'}",
                       "voices_total": V_total, "voices_per_gpu": n_local,
                       "kernel_form": m["kernel_form"], "walk": m["walk"],
                       "bus_reduce": (dist.reduce_via if dist else "none (one rank)"),
                       "parallelism": (f"voices sharded x{world} ("
                                       + (f"weak: {V} voices per GPU, the project grows with N; value = the merged project's frames/s, voice_frames_per_s scales"
                                          if weak else "strong: the fixed project split N ways")
                                       + "), no data-path collective, 1 RCCL bus reduce per render")},
            "timed_region": {"repeats": R, "statistic": "median repeat (by wall time)",
                             "ms_per_step_repeats": [w / K * 1e3 for w in m["walls"]],
                             "ms_per_step_min": min(m["walls"]) / K * 1e3,
                             "kernel_ms_repeats": m["kerns"],
                             "timeline_blocks": f"{W}..{W + K - 1} of the {period}-block project timeline (looped), each repeat from a reset state "
                                                f"(note-on at block 0, note-off at block {PJ.NOTE_OFF_BLOCK})"},
            "project_frames_per_s": project_fps,
            "voice_frames_per_s": project_fps * V_total,
            "path_effective_GBs": wl["bytes_per_vf"] * project_fps * V_total / 1e9,
            "roofline": roofline_block(args.workload, n_local, kern_ms, m["span_mode"], fused, window=(K, W), source_hash=ctx.debug_info().get("source_hash")),
            "output_check": {"finite": bool(np.isfinite(out_bus).all()), "peak_abs_bus_over_V": float(np.abs(out_bus).max() / V_total)},
            "streams": ctx.debug_info(),
        }
        if dist is not None:
            line["rccl_ranks"] = dist.rccl_ranks
        line["zero_segments"] = line["streams"].get("zero_segments")
        line["library"] = library_identity(line["streams"])
        if line["zero_segments"]:
            line["tainted"] = "the Welsh kernels counted zero-frame segments (csrc/diag.h; DESIGN.md section 7): this must not happen"
        if line["streams"].get("fast_table_misses"):
            line["tainted"] = "waves in a FAST body found a look-ahead table down (csrc/diag.h fast_table_misses): this must not happen"
    if dist is not None and not args.no_sections and not args.materialise and not args.interleaved:
        # ONE run of the driver's command carries all three N-GPU measurements (section_plan): the line's own measurement is one of
        # them, the other scaling mode of the same workload and config #5 follow with the same ranks and the same communicator
        plans = section_plan(world, rank, args.workload, args.voices)
        mine = "weak" if weak else "strong"
        sections = {}
        if rank == 0:
            sections[mine] = section_summary(m, K, plans[mine], dist, mine)
        other = "strong" if weak else "weak"
        lo2, hi2 = plans[other]["range"]
        m2 = bench_workload(ctx, args.workload, np.arange(lo2, hi2, dtype=np.int64), K, W, R, dist=dist)
        if rank == 0:
            sections[other] = section_summary(m2, K, plans[other], dist, other)
        if args.workload != "mixed-131072":
            lo3, hi3 = plans["mixed-131072"]["range"]
            Km = WORKLOADS["mixed-131072"]["blocks"]
            m3 = bench_workload(ctx, "mixed-131072", np.arange(lo3, hi3, dtype=np.int64), Km, 0, 3, dist=dist)
            if rank == 0:
                sections["mixed-131072"] = section_summary(m3, Km, plans["mixed-131072"], dist, "strong")
                sections["mixed-131072"]["blocks_timed"] = f"0..{Km - 1} (the whole project)"
        if rank == 0:
            line["sections"] = sections
            line["sections_note"] = ("one run, the same ranks and communicator: `strong` = the workload split N ways (the line's value unless --weak), "
                                     "`weak` = the workload's voice count on every rank, `mixed-131072` = config #5 split N ways")
    if dist is not None:
        dist.dist.barrier()
        dist.dist.destroy_process_group()
    if rank == 0 and world == 1:
        default_line = args.workload == "welsh-1m" and not args.voices and fused and not args.interleaved
        if not args.no_parity:
            line["parity_vs_oracle"] = sampled_parity(ctx, args.workload, V_total, min(K + W, 48), sample=256 if default_line else 128, fused=fused,
                                                      grouped=not args.interleaved)
        if default_line and not args.no_configs:
            line["configs"] = [config_entry(ctx, w, R) for w in ("welsh-256", "chain-4096", "sampler-16384", "mixed-131072")]
            Kf, Wf = min(K, 20), min(W, 5)
            line["configs"].append(window_entry(ctx, "welsh-1m-library", Kf, Wf, line["ms_per_step"] if (K, W) == (Kf, Wf) else None,
                                                "106 synthetic patches in the class proportions of the reference's patch library "
                                                "(profiles/r06_library_proportions.json): square / sawtooth LFOs on the pitch, ripples up to 10.7 under sweeps, "
                                                "a noise LFO on the pitch and the resonance routing (the exact-f64 kind: 1.9 % of the voices)"))
            line["configs"].append(form_entry(ctx, "welsh-1m-materialised", Kf, Wf, False, True,
                                              "entity-boundary form: every voice block written to HBM (2 GB per block), mixed from the render's row sums"))
            line["configs"].append(form_entry(ctx, "welsh-1m-interleaved-materialised", Kf, Wf, False, False,
                                              "the same with patch i mod 32 on voice i (no two neighbouring voices share a patch): regrouped inside the library"))
        if default_line and not args.no_shard_curve:
            if (K, W) == (min(K, 20), min(W, 5)) and R == SHORT_WINDOW_REPEATS:
                ms_full = line["ms_per_step"]
            else:  # (the same statistic as the shards': the median of the window's regions)
                mf = bench_workload(ctx, "welsh-1m", sel, min(K, 20), min(W, 5), SHORT_WINDOW_REPEATS)
                ms_full = mf["walls"][mf["median"]] / min(K, 20) * 1e3
            line["shard_curve"] = shard_curve(ctx, min(K, 20), min(W, 5), ms_full)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
    ctx.close()
    if rank == 0:
        # The JSON line goes out last: RCCL prints a version banner through C stdio, which would otherwise
        # be flushed at exit, after this line.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit(line)


if __name__ == "__main__":
    main()
