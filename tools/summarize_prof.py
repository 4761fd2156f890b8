#!/usr/bin/env python3
"""Reduce the rocprofv3 CSVs of tools/profile_round.sh to profiles/<round>_<workload>_{kernel_stats.csv,summary.json}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, rnd, workload = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "welsh-1m")
os.makedirs("profiles", exist_ok=True)
# "<workload>-window" = the same workload under the driver's command line (--steps 20 --warmup 5): tools/profile_round.sh
base_workload, windowed = (workload[:-len("-window")], True) if workload.endswith("-window") else (workload, False)
materialised = base_workload.endswith("-materialised")   # the entity-boundary form: bench.py --materialise
if materialised:
    base_workload = base_workload[:-len("-materialised")]
# REPEATS (environment, as tools/profile_round.sh ran the bench): timed regions per run.  A process's first regions run 5 - 12 %
# slower than what the device then sustains (profiles/r03_timed_regions.log), so DURATIONS are read off the LAST region only;
# counts (bytes, instructions) are the same in every region and are averaged over all of them.
REPEATS = int(os.environ.get("REPEATS", "1"))
summary = {"round": rnd, "workload": workload, "regions_per_run": REPEATS,
           "command": f"python3 bench.py --workload {base_workload}{' --materialise' if materialised else ''} --no-cpu-baseline --no-configs --no-parity --no-shard-curve --repeats {REPEATS} --no-watchdog"
                      + (" --steps 20 --warmup 5   (the driver's window: blocks 5..24 of the timeline)" if windowed else "   (defaults: --gpus 1 --steps 172 --warmup 4)")}
STEP_KERNELS = ("render", "_tp_kernel", "partial_", "mix_", "fx_", "block_", "_events_")   # what one step of the hot path launches

_glob = glob.glob


def newest_run(pattern):
    """gpurun merges a call's files INTO gpurun_out/: a pass directory that was collected twice holds both runs' files (named
    <pid>_*.csv).  Keep the files of the most recently written run only."""
    files = _glob(pattern)
    if not files:
        return files
    by_pid = collections.defaultdict(list)
    for f in files:
        by_pid[os.path.basename(f).split("_", 1)[0]].append(f)
    best = max(by_pid.values(), key=lambda fs: max(os.path.getmtime(f) for f in fs))
    return best


class _G:
    glob = staticmethod(newest_run)


glob = _G
ks = glob.glob(f"{out}/kt/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks[0], f"profiles/{rnd}_{workload}_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_stats"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in rows[:10]]
# A step is several kernels, some concurrent and (pipelined) overlapping the next step's, so no single kernel's
# average duration is "the step": the step period is read off the trace as the spacing of the kernel that ends
# every step (the bus reduction / mix).
kt = glob.glob(f"{out}/kt/*/*_kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt[0])))
    # (a lone time-parallel bank's reduction rides in its next render — groove_bank_render_mix_deferred — so its step ends with
    # the render kernel itself)
    for marker in ("partial_final_kernel", "partial_rows_kernel", "mix_final_kernel", "mix_partial_kernel", "welsh_tp_kernel", "sampler_tp_kernel", "fm_tp_kernel"):
        ends = sorted(int(r["End_Timestamp"]) for r in rows if marker in r["Kernel_Name"])
        if len(ends) > 40:
            n_steps = 25 if windowed else 176
            per = max(1, round(len(ends) / (n_steps * REPEATS)))       # a step may end with several launches of the marker (one per bank)
            ends = ends[per - 1::per]
            ends = ends[-n_steps:]                                     # the last region of the run
            steady = ends[5 if windowed else 4:]                       # its timed steps
            summary["step_period_from_trace"] = {
                "mean_us": (steady[-1] - steady[0]) / (len(steady) - 1) / 1e3, "steps": len(steady) - 1, "marker": marker,
                "note": "spacing of the step-ending kernel's completions over the timed steps; compare with roofline.kernel_ms of the bench line"}
            break
    # durations of every kernel over the LAST region only (its share of the dispatches, by start time)
    by_name = collections.defaultdict(list)
    for r in rows:
        by_name[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    last = []
    for name, ds in by_name.items():
        if not any(m in name for m in STEP_KERNELS):
            continue
        ds.sort()
        n_last = max(1, len(ds) // REPEATS)
        tail = [d for _, d in ds[-n_last:]]
        last.append({"name": name[:110], "calls_in_last_region": len(tail), "avg_ns": sum(tail) / len(tail), "max_ns": max(tail), "avg_ns_all_regions": sum(d for _, d in ds) / len(ds)})
    summary["kernel_durations_last_region"] = sorted(last, key=lambda r: -r["avg_ns"] * r["calls_in_last_region"])[:12]
for log in glob.glob(f"{out}/bench_kt.log"):
    for line in open(log):
        if line.startswith("{"):
            _line = json.loads(line)
            summary["bench_line_under_profiler"] = {k: v for k, v in _line.items() if k in
                                                    ("value", "ms_per_step", "steps", "warmup", "config", "roofline", "timed_region", "zero_segments")}
            # which code the counters describe (round 6): the running library's own source hash (groove_debug_info) and the commit the
            # in-tree binaries were built at (groove_amd/build_id.json), as bench.py put them on the line of this very run
            if _line.get("library"):
                summary["library"] = _line["library"]


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-70:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                agg[k]["_duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            meta[k] = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"),
                       "scratch": r.get("Scratch_Size"), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")}
    return {k: {"mean_per_dispatch": {c: sum(v) / len(v) for c, v in cs.items()}, "dispatches": max(len(v) for c, v in cs.items() if not c.startswith("_")),
                **meta[k]} for k, cs in agg.items() if "rocclr" not in k}


for sub in ("fetch", "write", "sq", "grbm", "mix1", "mix2"):
    summary[sub] = counters(sub)
STEPS = 176.0   # 172 timed + 4 warm-up steps per run, unless the bench line of the run says otherwise
_bl = summary.get("bench_line_under_profiler", {})
if _bl.get("steps"):
    STEPS = float(_bl["steps"] + _bl.get("warmup", 0)) * REPEATS
summary["steps_per_run"] = STEPS
summary["steps"], summary["warmup"] = _bl.get("steps"), _bl.get("warmup")


# Per STEP (one 256-frame block of the whole project): per-kernel totals over the run divided by the steps, summed over
# the kernels a step launches.  MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads 1/2 of
# the bytes of a wide coalesced streaming read on gfx950 — the raw and the doubled figure are both kept.
def per_step(sub, counter):
    tot = 0.0
    for k, v in summary.get(sub, {}).items():
        if any(m in k for m in STEP_KERNELS):
            tot += v["mean_per_dispatch"].get(counter, 0.0) * v["dispatches"] / STEPS
    return tot


w, f = per_step("write", "WRITE_SIZE") * 1024.0, per_step("fetch", "FETCH_SIZE") * 1024.0
summary["hbm_traffic_bytes_per_step"] = {"write": w, "fetch_raw": f, "fetch_x2_gfx950": 2 * f, "total_raw": w + f, "total_corrected": w + 2 * f}
summary["instructions_per_step"] = {"valu_wave_insts": per_step("sq", "SQ_INSTS_VALU"), "salu_wave_insts": per_step("sq", "SQ_INSTS_SALU"),
                                    "note": "SQ_INSTS_VALU / SQ_INSTS_SALU summed over the step's kernels (wave-level instructions)"}
# Measured instruction classes of a step (passes mix1 / mix2; wave-level instruction counts).  "other" = everything the class
# counters do not name (moves, compares, selects, DPP, bit operations, 64-bit moves ...).  Issue costs are the MEASURED ones
# of docs/VALU_COSTS.md (SIMD time per wave64 instruction on this chip, 8 waves per SIMD): fast 1.05 ns (fp32 add / mul /
# fma with VGPR operands, 32-bit integer), normal 1.9 ns (all f64 arithmetic, every conversion, 64-bit integer, compares,
# selects, DPP, fp32 with an SGPR operand), slow 3.4 ns (v_exp / v_rcp).  "other" is a mixture of fast and normal
# instructions: both bounds are kept.
mix_names = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F32",
             "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")
if summary.get("mix1"):
    mix = {}
    for nm in mix_names:
        v = per_step("mix1", nm) or per_step("mix2", nm)
        mix[nm[len("SQ_INSTS_"):].lower()] = v
    total = mix.get("valu") or 0.0
    f64 = mix["valu_add_f64"] + mix["valu_mul_f64"] + mix["valu_fma_f64"]
    f32 = mix["valu_add_f32"] + mix["valu_mul_f32"] + mix["valu_fma_f32"]
    named = f64 + f32 + mix["valu_trans_f32"] + mix["valu_cvt"] + mix["valu_int32"] + mix["valu_int64"]
    other = max(0.0, total - named)
    mix["f64_total"], mix["f32_total"], mix["other"] = f64, f32, other
    FAST, NORMAL, SLOW = 1.05, 1.75, 3.4  # re-measured in round 4 without the s_nop the compiler puts behind every inline-asm statement (profiles/r04_inst_cost.log): normal 1.9 -> 1.75
    base = FAST * (f32 + mix["valu_int32"]) + NORMAL * (f64 + mix["valu_cvt"] + mix["valu_int64"]) + SLOW * mix["valu_trans_f32"]
    mix["cost_weighted_simd_ns"] = {"other_all_fast": base + FAST * other, "other_all_normal": base + NORMAL * other,
                                    "costs_ns": {"fast": FAST, "normal": NORMAL, "slow": SLOW}, "source": "docs/VALU_COSTS.md (measured on this chip)"}
    mix["shares"] = {"f64": f64 / total, "f32_arith": f32 / total, "conversions": mix["valu_cvt"] / total, "transcendental": mix["valu_trans_f32"] / total,
                     "int32": mix["valu_int32"] / total, "int64": mix["valu_int64"] / total, "other": other / total} if total else {}
    mix["note"] = "wave-level instructions per step by class (PMC SQ_INSTS_VALU_*); cost_weighted_simd_ns = the per-class prices of single-instruction loops summed over the mix: it OVER-prices (classes overlap in the pipe) and is not a bound — profiles/rNN_mix_bound.json (tools/micro/mix_bound.hip) is the measured one"
    summary["valu_mix_per_step"] = mix
# The clock the chip held under each kernel (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE is summed over the 8
# XCDs, so clock = GRBM_GUI_ACTIVE / 8 / the dispatch's duration IN THE SAME PASS (a counter pass runs the kernels one
# at a time, so these are a kernel's own cycles and its own time — NOT the clock of the real run, where four render kernels
# share the chip; the quotient reads high on dispatches well under 0.3 ms).
clocks = {}
for k, v in summary.get("grbm", {}).items():
    g, d = v["mean_per_dispatch"].get("GRBM_GUI_ACTIVE"), v["mean_per_dispatch"].get("_duration_ns")
    if g and d:
        clocks[k] = {"avg_us_alone": d / 1e3, "clock_ghz": g / 8.0 / d}
if clocks:
    summary["clock_in_counter_pass"] = {"per_kernel": clocks,
                                   "note": "GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration in the counter pass (kernels run one at a time there); kernels shorter than ~0.3 ms read high"}
    long_ones = [c for c in clocks.values() if c["avg_us_alone"] >= 100.0]
    if long_ones:
        tot = sum(c["avg_us_alone"] for c in long_ones)
        summary["clock_in_counter_pass"]["ghz_weighted_long_kernels"] = sum(c["clock_ghz"] * c["avg_us_alone"] for c in long_ones) / tot
json.dump(summary, open(f"profiles/{rnd}_{workload}_summary.json", "w"), indent=1)
print(workload, json.dumps({k: summary[k] for k in ("hbm_traffic_bytes_per_step", "instructions_per_step", "step_period_from_trace") if k in summary}))
for r in summary.get("kernel_stats", [])[:8]:
    print(f"{r['pct']:6.2f}%  {r['avg_ns'] / 1e3:10.1f} us x {r['calls']:5d}  {r['name'][:90]}")
