#!/usr/bin/env python3
"""Reduce the rocprofv3 CSVs of tools/profile_round.sh to profiles/<round>_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, rnd = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
summary = {"round": rnd, "command": "python3 bench.py --no-cpu-baseline   (defaults: --gpus 1 --steps 172 --warmup 4, workload welsh-1m)"}

ks = glob.glob(f"{out}/kt/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks[0], f"profiles/{rnd}_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in rows[:8]]
# The fused step is several kernels running concurrently (and, pipelined, overlapping the next
# step's), so no single kernel's average duration is "the step": the step period is read off the
# trace as the spacing of the per-step bus reductions (partial_final_kernel ends once per step).
kt = glob.glob(f"{out}/kt/*/*_kernel_trace.csv")
if kt:
    ends = sorted(int(r["End_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "partial_final_kernel" in r["Kernel_Name"])
    if len(ends) > 20:
        steady = ends[4:]  # skip the warm-up steps
        summary["step_period_from_trace"] = {
            "mean_us": (steady[-1] - steady[0]) / (len(steady) - 1) / 1e3, "steps": len(steady) - 1,
            "note": "spacing of partial_final_kernel completions over the timed steps; compare with roofline.kernel_ms of the bench line"}
for log in glob.glob(f"{out}/bench_kt.log"):
    for line in open(log):
        if line.startswith("{"):
            summary["bench_line_under_profiler"] = json.loads(line)


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-70:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"),
                       "scratch": r.get("Scratch_Size"), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")}
    return {k: {"mean_per_dispatch": {c: sum(v) / len(v) for c, v in cs.items()}, "dispatches": max(len(v) for v in cs.values()),
                **meta[k]} for k, cs in agg.items() if "rocclr" not in k}


for sub in ("fetch", "write", "sq", "grbm"):
    summary[sub] = counters(sub)

# HBM traffic per STEP (one 256-frame block of the whole project): the Welsh render runs as up to
# six concurrent kernels (one per base kind) plus the two partial-row reductions, so the
# per-kernel means are summed.  MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
# FETCH_SIZE reads 1/2 of the bytes of a wide coalesced streaming read on gfx950 — the state loads
# here are 4 B/lane buffer loads (an uncalibrated width), so the raw and the doubled figure are kept.
def per_step(sub, counter):
    tot = 0.0
    for k, v in summary.get(sub, {}).items():
        if "welsh_render" in k or "partial_" in k or "mix_" in k:
            tot += v["mean_per_dispatch"].get(counter, 0.0)
    return tot * 1024.0


w, f = per_step("write", "WRITE_SIZE"), per_step("fetch", "FETCH_SIZE")
summary["dominant_kernel"] = "welsh_render_uniform_kernel<fused, LFO mode, retune> (one per base kind, concurrent, blocks pipelined) + partial_rows/final"
summary["hbm_traffic_bytes_per_step"] = {"write": w, "fetch_raw": f, "fetch_x2_gfx950": 2 * f,
                                         "total_raw": w + f, "total_corrected": w + 2 * f}
json.dump(summary, open(f"profiles/{rnd}_summary.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("dominant_kernel", "hbm_traffic_bytes_per_step", "step_period_from_trace") if k in summary}, indent=1))
for r in summary.get("kernel_stats", []):
    print(f"{r['pct']:6.2f}%  {r['avg_ns'] / 1e3:10.1f} us x {r['calls']:4d}  {r['name']}")
