#!/bin/bash
# SQ counters of the Welsh render kernels for several builds of the library, one rocprofv3 --pmc pass each (in ONE job):
#   tools/pmc_ab.sh "<bench args>" groove_amd/libvar_A.so groove_amd/libvar_B.so ...
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ARGS="$1"; shift
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
for lib in "$@"; do
  name=$(basename $lib .so)
  cp "$lib" groove_amd/libgroove_hip.so
  rm -rf gpurun_out/pmc_$name
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py $ARGS --no-cpu-baseline --no-configs --no-parity --no-shard-curve --repeats 1 --no-watchdog > gpurun_out/pmc_$name.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcg_$name -- python3 bench.py $ARGS --no-cpu-baseline --no-configs --no-parity --no-shard-curve --repeats 1 --no-watchdog > gpurun_out/pmcg_$name.log 2>&1
  python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc_{name}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "welsh_render" not in k: continue
        k = k.split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
tot = collections.defaultdict(float)
for k, c in sorted(agg.items()):
    d = cnt[k]
    print(f"{name} {k}: dispatches {d} valu/wave-frame {c['SQ_INSTS_VALU']/c['SQ_WAVES']/256:.1f} salu {c['SQ_INSTS_SALU']/c['SQ_WAVES']/256:.1f} "
          f"active_valu(quad)/wave-frame {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVES']/256:.1f} wave_cycles(quad)/wave-frame {c['SQ_WAVE_CYCLES']/c['SQ_WAVES']/256:.1f} "
          f"wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f} wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f}")
    for x in c: tot[x] += c[x] / d
print(f"{name} per step: VALU insts {tot['SQ_INSTS_VALU']:.4g} active_valu quad-cycles {tot['SQ_ACTIVE_INST_VALU']:.4g} busy {tot['SQ_BUSY_CYCLES']:.4g}")
g = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcg_{name}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "welsh_render" in r["Kernel_Name"]: g[r["Kernel_Name"].split("(")[0][-48:]].append(float(r["Counter_Value"]))
for f in glob.glob(f"gpurun_out/pmcg_{name}/*/*_kernel_trace.csv"):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "welsh_render" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0][-48:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(dur):
        gm = sum(g[k]) / max(1, len(g[k])); dm = sum(dur[k]) / len(dur[k])
        print(f"{name} {k}: mean duration {dm/1e3:.1f} us, GRBM_GUI_ACTIVE {gm:.4g} -> clock ~{gm/8/dm:.2f} GHz")
import json
for line in open(f"gpurun_out/pmc_{name}.log"):
    if line.startswith("{"): print(name, "ms/step under profiler", json.loads(line)["ms_per_step"])
PY
done
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
