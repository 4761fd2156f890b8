#!/bin/bash
# config #3 inside whole bench.py runs (fresh processes): hunting the run whose chain-4096 repeats all read ~0.06 ms
cd "${GRAFT_REPO_ROOT:-.}"
for i in $(seq 1 ${N:-12}); do
python3 bench.py --no-cpu-baseline --no-shard-curve "$@" > /dev/null 2>&1
python3 - <<'P'
import json
d=json.load(open("bench_detail.json"))
for c in d["configs"]:
    if c["workload"] in ("welsh-256","chain-4096"):
        print(c["workload"], [round(x,4) for x in c["ms_per_step_repeats"]], c.get("library_counters_after"))
P
done
