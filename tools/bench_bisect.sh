#!/bin/bash
# config #3 inside whole bench.py runs (fresh processes), the commands exactly as the driver / a user gives them
cd "${GRAFT_REPO_ROOT:-.}"
run() { python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('[$*]', d['ms_per_step'], [(c['workload'], c['ms_per_step']) for c in d.get('configs',[]) if c['workload'] in ('chain-4096','welsh-256')])"; }
for i in 1 2 3; do
run
run --gpus 1 --steps 20 --warmup 5
run --no-cpu-baseline
done
