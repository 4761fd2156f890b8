#!/bin/bash
# config #3 (chain-4096) alone in fresh processes, six per rotation slack (GROOVE_PROJECT_PACED_SLACK): the spread of its step by process
cd "${GRAFT_REPO_ROOT:-.}"
for slack in 2 3 1; do for i in 1 2 3 4 5 6; do
  GROOVE_PROJECT_PACED_SLACK=$slack python3 bench.py --workload chain-4096 --no-cpu-baseline --no-configs --no-shard-curve --no-watchdog --no-parity 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('slack $slack', round(d['ms_per_step'],4))"
done; done
