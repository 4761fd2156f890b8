#!/bin/bash
# The default stream layout (three priorities, eight streams) against GROOVE_SAFE_STREAMS=1 (one priority, four streams) in ONE job,
# alternating: the driver's window of the million-voice project, then config #5 on one GPU and config #3.
cd "${GRAFT_REPO_ROOT:-.}"
one() { python3 bench.py "$@" --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.4f}\")"; }
for rep in 1 2 3 4; do
  echo "welsh-1m window   default $(one --steps 20 --warmup 5)   safe $(GROOVE_SAFE_STREAMS=1 one --steps 20 --warmup 5)"
done
for rep in 1 2; do
  echo "mixed-131072      default $(one --workload mixed-131072)   safe $(GROOVE_SAFE_STREAMS=1 one --workload mixed-131072)"
  echo "chain-4096        default $(one --workload chain-4096)   safe $(GROOVE_SAFE_STREAMS=1 one --workload chain-4096)"
  echo "welsh-1m project  default $(one)   safe $(GROOVE_SAFE_STREAMS=1 one)"
done
