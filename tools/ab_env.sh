#!/bin/bash
# A/B two ENVIRONMENT settings of one build of libgroove_hip.so inside ONE gpurun job (boxes differ by a few per cent in wall
# time, so only in-job comparisons count):
#   tools/ab_env.sh "<bench args>" "GROOVE_LOOK_AHEAD=1" "GROOVE_LOOK_AHEAD=3" [...]
# Alternates the settings REPS times and prints frames/s per run.  (tools/ab_bench.sh does the same over builds.)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ARGS="$1"; shift
REPS=${REPS:-3}
for rep in $(seq $REPS); do
  for setting in "$@"; do
    v=$(env $setting timeout ${RUN_TIMEOUT:-180} python3 bench.py $ARGS --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.0f} frames/s  {d['ms_per_step']:.4f} ms/step\")")
    echo "$setting: $v"
  done
done
