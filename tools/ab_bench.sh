#!/bin/bash
# A/B two builds of libgroove_hip.so inside ONE gpurun job (MI355X boxes differ by ~5 % in
# wall time, so only in-job comparisons count):
#   tools/ab_bench.sh "<bench args>" groove_amd/libvar_A.so groove_amd/libvar_B.so [...]
# Alternates the variants REPS times and prints frames/s per run.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ARGS="$1"; shift
REPS=${REPS:-3}
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
for rep in $(seq $REPS); do
  for lib in "$@"; do
    src="$lib"; [ "$(basename $lib)" = "libgroove_hip.so" ] && src=/tmp/base_lib.so   # the in-tree build itself: its saved copy
    cp "$src" groove_amd/libgroove_hip.so
    v=$(timeout ${RUN_TIMEOUT:-180} python3 bench.py $ARGS --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['value']:.0f} frames/s  {d['ms_per_step']:.4f} ms/step\")")
    echo "$(basename $lib): $v"
  done
done
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
