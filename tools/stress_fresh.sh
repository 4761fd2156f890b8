#!/bin/bash
# Liveness gate for the multi-stream million-voice path: N FRESH processes, each renders 1,000,000
# Welsh voices for a few blocks through the per-kind pipelined kernels (groove_amd/canary.py) under its own timeout.
#   tools/stress_fresh.sh [processes = 20] [timeout seconds = 60] [extra environment, e.g. GROOVE_SAFE_STREAMS=1]
# Output: gpurun_out/stress_fresh.log — one line per process ("ok <ms>", "STALL (library deadline)" or "TIMEOUT").
set -u
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-20}; TO=${2:-60}; shift 2 2>/dev/null || true
mkdir -p gpurun_out
LOG=gpurun_out/stress_fresh.log
echo "# $N fresh processes, timeout $TO s, env: $*" >> $LOG
ok=0; bad=0
for i in $(seq 1 $N); do
  out=$(env GROOVE_NO_CANARY=1 GROOVE_SYNC_TIMEOUT_MS=20000 "$@" timeout $TO python3 -m groove_amd.canary 2>&1 | tail -1)
  rc=$?
  if echo "$out" | grep -q "^canary: "; then ok=$((ok + 1)); echo "process $i: ok  $out" >> $LOG
  elif echo "$out" | grep -q "not complete after"; then bad=$((bad + 1)); echo "process $i: STALL (library deadline)  $out" >> $LOG
  else bad=$((bad + 1)); echo "process $i: TIMEOUT or failure  $out" >> $LOG; fi
done
echo "stress_fresh: $ok ok, $bad stalled or failed of $N" | tee -a $LOG
