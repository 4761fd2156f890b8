#!/bin/bash
# Liveness AND integrity gate for the multi-stream million-voice path: N FRESH processes, each renders
# 1,000,000 Welsh voices for eight blocks through the per-kind pipelined kernels (tools/fresh_render.py) under its own timeout
# and prints the CRC-32 of the bus it rendered and the library's zero-segment counter.
#   tools/stress_fresh.sh [processes = 20] [timeout seconds = 60] [extra environment, e.g. GROOVE_SAFE_STREAMS=1]
# Output: gpurun_out/stress_fresh.log — one line per process, then the verdict: how many completed, how many DISTINCT CRCs
# (must be 1: with the segment guard in place a corrupted run would complete, so liveness alone proves nothing) and the sum of
# the zero-segment counters (must be 0).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-20}; TO=${2:-60}; shift 2 2>/dev/null || true
mkdir -p gpurun_out
LOG=gpurun_out/stress_fresh.log
echo "# $N fresh processes, timeout $TO s, env: $*" >> $LOG
ok=0; bad=0
for i in $(seq 1 $N); do
  out=$(env GROOVE_SYNC_TIMEOUT_MS=20000 "$@" timeout $TO python3 tools/fresh_render.py 2>&1 | tail -1)
  if echo "$out" | grep -q "^fresh: "; then ok=$((ok + 1)); echo "process $i: ok  $out" >> $LOG
  elif echo "$out" | grep -q "not complete after"; then bad=$((bad + 1)); echo "process $i: STALL (library deadline)  $out" >> $LOG
  else bad=$((bad + 1)); echo "process $i: TIMEOUT or failure  $out" >> $LOG; fi
done
crcs=$(grep "^process .*: ok" $LOG | tail -$ok | sed 's/.* crc \([0-9a-f]*\) .*/\1/' | sort -u | wc -l)
zeros=$(grep "^process .*: ok" $LOG | tail -$ok | sed 's/.*zero_segments \([0-9]*\).*/\1/' | awk '{s += $1} END {print s + 0}')
echo "stress_fresh: $ok ok, $bad stalled or failed of $N; distinct bus CRCs $crcs; zero segments counted $zeros" | tee -a $LOG
