#!/bin/bash
# N fresh processes of tools/micro/stall_repro (each under its own timeout), several stream layouts:
#   tools/micro/stall_repro.sh [processes per layout = 30] [timeout seconds = 25]
# Output: gpurun_out/stall_repro.log — one line per process; "TIMEOUT" marks a process that had to be killed.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-30}; TO=${2:-25}
mkdir -p gpurun_out
LOG=gpurun_out/stall_repro.log
: > $LOG
BIN=tools/micro/stall_repro
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN tools/micro/stall_repro.hip || exit 1
#        blocks kinds placeholder scratch low flat
LAYOUTS=("200 4 0 1 3 0" "200 4 0 0 3 0" "200 3 1 1 3 0" "200 4 0 1 3 1")
for L in "${LAYOUTS[@]}"; do
  for i in $(seq 1 $N); do
    timeout $TO $BIN $L >> $LOG 2>&1
    rc=$?
    [ $rc -ne 0 ] && echo "stall_repro $L: process $i rc=$rc $([ $rc -eq 124 ] && echo TIMEOUT)" >> $LOG
  done
done
echo "layouts x processes: ${#LAYOUTS[@]} x $N; lines: $(wc -l < $LOG); stalls: $(grep -c 'STALL\|TIMEOUT' $LOG)" | tee -a $LOG
