#!/bin/bash
# config #3 (chain-4096) with the reverb's all-passes on the library's all-pass stream (the paced walk's default) and behind the
# run on the ctx stream, with 0 - 2 extra blocks in the rotation, alternately in one job:  tools/ap_stream_ab.sh
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do
  for cfg in "0 0" "1 0" "1 1" "1 2" "0 1"; do
    set -- $cfg
    v=$(GROOVE_PROJECT_ALLPASS_STREAM=$1 GROOVE_PROJECT_PACED_SLACK=$2 timeout 200 python3 bench.py --workload chain-4096 --no-cpu-baseline --no-configs --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.4f} ms/step  parity {d.get('parity_rms')}\")")
    echo "all-pass stream $1, slack $2: $v"
  done
done
