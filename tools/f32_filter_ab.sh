#!/bin/bash
# The fp32 filter kind on / off (GROOVE_F32_FILTER) in ONE gpurun job, alternating: the driver's window of the million-voice project.
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do
  for f in 1 0; do
    v=$(GROOVE_F32_FILTER=$f timeout 240 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-shard-curve --no-watchdog ${EXTRA_ARGS:-} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.4f} ms/step  parity_rms {d.get('parity_rms')}\")")
    echo "GROOVE_F32_FILTER=$f: $v"
  done
done
