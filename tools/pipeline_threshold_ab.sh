#!/bin/bash
# Mid-size Welsh banks: the all-kinds kernel (default below ~550,000 voices) against the per-kind pipelined kernels (forced by
# GROOVE_PIPELINE_MIN_WAVES=1) — which alone carry the fp32 filter kind — in ONE gpurun job, the driver's window.
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for v in 350000 420000 500000; do
    for pm in default 1; do
      if [ $pm = default ]; then unset GROOVE_PIPELINE_MIN_WAVES; else export GROOVE_PIPELINE_MIN_WAVES=$pm; fi
      r=$(timeout 200 python3 bench.py --voices $v --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.4f} {d['config']['kernel_form'][:40]}\")")
      echo "$v voices, pipeline_min_waves $pm: $r"
    done
  done
done
