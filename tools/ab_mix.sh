#!/bin/bash
# In-job A/B of the MIX kernel (kernels.h; GROOVE_MIX_KERNEL=0: one launch per base kind, round 5's form) over the driver's window,
# for the 32-patch benchmark table and the library-proportioned one: two passes each way, interleaved.   tools/ab_mix.sh [out-dir]
OUT=${1:-gpurun_out/ab_mix}; mkdir -p $OUT
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog"
for pass in 1 2; do
  for W in welsh-1m welsh-1m-library; do
    for M in 0 1; do
      GROOVE_MIX_KERNEL=$M timeout 300 $B --workload $W > $OUT/${W}_mix${M}_$pass.json 2> $OUT/${W}_mix${M}_$pass.err
      python3 - <<PY
import json
l=json.loads(open("$OUT/${W}_mix${M}_$pass.json").read().strip().splitlines()[-1])
d=json.load(open("bench_detail.json"))
print("$W mix=$M pass $pass: ms_per_step %.4f  regions %s" % (l["ms_per_step"], ["%.4f"%x for x in d["timed_region"]["ms_per_step_repeats"]]))
PY
    done
  done
done
