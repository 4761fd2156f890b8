#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun), for every bench workload:
#   tools/profile_round.sh r02 [workload ...]        (default: all five)
# Per workload, each in its own run (gpurun refuses --pmc together with other trace domains):
#   1. --kernel-trace --stats                     -> profiles/<round>_<workload>_kernel_stats.csv
#   2. --pmc FETCH_SIZE / WRITE_SIZE / SQ mix / GRBM_GUI_ACTIVE, one pass each
# and tools/summarize_prof.py reduces them to profiles/<round>_<workload>_summary.json.
# Everything is written under gpurun_out/ (scratch, merged back by gpurun); afterwards, in the development container,
#   for w in welsh-1m welsh-256 chain-4096 sampler-16384 mixed-131072; do python3 tools/summarize_prof.py gpurun_out/prof_r02_$w r02 $w; done
# writes the tracked summaries under profiles/ (the GPU box's own profiles/ directory does not travel back).
set -u
ROUND=${1:-r02}; shift || true
WORKLOADS=${@:-welsh-1m welsh-256 chain-4096 sampler-16384 mixed-131072}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p profiles
TO="timeout ${PASS_TIMEOUT:-420}"   # one pass of one workload takes 20-60 s; a box that crawls must not eat the budget
for W in $WORKLOADS; do
  OUT=gpurun_out/prof_${ROUND}_$W
  rm -rf $OUT; mkdir -p $OUT
  # "<workload>-window": the same workload under the DRIVER's command line (--steps 20 --warmup 5), so that the counters
  # describe the blocks the driver's line times (blocks 5..24: every voice sounding)
  WL=${W%-window}; WIN=""; [ "$WL" != "$W" ] && WIN="--steps 20 --warmup 5"
  # "<workload>-materialised[-window]": the entity-boundary form (every voice block written to HBM, then mixed): bench.py --materialise
  MAT=""; case "$WL" in *-materialised) WL=${WL%-materialised}; MAT="--materialise";; esac
  # REPEATS timed regions per run (default 5 for the window, 1 otherwise): durations are summarised from the LAST region, where the
  # device is at speed (tools/summarize_prof.py)
  R=${REPEATS:-1}; [ -n "$WIN" ] && R=${REPEATS:-5}
  BENCH="python3 bench.py --workload $WL $WIN $MAT --no-cpu-baseline --no-configs --no-parity --no-shard-curve --repeats $R --no-watchdog"   # (the watchdog would start a child from under the profiler)
  $TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_kt.log 2>&1
  $TO rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/bench_fetch.log 2>&1
  $TO rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/bench_write.log 2>&1
  $TO rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- $BENCH > $OUT/bench_sq.log 2>&1
  $TO rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $BENCH > $OUT/bench_grbm.log 2>&1
  if [ "${MIX:-1}" = "1" ]; then   # measured instruction classes (f64 / conversions / transcendental / integer / fp32)
    $TO rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT/mix1 -- $BENCH > $OUT/bench_mix1.log 2>&1
    $TO rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/mix2 -- $BENCH > $OUT/bench_mix2.log 2>&1
  fi
  REPEATS=$R python3 tools/summarize_prof.py $OUT $ROUND $W
done
