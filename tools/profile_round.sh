#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01
# 1. --kernel-trace --stats of the default bench command        → profiles/<round>_kernel_stats.csv
# 2. PMC passes (own runs, no tracing domains besides kernel dispatch counters):
#      FETCH_SIZE            → HBM-side read bytes  (gfx950: reports 1/2 of wide streaming reads)
#      WRITE_SIZE            → HBM-side written bytes
#      SQ instruction mix    → VALU / SALU / wave-cycle budget of the dominant kernel
# Everything is written under gpurun_out/ (scratch) and the summaries copied to profiles/.
set -u
ROUND=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT profiles
BENCH="python3 bench.py --no-cpu-baseline"   # the default bench command (172 steps, 4 warm-up)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- $BENCH > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -- $BENCH > $OUT/bench_grbm.log 2>&1
python3 tools/summarize_prof.py $OUT $ROUND
