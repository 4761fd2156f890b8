#!/bin/bash
# Diagnostic counter passes over the million-voice window (the driver's command line): LDS activity and conflicts, issue activity per
# instruction type, instruction fetch — one rocprofv3 --pmc pass each (never with a trace domain).  Output: gpurun_out/diag_pmc/<pass>/.
# Reduce with: python3 tools/diag_pmc.py gpurun_out/diag_pmc
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/diag_pmc; rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --workload ${WORKLOAD:-welsh-1m} --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity --no-shard-curve --repeats 2 --no-watchdog"
pass() { n=$1; shift; timeout 420 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- $BENCH > $OUT/$n.log 2>&1; }
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
pass issue SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES
pass fetch SQ_IFETCH SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
