cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_split.py -q -x -p no:cacheprovider > gpurun_out/r3_pytest17a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest17a.log)
tail -30 gpurun_out/r3_pytest17a.log | grep -E "passed|failed|FAILED|rc=|Error|assert" | head -20
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
run() { env "$@" timeout 200 $B $ARGS 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LABEL', ' '.join('$@'.split()), round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; }
for v in 32768 65536 100000 125000; do
  ARGS="--steps 20 --warmup 5 --voices $v"; LABEL="voices=$v"
  run GROOVE_SPLIT_MAX_WAVES=0
  run GROOVE_SPLIT_MAX_WAVES=4096 GROOVE_SPLIT_ROLES=3
  run GROOVE_SPLIT_MAX_WAVES=4096 GROOVE_SPLIT_ROLES=2
done 2>&1 | tee gpurun_out/r3_split2_ab.log
ARGS="--workload mixed-131072"; LABEL="mixed"
run GROOVE_SPLIT_ROLES=3 2>&1 | tee -a gpurun_out/r3_split2_ab.log
run GROOVE_SPLIT_ROLES=2 2>&1 | tee -a gpurun_out/r3_split2_ab.log
