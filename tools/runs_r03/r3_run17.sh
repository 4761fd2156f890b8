cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
run() { TAG="$*"; env "$@" timeout 200 $B $ARGS 2>/dev/null | tail -1 | TAG="$TAG" LABEL="$LABEL" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['LABEL'], os.environ['TAG'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; }
for v in 32768 65536 100000 125000; do
  ARGS="--steps 20 --warmup 5 --voices $v"; LABEL="voices=$v"
  run GROOVE_SPLIT_MAX_WAVES=0
  run GROOVE_SPLIT_MAX_WAVES=4096 GROOVE_SPLIT_ROLES=3
  run GROOVE_SPLIT_MAX_WAVES=4096 GROOVE_SPLIT_ROLES=2
done 2>&1 | tee gpurun_out/r3_split2_ab.log
ARGS="--workload mixed-131072"; LABEL="mixed"
run GROOVE_SPLIT_ROLES=3 2>&1 | tee -a gpurun_out/r3_split2_ab.log
run GROOVE_SPLIT_ROLES=2 2>&1 | tee -a gpurun_out/r3_split2_ab.log
