cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -15 | tee gpurun_out/r3_split4_pytest.log
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
run() { label=$1; shift; env "$@" | tail -1 | L="$label" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['L'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; }
{
for rep in 1 2; do
for r in 3 4; do
  run "welsh-65536 roles$r" GROOVE_SPLIT_ROLES=$r timeout 200 $B --workload welsh-1m --voices 65536 2>/dev/null
  run "welsh-32768 roles$r" GROOVE_SPLIT_ROLES=$r timeout 200 $B --workload welsh-1m --voices 32768 2>/dev/null
  run "mixed-131072 roles$r" GROOVE_SPLIT_ROLES=$r timeout 200 $B --workload mixed-131072 2>/dev/null
done
done
} 2>&1 | tee gpurun_out/r3_split4_ab.log
