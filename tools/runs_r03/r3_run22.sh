cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
run() { # label, env..., -- args
  label=$1; shift
  env "$@" | tail -1 | L="$label" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['L'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
}
{
for rep in 1 2; do
for w in welsh-256 sampler-16384; do
  run "$w plain" GROOVE_PIPE_SLOTS=4 timeout 200 $B --workload $w 2>/dev/null
  run "$w always slots2" GROOVE_PIPELINE_ALWAYS=1 GROOVE_PIPE_SLOTS=2 timeout 200 $B --workload $w 2>/dev/null
  run "$w always slots3" GROOVE_PIPELINE_ALWAYS=1 GROOVE_PIPE_SLOTS=3 timeout 200 $B --workload $w 2>/dev/null
  run "$w always slots4" GROOVE_PIPELINE_ALWAYS=1 GROOVE_PIPE_SLOTS=4 timeout 200 $B --workload $w 2>/dev/null
done
for w in welsh-1m mixed-131072; do
  run "$w slots2" GROOVE_PIPE_SLOTS=2 timeout 200 $B --workload $w 2>/dev/null
  run "$w slots4" GROOVE_PIPE_SLOTS=4 timeout 200 $B --workload $w 2>/dev/null
done
for v in 65536 250000; do
  run "welsh-$v plain" GROOVE_PIPE_SLOTS=4 timeout 200 $B --workload welsh-1m --voices $v 2>/dev/null
  run "welsh-$v always4" GROOVE_PIPELINE_ALWAYS=1 GROOVE_PIPE_SLOTS=4 timeout 200 $B --workload welsh-1m --voices $v 2>/dev/null
done
done
} 2>&1 | tee gpurun_out/r3_slots.log
