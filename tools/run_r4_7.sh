set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
M=gpurun_out/r4/materialised_bind_ab.log; : > $M
BM="python3 bench.py --steps 20 --warmup 5 --materialise --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 5"
for rep in 1 2 3; do
  for be in 1 0; do echo -n "materialised BIND=$be: " >> $M; GROOVE_BIND_EVENTS=$be timeout 300 $BM 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['zero_segments'])" >> $M 2>&1; done
  for be in 1 0; do echo -n "interleaved materialised BIND=$be: " >> $M; GROOVE_BIND_EVENTS=$be timeout 300 $BM --interleaved 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['zero_segments'])" >> $M 2>&1; done
done
cat $M
