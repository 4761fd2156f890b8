set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/gputests3.log; cat gpurun_out/r4/gputests3.log
L=gpurun_out/r4/chain_ab.log; : > $L
B="python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog"
one() { echo -n "$1: " >> $L; env $2 timeout 300 $B $3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), [round(x,5) for x in d['timed_region']['ms_per_step_repeats']], d['config']['walk'][:40], d['zero_segments'])" >> $L 2>&1; }
for rep in 1 2 3; do
  one "paced bind" "GROOVE_BIND_EVENTS=1" ""
  one "paced nobind" "GROOVE_BIND_EVENTS=0" ""
  one "unpaced bind" "GROOVE_BIND_EVENTS=1" "--no-pacing"
  one "unpaced nobind (round 3)" "GROOVE_BIND_EVENTS=0" "--no-pacing"
done
cat $L
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/kt_chain_paced -- $B --repeats 1 > gpurun_out/r4/kt_chain_paced.log 2>&1
# million-voice line with the events bound again
for rep in 1 2; do
  for be in 1 0; do echo -n "welsh-1m window BIND=$be: " >> $L; GROOVE_BIND_EVENTS=$be timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])" >> $L; done
done
tail -4 $L
