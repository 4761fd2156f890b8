set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4/gputests5.log; cat gpurun_out/r4/gputests5.log
M=gpurun_out/r4/wg_reverse_ab.log; : > $M
for rep in 1 2; do
  for v in 125000 250000 350000 500000 1000000; do
    for lib in libgroove_hip.so libvar_wg_reverse.so; do
      echo -n "$v voices $lib: " >> $M
      GROOVE_LIB_PATH=$PWD/groove_amd/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --voices $v --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['zero_segments'])" >> $M 2>&1
    done
  done
  for lib in libgroove_hip.so libvar_wg_reverse.so; do
    echo -n "mixed-131072 $lib: " >> $M
    GROOVE_LIB_PATH=$PWD/groove_amd/$lib timeout 300 python3 bench.py --workload mixed-131072 --no-cpu-baseline --no-parity --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])" >> $M 2>&1
  done
done
cat $M
