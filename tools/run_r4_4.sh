set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4/gputests4.log; cat gpurun_out/r4/gputests4.log
L=gpurun_out/r4/chain_slack.log; : > $L
B="python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog"
one() { echo -n "$1: " >> $L; env $2 timeout 300 $B $3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), [round(x,5) for x in d['timed_region']['ms_per_step_repeats']], d['zero_segments'])" >> $L 2>&1; }
for rep in 1 2; do
  for s in 0 1 2 3 5; do one "paced slack $s" "GROOVE_PACED_SLACK=$s" ""; done
  one "unpaced" "X=1" "--no-pacing"
done
cat $L
# store wave A/B on the materialised million-voice form
M=gpurun_out/r4/store_wave_ab.log; : > $M
BM="python3 bench.py --steps 20 --warmup 5 --materialise --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 5"
for rep in 1 2 3; do
  for lib in libgroove_hip.so libvar_store_wave.so; do echo -n "$lib: " >> $M; GROOVE_LIB_PATH=$PWD/groove_amd/$lib timeout 300 $BM 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['zero_segments'])" >> $M 2>&1; done
done
cat $M
GROOVE_LIB_PATH=$PWD/groove_amd/libvar_store_wave.so timeout 600 python3 -m pytest tests/test_gpu_welsh.py tests/test_gpu_split.py tests/test_gpu_fullsize.py tests/test_gpu_async.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4/gputests4_sw.log; cat gpurun_out/r4/gputests4_sw.log
