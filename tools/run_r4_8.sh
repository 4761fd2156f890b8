set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 600 python3 -m pytest tests/test_gpu_independent.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r4/gputests8.log; cat gpurun_out/r4/gputests8.log
L=gpurun_out/r4/ring_nt_ab.log; : > $L
B="python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog"
for rep in 1 2 3; do
  for lib in libgroove_hip.so libvar_ring_nt.so; do echo -n "chain-4096 $lib: " >> $L; GROOVE_LIB_PATH=$PWD/groove_amd/$lib timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), [round(x,5) for x in d['timed_region']['ms_per_step_repeats']])" >> $L 2>&1; done
done
cat $L
GROOVE_LIB_PATH=$PWD/groove_amd/libvar_ring_nt.so timeout 600 python3 -m pytest tests/test_gpu_fx.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
