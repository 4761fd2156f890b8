cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for rep in 1 2; do for m in 0 16384; do
for w in "--workload mixed-131072 --voices 16384" "--workload mixed-131072 --voices 4096" "--workload mixed-131072 --voices 8192"; do
GROOVE_TAKE_TURNS_MAX_VOICES=$m timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w take_turns_max=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done; done; done
} 2>&1 | tee gpurun_out/r3_turns_ab.log
