cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_time_parallel.py tests/test_gpu_async.py tests/test_gpu_fx.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for rep in 1 2; do for m in 0 1; do
for w in "--workload chain-4096" "--materialise --steps 20 --warmup 5"; do
GROOVE_BIND_FREE_EVENTS=$m timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w bind=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done; done; done
} 2>&1 | tee gpurun_out/r3_bind_ab.log
