cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_configs.py tests/test_gpu_time_parallel.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for m in 0 1; do
for w in "--workload welsh-256" "--voices 512" "--voices 1024" "--voices 2048" "--workload sampler-16384"; do
GROOVE_DEFER_BUS=$m timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w defer=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done; done
} 2>&1 | tee gpurun_out/r3_defer512_ab.log
