cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 1500 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_split.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for m in 0 1; do for v in 20000 65536 125000 250000 500000; do
GROOVE_DEFER_BUS=$m timeout 200 $B --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('voices $v defer=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done; done
} 2>&1 | tee gpurun_out/r3_defer_serial_ab.log
