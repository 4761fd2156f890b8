cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_time_parallel.py tests/test_gpu_welsh.py tests/test_gpu_async.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for w in "--workload chain-4096" "--workload welsh-256" "--voices 4096" "--voices 8192" "--voices 16384"; do
timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done
for v in 20000 24576 32768 49152; do for m in 16384 65536; do
GROOVE_TP_MAX_VOICES=$m timeout 200 $B --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('voices $v tp_max=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['config']['kernel_form'])"
done; done
} 2>&1 | tee gpurun_out/r3_alias_ab.log
