cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_gpu_time_parallel.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for rep in 1 2; do for m in 0 3073; do
for w in "--workload chain-4096" "--voices 4096" "--voices 8192" "--voices 16384"; do
GROOVE_TP_VPW2_MIN_VOICES=$m timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w vpw2_min=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['config']['kernel_form'])"
done; done; done
} 2>&1 | tee gpurun_out/r3_vpw2_ab.log
