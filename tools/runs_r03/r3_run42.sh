cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
{
for v in 18432 20480 22528 24576; do for m in 16384 65536; do
GROOVE_TP_MAX_VOICES=$m timeout 200 $B --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('voices $v tp_max=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['config']['kernel_form'])"
done; done
for w in "--workload mixed-131072" "--workload mixed-131072 --voices 16384" "--workload sampler-16384"; do
timeout 200 $B $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['config']['kernel_form'])"
done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
} 2>&1 | tee gpurun_out/r3_thresh.log
