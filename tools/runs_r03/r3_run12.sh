cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for rep in 1 2; do for pad in 0 26000 55000 90000; do GROOVE_TP_PAD_LDS=$pad timeout 200 $B --workload chain-4096 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain-4096 pad=$pad', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done; done 2>&1 | tee gpurun_out/r3_pad_ab.log
