cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for w in welsh-256 sampler-16384; do for pm in 8600 1; do GROOVE_PIPELINE_MIN_WAVES=$pm timeout 200 $B --workload $w 2>/dev/null | tail -1 | W=$w PM=$pm python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['W'], 'pipeline_min_waves', os.environ['PM'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done; done 2>&1 | tee gpurun_out/r3_small_pipe.log
