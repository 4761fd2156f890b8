cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --steps 20 --warmup 5"
for a in "--materialise" "--materialise --no-render-ahead" "--materialise --interleaved" ""; do timeout 300 $B $a 2>/dev/null | tail -1 | A="$a" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('welsh-1m', os.environ['A'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['output_check'])"; done 2>&1 | tee gpurun_out/r3_mat_ahead.log
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_async.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3)
