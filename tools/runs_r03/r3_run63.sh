# how the timed regions of one process compare (the first is the slowest, the third the fastest: what settles?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-shard-curve --no-parity"
for rep in 1 2; do
timeout 300 $B --repeats 9 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('welsh-1m 9 regions', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done 2>&1 | tee gpurun_out/r3_regions.log
