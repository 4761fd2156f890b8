# RECORD of a round-3 measurement job (kept because docs/HISTORY.md cites its log): the build variants and GROOVE_* knobs it names were
# measured, not kept, and removed in round 4 — the script documents how the numbers were taken; it no longer runs against this tree.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-shard-curve --no-parity"
for rep in 1 2 3; do
timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
GROOVE_BENCH_FORCE_DIST=1 timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d.get('streams',{}).get('comm_before_streams'))"
GROOVE_BENCH_FORCE_DIST=1 GROOVE_COMM_AFTER_STREAMS=1 timeout 300 $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist comm-after', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d.get('streams',{}).get('comm_before_streams'))"
done 2>&1 | tee gpurun_out/r3_dist_ab.log
