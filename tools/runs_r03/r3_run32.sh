cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in paced safe; do
rm -rf gpurun_out/ktc_$v
if [ $v = safe ]; then export GROOVE_SAFE_STREAMS=1; fi
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktc_$v -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 > gpurun_out/ktc_$v.log 2>&1
echo "== $v"; tail -1 gpurun_out/ktc_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
python3 tools/timeline.py gpurun_out/ktc_$v 0.6 12
done 2>&1 | tee gpurun_out/r3_tl_paced.log
