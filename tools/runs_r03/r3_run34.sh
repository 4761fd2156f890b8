cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so /tmp/base.so; cp groove_amd/libvar_unsafe.so groove_amd/libgroove_hip.so
export GROOVE_UNSAFE_NO_FREE_WAIT=1
python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('unsafe no-wait', d['ms_per_step'], d['timed_region']['ms_per_step_repeats'])"
rm -rf gpurun_out/ktc_unsafe
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktc_unsafe -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 > gpurun_out/ktc_unsafe.log 2>&1
python3 tools/timeline.py gpurun_out/ktc_unsafe 0.6 16
cp /tmp/base.so groove_amd/libgroove_hip.so
