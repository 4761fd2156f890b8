cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/ktc2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktc2 -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 > gpurun_out/ktc2.log 2>&1
python3 tools/timeline.py gpurun_out/ktc2 0.6 40 | tee gpurun_out/tl_chain2.txt
