cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_split.py -q -x -p no:cacheprovider > gpurun_out/r3_pytest8a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest8a.log)
tail -30 gpurun_out/r3_pytest8a.log | grep -E "passed|failed|FAILED|rc=|Error|assert" | head -20
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for sp in 2048 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --workload mixed-131072 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed-131072 split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done 2>&1 | tee gpurun_out/r3_split_ab8.log
for v in 20000 32768 49152 65536 80000; do for sp in 4096 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --steps 20 --warmup 5 --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('welsh voices=$v split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done; done 2>&1 | tee -a gpurun_out/r3_split_ab8.log
