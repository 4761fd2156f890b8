cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_split.py -q -x -p no:cacheprovider > gpurun_out/r3_pytest6a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest6a.log)
tail -30 gpurun_out/r3_pytest6a.log | grep -E "passed|failed|FAILED|rc=|Error|assert" | head -20
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for sp in 2048 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --workload mixed-131072 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed-131072 split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], d['config']['kernel_form'])"; done 2>&1 | tee gpurun_out/r3_split_ab.log
for v in 65536 125000 250000; do for sp in 4096 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --steps 20 --warmup 5 --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('welsh voices=$v split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done; done 2>&1 | tee -a gpurun_out/r3_split_ab.log
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest6.log)
tail -25 gpurun_out/r3_pytest6.log | grep -E "passed|failed|FAILED|rc="
cp groove_amd/libgroove_hip.so groove_amd/libvar_new.so
REPS=3 tools/ab_bench.sh "--steps 20 --warmup 5" groove_amd/libvar_prev.so groove_amd/libvar_new.so 2>&1 | tee gpurun_out/r3_refactor_ab.log
REPS=3 tools/ab_bench.sh "--steps 20 --warmup 5 --materialise" groove_amd/libvar_new.so groove_amd/libvar_nt.so 2>&1 | tee gpurun_out/r3_nt_ab.log
