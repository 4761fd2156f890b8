cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > gpurun_out/r3_pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest1.log)
tail -5 gpurun_out/r3_pytest1.log
(timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?")
tail -c 600 gpurun_out/r3_bench1.err
for hd in "" "--no-head-ahead"; do timeout 200 python bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog $hd 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain-4096 $hd', d['ms_per_step'], d['timed_region']['ms_per_step_repeats'])"; done 2>&1 | tee gpurun_out/r3_chain_ab1.log
tools/micro/stall_repro.sh 30 25 | tail -2
