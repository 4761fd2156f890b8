cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/stress_fresh.log
tools/stress_fresh.sh 60 60
tools/stress_fresh.sh 20 60 GROOVE_KIND_STREAMS=4
tools/micro/stall_repro.sh 60 25 | tail -2
(timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-shard-curve --no-cpu-baseline > gpurun_out/r3_bench13_driver.json 2> gpurun_out/r3_bench13.err; echo "bench rc=$?")
python3 -c "import json; d=json.loads(open('gpurun_out/r3_bench13_driver.json').read().strip().splitlines()[-1]); print('driver cmd on this box', d['ms_per_step'], d['timed_region']['ms_per_step_repeats'], d['watchdog'])"
