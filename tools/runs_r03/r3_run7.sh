cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_split.py -q -x -p no:cacheprovider > gpurun_out/r3_pytest7a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest7a.log)
tail -30 gpurun_out/r3_pytest7a.log | grep -E "passed|failed|FAILED|rc=|Error|assert" | head -20
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for sp in 2048 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --workload mixed-131072 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed-131072 split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done 2>&1 | tee gpurun_out/r3_split_ab7.log
for v in 32768 65536 90000 125000; do for sp in 4096 0; do GROOVE_SPLIT_MAX_WAVES=$sp timeout 200 $B --steps 20 --warmup 5 --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('welsh voices=$v split_max=$sp', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done; done 2>&1 | tee -a gpurun_out/r3_split_ab7.log
rm -rf gpurun_out/prof_split; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_split -- python3 bench.py --workload mixed-131072 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 > gpurun_out/prof_split.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_split/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(f"{float(r['Percentage']):6.2f}% {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5}  {r['Name'][:100]}")
PY
