cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest2.log)
tail -25 gpurun_out/r3_pytest2.log | grep -E "passed|failed|FAILED|rc="
B="python bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for hd in "" "--head-unfused" "--no-head-ahead"; do timeout 200 $B --workload chain-4096 $hd 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain-4096 $hd', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']], 'kernel_ms', [round(x,4) for x in d['timed_region']['kernel_ms_repeats']])"; done 2>&1 | tee gpurun_out/r3_chain_ab2.log
for a in "--materialise" "--interleaved --materialise" "--interleaved"; do timeout 300 $B --steps 20 --warmup 5 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('welsh-1m $a', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done 2>&1 | tee gpurun_out/r3_forms2.log
rm -rf gpurun_out/prof_chain; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 > gpurun_out/prof_chain.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_chain/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"{float(r['Percentage']):6.2f}% {float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5}  {r['Name'][:100]}")
PY
