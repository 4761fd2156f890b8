cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest3.log)
tail -25 gpurun_out/r3_pytest3.log | grep -E "passed|failed|FAILED|rc="
B="python bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for hd in "" "--head-unfused" "--no-head-ahead"; do timeout 200 $B --workload chain-4096 $hd 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain-4096 $hd', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done 2>&1 | tee gpurun_out/r3_chain_ab3.log
GROOVE_FX_LDS_STAGING=1 timeout 200 $B --workload chain-4096 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain-4096 LDS-staged chorus taps', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])" | tee -a gpurun_out/r3_chain_ab3.log
# LDS staging A/B: kernel time and FETCH_SIZE of fx_run_kernel<4>, both variants
for v in 0 1; do
  rm -rf gpurun_out/lds_kt_$v gpurun_out/lds_fetch_$v
  GROOVE_FX_LDS_STAGING=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lds_kt_$v -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 --no-render-ahead > gpurun_out/lds_kt_$v.log 2>&1
  GROOVE_FX_LDS_STAGING=$v timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/lds_fetch_$v -- python3 bench.py --workload chain-4096 --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve --repeats 1 --no-render-ahead > gpurun_out/lds_fetch_$v.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/r3_lds_ab.log
import csv, glob, collections
for v in (0, 1):
    for f in glob.glob(f"gpurun_out/lds_kt_{v}/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "fx_run_kernel" in r["Name"]: print(f"LDS_STAGING={v} fx_run_kernel avg {float(r['AverageNs'])/1e3:.2f} us x {r['Calls']}")
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/lds_fetch_{v}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fx_run_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, vals in agg.items(): print(f"LDS_STAGING={v} fx_run_kernel {k} mean {sum(vals)/len(vals):.1f} KiB per launch ({len(vals)} launches)")
PY
REPS=3 tools/ab_bench.sh "--steps 20 --warmup 5 --materialise" groove_amd/libgroove_hip.so groove_amd/libvar_nt.so 2>&1 | tee gpurun_out/r3_nt_ab.log
GROOVE_TP_MAX_VOICES=0 timeout 600 python3 tools/patch_cost.py --voices 2048 --blocks 40 2>&1 | tee gpurun_out/r3_patch_cost_2048.log | tail -40
(timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err; echo "bench rc=$?")
