# final evidence of the round: GPU tests, smoke, the two bench lines, rocprofv3 passes of the workloads whose kernels changed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r3_pytest43.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -6 | tee gpurun_out/r3_smoke43.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench43_driver.json 2> gpurun_out/r3_bench43_driver.err
timeout 900 python3 bench.py > gpurun_out/r3_bench43_default.json 2> gpurun_out/r3_bench43_default.err
PASS_TIMEOUT=300 bash tools/profile_round.sh r03 welsh-256 chain-4096 > gpurun_out/r3_profile43.log 2>&1
python3 - <<'PY'
import json
for f in ('driver','default'):
    d=json.loads(open(f'gpurun_out/r3_bench43_{f}.json').read().strip().split('\n')[-1])
    print(f, d['value'], d['ms_per_step'], d.get('watchdog'), d['roofline'].get('frac'))
    for c in d.get('configs',[]): print('  ', c.get('workload'), round(c.get('ms_per_step'),4), c.get('frac'), c.get('bound'), (c.get('parity_vs_oracle') or {}).get('bus_rms_err'), (c.get('parity_vs_oracle') or {}).get('kernel_form'))
    sc=d.get('shard_curve') or {}
    print('  ', [(r['voices_per_gpu'], round(r['ms_per_step'],4), round(r.get('implied_efficiency',0),3)) for r in sc.get('welsh-1m',[])], (sc.get('mixed-131072_shard_of_8') or {}).get('ms_per_step'))
PY
