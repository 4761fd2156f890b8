cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee gpurun_out/r3_pytest79.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench79_driver.json 2> gpurun_out/r3_bench79_driver.err
timeout 900 python3 bench.py > gpurun_out/r3_bench79_default.json 2> gpurun_out/r3_bench79_default.err
python3 - <<'PY'
import json
for f in ('driver','default'):
    d=json.loads(open(f'gpurun_out/r3_bench79_{f}.json').read().strip().split('\n')[-1])
    print(f, d['value'], d['ms_per_step'], d.get('watchdog'), d['roofline'].get('frac'), d['parity_vs_oracle']['bus_rms_err'])
    for c in d.get('configs',[]): print('  ', c.get('workload'), round(c.get('ms_per_step'),4), c.get('frac'), (c.get('parity_vs_oracle') or {}).get('bus_rms_err'))
    sc=d.get('shard_curve') or {}
    print('  ', [(r['voices_per_gpu'], round(r['ms_per_step'],4), round(r.get('implied_efficiency',0),3)) for r in sc.get('welsh-1m',[])], (sc.get('mixed-131072_shard_of_8') or {}).get('ms_per_step'), (sc.get('mixed-131072_shard_of_8') or {}).get('ms_per_step_repeats'))
PY
