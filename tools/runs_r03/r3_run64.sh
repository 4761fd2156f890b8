cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench64_driver_$i.json 2> gpurun_out/r3_bench64_driver_$i.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r3_bench64_driver_$i.json').read().strip().split('\n')[-1])
print('driver run $i', d['value'], d['ms_per_step'], d.get('watchdog'), [ (c['workload'], round(c['ms_per_step'],4)) for c in d['configs']], d['shard_curve']['mixed-131072_shard_of_8']['ms_per_step'])
PY
tail -2 gpurun_out/r3_bench64_driver_$i.err
done
timeout 900 python3 bench.py > gpurun_out/r3_bench64_default.json 2> gpurun_out/r3_bench64_default.err
python3 - <<'PY'
import json
for f in ('driver_1','default'):
    d=json.loads(open(f'gpurun_out/r3_bench64_{f}.json').read().strip().split('\n')[-1])
    print(f, d['value'], d['ms_per_step'], d.get('watchdog'), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])
    for c in d.get('configs',[]): print('  ', c.get('workload'), round(c.get('ms_per_step'),4), [round(x,4) for x in c.get('ms_per_step_repeats',[])])
    sc=d.get('shard_curve') or {}
    print('  ', [(r['voices_per_gpu'], round(r['ms_per_step'],4), round(r.get('implied_efficiency',0),3)) for r in sc.get('welsh-1m',[])], (sc.get('mixed-131072_shard_of_8') or {}).get('ms_per_step'))
PY
