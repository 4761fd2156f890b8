cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench59_driver_$i.json 2> gpurun_out/r3_bench59_driver_$i.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r3_bench59_driver_$i.json').read().strip().split('\n')[-1])
print('driver run $i', d['value'], d['ms_per_step'], d.get('watchdog'), [ (c['workload'], round(c['ms_per_step'],4)) for c in d['configs']], d['shard_curve']['mixed-131072_shard_of_8']['ms_per_step'])
PY
tail -2 gpurun_out/r3_bench59_driver_$i.err
done
