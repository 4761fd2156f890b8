cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r3_pytest25.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/r3_smoke25.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench25.json 2> gpurun_out/r3_bench25.err; tail -c 600 gpurun_out/r3_bench25.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench25.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('watchdog'))
for c in d.get('configs',[]): print(c.get('workload'), c.get('ms_per_step'), c.get('kernel_form','')[:60])
print(d.get('shard_curve'))
PY
