cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2 3; do
timeout 900 python3 bench.py > gpurun_out/r3_bench52_default_$i.json 2> gpurun_out/r3_bench52_default_$i.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r3_bench52_default_$i.json').read().strip().split('\n')[-1])
print('default run $i', d['value'], d['ms_per_step'], d.get('watchdog'))
PY
tail -2 gpurun_out/r3_bench52_default_$i.err
done
rm -f gpurun_out/stress_fresh.log
bash tools/stress_fresh.sh 30 90
tail -3 gpurun_out/stress_fresh.log
