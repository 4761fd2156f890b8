# the product build with the segment guard: the driver's command with nine regions, N fresh runs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/r3_stall_hunt10.log
for i in $(seq 1 ${1:-40}); do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline > /tmp/hunt_$i.json 2> /tmp/hunt_$i.err
  python3 - <<PY >> gpurun_out/r3_stall_hunt10.log
import json
d=json.loads(open('/tmp/hunt_$i.json').read().strip().split('\n')[-1])
print('run $i', round(d['ms_per_step'],4), d.get('watchdog'))
PY
  grep -h "stall was seen while\|still busy" /tmp/hunt_$i.err | cut -c1-600 >> gpurun_out/r3_stall_hunt10.log
done
grep -c "killed': 0" gpurun_out/r3_stall_hunt10.log; grep -c "killed': [12]" gpurun_out/r3_stall_hunt10.log
