# hunting the stall: the driver's command N times, binding of events on and off in alternation, phase of a stall in the log
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/r3_stall_hunt.log
for i in $(seq 1 ${1:-12}); do
  b=$((i % 2))
  GROOVE_BIND_EVENTS=$b timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /tmp/hunt_$i.json 2> /tmp/hunt_$i.err
  python3 - <<PY >> gpurun_out/r3_stall_hunt.log
import json
d=json.loads(open('/tmp/hunt_$i.json').read().strip().split('\n')[-1])
print('run $i bind=$b', round(d['ms_per_step'],4), d.get('watchdog'))
PY
  grep -h "stall was seen while\|still busy" /tmp/hunt_$i.err | cut -c1-300 >> gpurun_out/r3_stall_hunt.log
done
cat gpurun_out/r3_stall_hunt.log
