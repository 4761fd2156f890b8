# the stall with a heartbeat: the diagnostic build (-DGROOVE_HEARTBEAT: workgroups of the per-kind kernels started / finished, in
# host memory) in place of the library, the driver's command with nine regions, N fresh runs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so /tmp/base_lib.so; cp groove_amd/libvar_hb.so groove_amd/libgroove_hip.so
rm -f gpurun_out/r3_stall_hunt9.log
for i in $(seq 1 ${1:-30}); do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 9 --no-cpu-baseline --no-configs --no-shard-curve --no-parity > /tmp/hunt_$i.json 2> /tmp/hunt_$i.err
  python3 - <<PY >> gpurun_out/r3_stall_hunt9.log
import json
d=json.loads(open('/tmp/hunt_$i.json').read().strip().split('\n')[-1])
print('run $i', round(d['ms_per_step'],4), d.get('watchdog'))
PY
  grep -h "stall was seen while\|still busy" /tmp/hunt_$i.err | cut -c1-1200 >> gpurun_out/r3_stall_hunt9.log
done
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
grep -c "killed': 0" gpurun_out/r3_stall_hunt9.log; grep -o "heartbeat.*finished [0-9]*" gpurun_out/r3_stall_hunt9.log
