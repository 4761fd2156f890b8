cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3_pytest40.log 2>&1; tail -5 gpurun_out/r3_pytest40.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for v in 1024 2048 3072; do for m in 1 3073; do
GROOVE_TP_VPW2_MIN_VOICES=$m timeout 200 $B --voices $v 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('voices $v vpw2_min=$m', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"
done; done 2>&1 | tee gpurun_out/r3_vpw_sweep.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3_bench40.json 2> gpurun_out/r3_bench40.err
tail -3 gpurun_out/r3_bench40.err
python3 - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench40.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('watchdog'))
for c in d.get('configs',[]): print(c['workload'], round(c['ms_per_step'],4), c.get('roofline',{}).get('frac'), c.get('kernel_form'))
sc=d.get('shard_curve',{})
print([ (r['voices_per_gpu'], round(r['ms_per_step'],4)) for r in sc.get('welsh-1m',[])], sc.get('mixed-131072_shard_of_8',{}).get('ms_per_step'))
P
