B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog"
for pass in 1 2; do
for E in bank placeholder; do  # (GROOVE_EXACT_STREAM: b = first bank stream, anything else = the placeholder stream, the default)
  GROOVE_EXACT_STREAM=$E timeout 300 $B --workload welsh-1m-library > /tmp/o.json 2>/dev/null
  python3 -c "
import json
l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); d=json.load(open('bench_detail.json'))
print('library exact-stream=$E pass $pass: %.4f' % l['ms_per_step'], ['%.4f'%x for x in d['timed_region']['ms_per_step_repeats']])"
done; done
timeout 300 $B --workload welsh-1m > /tmp/o.json 2>/dev/null; python3 -c "
import json
l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); d=json.load(open('bench_detail.json'))
print('headline: %.4f' % l['ms_per_step'], ['%.4f'%x for x in d['timed_region']['ms_per_step_repeats']])"
