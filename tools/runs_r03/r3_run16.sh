cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest16.log)
tail -25 gpurun_out/r3_pytest16.log | grep -E "passed|failed|FAILED|rc="
(timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench16_driver.json 2> gpurun_out/r3_bench16_driver.err; echo "bench(driver cmd) rc=$?")
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench16_driver.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['timed_region']['ms_per_step_repeats'])
for c in d['configs']: print(' ', c['workload'], round(c['ms_per_step'],4))
for r in d['shard_curve']['welsh-1m']: print('  shard', r['gpus'], r['voices_per_gpu'], round(r['ms_per_step'],4), round(r['implied_efficiency'],3))
PY
