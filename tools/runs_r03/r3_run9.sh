cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest9.log)
tail -25 gpurun_out/r3_pytest9.log | grep -E "passed|failed|FAILED|rc="
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke9.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3_smoke9.log); tail -5 gpurun_out/r3_smoke9.log
(timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench9_driver.json 2> gpurun_out/r3_bench9_driver.err; echo "bench(driver cmd) rc=$?")
(timeout 900 python bench.py > gpurun_out/r3_bench9_default.json 2> gpurun_out/r3_bench9_default.err; echo "bench(default) rc=$?")
PASS_TIMEOUT=300 tools/profile_round.sh r03 welsh-1m-window welsh-1m chain-4096 mixed-131072 sampler-16384 welsh-256 2>&1 | grep -E "^(welsh|chain|mixed|sampler)" | cut -c1-200
