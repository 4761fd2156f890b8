set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 300 tools/micro/mix_bound > gpurun_out/r4/mix_bound2.json 2>/dev/null; cat gpurun_out/r4/mix_bound2.json
timeout 900 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r4/gputests6.log; cat gpurun_out/r4/gputests6.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke6.log 2>&1; tail -3 gpurun_out/r4/smoke6.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_line_driver_command.json 2> gpurun_out/r4/bench_driver.err
timeout 1500 python3 bench.py > gpurun_out/r4/bench_line_default.json 2> gpurun_out/r4/bench_default.err
tail -c 600 gpurun_out/r4/bench_line_default.json
PASS_TIMEOUT=300 bash tools/profile_round.sh r04 welsh-1m-window welsh-1m 2>&1 | grep -v "^ " | tail -4
PASS_TIMEOUT=300 MIX=0 bash tools/profile_round.sh r04 welsh-256 chain-4096 sampler-16384 mixed-131072 2>&1 | grep -v "^ " | tail -6
