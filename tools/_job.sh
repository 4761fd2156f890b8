set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 1200 python3 -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4/gputests10.log; cat gpurun_out/r4/gputests10.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_line_driver_command_b.json 2> gpurun_out/r4/bench_driver_b.err; tail -c 400 gpurun_out/r4/bench_line_driver_command_b.json
