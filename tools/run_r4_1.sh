set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
L=gpurun_out/r4/zero_hunt.log
echo "== shadow-in-min diagnostic build (round 3's minimum), 1,000,000 voices" > $L
GROOVE_LIB_PATH=$PWD/groove_amd/libgroove_diag_shadow.so timeout 600 python3 tools/zero_segment_hunt.py --iters 1500 --label shadow-in-min >> $L 2>&1
echo "== product build" >> $L
timeout 600 python3 tools/zero_segment_hunt.py --iters 1500 --label product >> $L 2>&1
echo "== shadow-in-min diagnostic build, 700,000 voices" >> $L
GROOVE_LIB_PATH=$PWD/groove_amd/libgroove_diag_shadow.so timeout 600 python3 tools/zero_segment_hunt.py --iters 1000 --voices 700000 --label shadow-in-min-700k >> $L 2>&1
cat $L
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/gputests1.log; cat gpurun_out/r4/gputests1.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_driver_1.json 2> gpurun_out/r4/bench_driver_1.err; tail -c 3000 gpurun_out/r4/bench_driver_1.json | head -c 1500; tail -5 gpurun_out/r4/bench_driver_1.err
