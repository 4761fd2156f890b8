cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
L="groove_amd/libgroove_hip.so groove_amd/libvar_tpw4.so groove_amd/libvar_st2.so groove_amd/libvar_st3.so groove_amd/libvar_st4.so"
{
REPS=2 tools/ab_bench.sh "--workload chain-4096" $L 2>&1 | sed "s/^/chain-4096 /"
REPS=2 tools/ab_bench.sh "--workload welsh-256" $L 2>&1 | sed "s/^/welsh-256 /"
REPS=1 tools/ab_bench.sh "--voices 16384" $L 2>&1 | sed "s/^/welsh-16384 /"
REPS=1 tools/ab_bench.sh "--voices 4096" $L 2>&1 | sed "s/^/welsh-4096 /"
cp groove_amd/libvar_st4.so groove_amd/libgroove_hip.so
timeout 600 python3 -m pytest tests/test_gpu_time_parallel.py tests/test_gpu_welsh.py -x -q -m gpu 2>&1 | tail -5
} | tee gpurun_out/r3_tpw4_ab.log
