cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
L="groove_amd/libgroove_hip.so groove_amd/libvar_w4.so"
{
timeout 900 python3 -m pytest tests/test_gpu_time_parallel.py tests/test_gpu_welsh.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
REPS=2 tools/ab_bench.sh "--workload chain-4096" $L 2>&1 | sed "s/^/chain-4096 /"
REPS=2 tools/ab_bench.sh "--workload welsh-256" $L 2>&1 | sed "s/^/welsh-256 /"
REPS=1 tools/ab_bench.sh "--voices 16384" $L 2>&1 | sed "s/^/welsh-16384 /"
REPS=1 tools/ab_bench.sh "--voices 4096" $L 2>&1 | sed "s/^/welsh-4096 /"
REPS=1 tools/ab_bench.sh "--voices 2048" $L 2>&1 | sed "s/^/welsh-2048 /"
GROOVE_TP_VPW2_MIN_VOICES=0 REPS=1 tools/ab_bench.sh "--voices 4096" $L 2>&1 | sed "s/^/vpw1 welsh-4096 /"
GROOVE_TP_VPW2_MIN_VOICES=0 REPS=1 tools/ab_bench.sh "--workload chain-4096" $L 2>&1 | sed "s/^/vpw1 chain-4096 /"
} | tee gpurun_out/r3_single_inst_ab.log
