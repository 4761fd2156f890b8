cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
L="groove_amd/libgroove_hip.so groove_amd/libvar_u2.so groove_amd/libvar_u4.so"
{
for m in 3073 0; do export GROOVE_TP_VPW2_MIN_VOICES=$m
REPS=1 tools/ab_bench.sh "--workload chain-4096" $L 2>&1 | sed "s/^/vpw2min=$m chain-4096 /"
REPS=1 tools/ab_bench.sh "--workload welsh-256" $L 2>&1 | sed "s/^/vpw2min=$m welsh-256 /"
REPS=1 tools/ab_bench.sh "--voices 16384" $L 2>&1 | sed "s/^/vpw2min=$m welsh-16384 /"
REPS=1 tools/ab_bench.sh "--voices 4096" $L 2>&1 | sed "s/^/vpw2min=$m welsh-4096 /"
done
} | tee gpurun_out/r3_unroll_ab.log
