cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so groove_amd/libvar_final.so
REPS=2 tools/ab_bench.sh "--steps 20 --warmup 5 --materialise" groove_amd/libvar_final.so groove_amd/libvar_probe1.so groove_amd/libvar_probe2.so 2>&1 | tee gpurun_out/r3_store_probe.log
REPS=1 tools/ab_bench.sh "--steps 20 --warmup 5" groove_amd/libvar_final.so 2>&1 | tee -a gpurun_out/r3_store_probe.log
tools/micro/write_bw | tee -a gpurun_out/r3_store_probe.log
