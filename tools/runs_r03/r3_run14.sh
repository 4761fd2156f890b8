cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
# the scratch hypothesis: round 2's form of the class bodies (external linkage: 84-172 bytes of callee-saved scratch per lane) under the same gate
cp groove_amd/libgroove_hip.so /tmp/final_lib.so
cp groove_amd/libvar_extern.so groove_amd/libgroove_hip.so
rm -f gpurun_out/stress_fresh.log
tools/stress_fresh.sh 50 45 GROOVE_KIND_STREAMS=4
tools/stress_fresh.sh 50 45
cp gpurun_out/stress_fresh.log gpurun_out/stress_fresh_extern_bodies.log
cp /tmp/final_lib.so groove_amd/libgroove_hip.so
