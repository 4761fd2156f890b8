# on the GPU box: swap the measurement build in, run the probe, swap back
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
cp groove_amd/libvar_tpprobe.so groove_amd/libgroove_hip.so
timeout 300 python3 tools/tp_probe.py 256 4096 16384 2>&1 | tee gpurun_out/r3_tp_probe.log
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
timeout 300 python3 tools/tp_bench.py 256 4096 16384 2>&1 | tee gpurun_out/r3_tp_bench0.log
