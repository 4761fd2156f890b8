# on the GPU box: swap the measurement build in, run the probe over bank sizes / patch sets, swap back
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
cp groove_amd/libvar_probe.so groove_amd/libgroove_hip.so
{
timeout 120 python3 tools/split_probe.py --voices 65536 --patches all
for p in $(seq 0 31); do timeout 120 python3 tools/split_probe.py --voices 49152 --patches $p | tail -4; done
} 2>&1 | tee gpurun_out/r3_split_probe.log
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
{
timeout 120 python3 tools/split_probe.py --voices 65536 --patches all --no-probe
for p in $(seq 0 31); do timeout 120 python3 tools/split_probe.py --voices 49152 --patches $p --no-probe | tail -1; done
} 2>&1 | tee gpurun_out/r3_split_noprobe.log
