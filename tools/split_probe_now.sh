#!/bin/bash
# tools/split_probe.py on the measurement build (make -C groove_amd BUILD=build_probe OUT=libvar_probe.so EXTRA=-DGROOVE_SPLIT_PROBE), 65,536 voices
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
cp groove_amd/libvar_probe.so groove_amd/libgroove_hip.so
timeout 120 python3 tools/split_probe.py --voices 65536 --patches all 2>&1 | tee gpurun_out/r5/split_probe.log
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
