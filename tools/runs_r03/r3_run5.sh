cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest5.log)
tail -25 gpurun_out/r3_pytest5.log | grep -E "passed|failed|FAILED|rc="
REPS=3 tools/ab_bench.sh "--steps 20 --warmup 5" groove_amd/libvar_prev.so groove_amd/libgroove_hip.so 2>&1 | tee gpurun_out/r3_refactor_ab.log
REPS=2 tools/ab_bench.sh "--workload mixed-131072" groove_amd/libvar_prev.so groove_amd/libgroove_hip.so 2>&1 | tee -a gpurun_out/r3_refactor_ab.log
