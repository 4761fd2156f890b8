cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so groove_amd/libvar_new.so
REPS=3 tools/ab_bench.sh "--workload mixed-131072" groove_amd/libvar_new.so groove_amd/libvar_prio.so 2>&1 | tee gpurun_out/r3_prio_ab.log
REPS=2 tools/ab_bench.sh "--steps 20 --warmup 5 --voices 65536" groove_amd/libvar_new.so groove_amd/libvar_prio.so 2>&1 | tee -a gpurun_out/r3_prio_ab.log
(GROOVE_SAFE_STREAMS=1 timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r3_pytest10_safe.log 2>&1; echo "pytest(safe streams) rc=$?" >> gpurun_out/r3_pytest10_safe.log)
tail -25 gpurun_out/r3_pytest10_safe.log | grep -E "passed|failed|FAILED|rc="
