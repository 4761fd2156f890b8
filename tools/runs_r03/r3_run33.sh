cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for p in 0 1; do for b in 0 1; do echo "paced=$p bind=$b"; GROOVE_HOST_PACED=$p GROOVE_BIND_EVENTS=$b python3 tools/step_submit_cost.py chain-4096 mixed-131072 2>&1 | tail -2; done; done | tee gpurun_out/r3_submit_cost2.log
