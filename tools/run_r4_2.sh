set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4
timeout 300 tools/micro/mix_bound > gpurun_out/r4/mix_bound.json 2> gpurun_out/r4/mix_bound.err; cat gpurun_out/r4/mix_bound.json
timeout 600 tools/micro/inst_cost > gpurun_out/r4/inst_cost.log 2>&1; tail -3 gpurun_out/r4/inst_cost.log
rm -f gpurun_out/stress_fresh.log
bash tools/stress_fresh.sh 100 60; cp gpurun_out/stress_fresh.log gpurun_out/r4/stress_fresh_100.log
PASS_TIMEOUT=300 bash tools/profile_round.sh r04 welsh-1m-window 2>&1 | tail -15
PASS_TIMEOUT=300 MIX=0 bash tools/profile_round.sh r04 chain-4096 2>&1 | tail -12
