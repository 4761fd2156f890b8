#!/bin/bash
# The round's evidence in ONE gpurun job, on the build that is in the tree (run through gpurun from the repository root):
#   gpurun --timeout 4500 -- tools/take_evidence.sh r06
# 1. the GPU tier; 2. rocprofv3 passes of all eight workloads (tools/profile_round.sh: kernel trace + PMC passes, summaries written to
# profiles/ ON THE BOX so that the bench lines below find the profile of their own library); 3. the mix kernel's LDS / issue / fetch
# counters (tools/diag_pmc.sh); 4. the measured vector-issue bound (tools/micro/mix_bound); 5. the driver's command, the default line
# and the library-proportioned bank; 6. the fresh-process gate (25 processes, one bus CRC); 7. sixty fresh seeds per seeded GPU test;
# 8. the repeat-render soak.  Everything that is to be committed lands in gpurun_out/ev/ (copy it into profiles/ in the development
# container: summaries and CSVs as they are, line_*.json -> r06_bench_line_*.json, detail_*.json -> r06_bench_detail_*.json).
set -u
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_tier.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/gpu_tier.log | head -5
MIX=1 timeout 2400 tools/profile_round.sh $R welsh-1m-window welsh-1m welsh-1m-library-window welsh-1m-materialised-window welsh-256 chain-4096 sampler-16384 mixed-131072 > gpurun_out/profile_round.log 2>&1
rm -rf gpurun_out/ev; mkdir -p gpurun_out/ev
tools/diag_pmc.sh; python3 tools/diag_pmc.py gpurun_out/diag_pmc > profiles/${R}_diag_pmc.json
tools/micro/mix_bound > profiles/${R}_mix_bound.json 2> gpurun_out/mix_bound.err
cp profiles/${R}_*summary.json profiles/${R}_*_kernel_stats.csv profiles/${R}_diag_pmc.json profiles/${R}_mix_bound.json gpurun_out/ev/
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/ev/line_driver.json 2> gpurun_out/ev/line_driver.err; cp bench_detail.json gpurun_out/ev/detail_driver.json
timeout 600 python bench.py > gpurun_out/ev/line_default.json 2> gpurun_out/ev/line_default.err; cp bench_detail.json gpurun_out/ev/detail_default.json
timeout 300 python bench.py --workload welsh-1m-library --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ev/line_library.json 2>/dev/null
rm -f gpurun_out/stress_fresh.log; tools/stress_fresh.sh 25 60 | tail -1; cp gpurun_out/stress_fresh.log gpurun_out/ev/${R}_stress_fresh_25.log
GROOVE_TEST_SEEDS=60 GROOVE_TEST_SEED_BASE=${SEED_BASE:-63000} timeout 900 python -m pytest tests -m gpu -q -k "random or seed or drawn" > gpurun_out/seeds.log 2>&1; grep -E "passed|failed|^FAILED" gpurun_out/seeds.log | head -3
timeout 600 python tools/soak.py > gpurun_out/ev/${R}_soak.json 2>/dev/null
cut -c1-200 gpurun_out/ev/line_driver.json
