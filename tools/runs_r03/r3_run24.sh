cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
run() { label=$1; shift; env "$@" | tail -1 | L="$label" python3 -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['L'], round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; }
{
for rep in 1 2; do
  cp /tmp/base_lib.so groove_amd/libgroove_hip.so
  run "base mixed-131072 roles3" GROOVE_SPLIT_ROLES=3 timeout 200 $B --workload mixed-131072 2>/dev/null
  run "base mixed-131072 roles4" GROOVE_SPLIT_ROLES=4 timeout 200 $B --workload mixed-131072 2>/dev/null
  cp groove_amd/libvar_w5.so groove_amd/libgroove_hip.so
  run "w5 mixed-131072 roles4" GROOVE_SPLIT_ROLES=4 timeout 200 $B --workload mixed-131072 2>/dev/null
  run "w5 welsh-65536 roles4" GROOVE_SPLIT_ROLES=4 timeout 200 $B --workload welsh-1m --voices 65536 2>/dev/null
  run "w5 welsh-65536 roles3" GROOVE_SPLIT_ROLES=3 timeout 200 $B --workload welsh-1m --voices 65536 2>/dev/null
done
} 2>&1 | tee gpurun_out/r3_split4_w5.log
cp groove_amd/libvar_probe.so groove_amd/libgroove_hip.so
{
GROOVE_SPLIT_ROLES=4 timeout 120 python3 tools/split_probe.py --voices 65536 --patches all
for p in 0 1 2 3 4 8 26 30; do GROOVE_SPLIT_ROLES=4 timeout 120 python3 tools/split_probe.py --voices 49152 --patches $p | tail -5; done
} 2>&1 | tee gpurun_out/r3_split4_probe.log
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
