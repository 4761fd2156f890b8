cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
cp groove_amd/libgroove_hip.so groove_amd/libvar_any3.so
for v in 125000 250000 400000 500000; do REPS=2 tools/ab_bench.sh "--steps 20 --warmup 5 --voices $v" groove_amd/libvar_any3.so groove_amd/libvar_any4.so groove_amd/libvar_any5.so 2>&1 | sed "s/^/voices=$v /"; done | tee gpurun_out/r3_any_ab.log
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
for fm in 131072 0; do GROOVE_FM_TP_MAX_VOICES=$fm timeout 200 $B --workload mixed-131072 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mixed-131072 fm_tp_max=$fm', round(d['ms_per_step'],4), [round(x,4) for x in d['timed_region']['ms_per_step_repeats']])"; done 2>&1 | tee gpurun_out/r3_fm_ab.log
