cd "${GRAFT_REPO_ROOT:-.}"
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
cp groove_amd/libvar_ap.so groove_amd/libgroove_hip.so
for rep in 1 2 3; do
for k in -1 0 1 2 7; do
  v=$(GROOVE_FX_AP_STREAM=$k timeout 200 python3 bench.py --workload chain-4096 --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.4f} ms/step\")")
  echo "ap stream $k: $v"
done
done
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
