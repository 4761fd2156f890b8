#!/bin/bash
# The small-bank workloads (time-parallel kernels) for two or more builds of the library inside ONE gpurun job:
#   tools/small_ab.sh groove_amd/libvar_A.so groove_amd/libgroove_hip.so
cd "${GRAFT_REPO_ROOT:-.}"
cp groove_amd/libgroove_hip.so /tmp/base_lib.so
for rep in 1 2; do
  for lib in "$@"; do
    src="$lib"; [ "$(basename $lib)" = "libgroove_hip.so" ] && src=/tmp/base_lib.so
    cp "$src" groove_amd/libgroove_hip.so
    for w in welsh-256 chain-4096 sampler-16384; do
      v=$(timeout 200 python3 bench.py --workload $w --no-cpu-baseline --no-configs --no-parity --no-shard-curve --no-watchdog --repeats 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(f\"{d['ms_per_step']:.5f}\")")
      echo "$(basename $lib) $w $v ms/step"
    done
    python3 tools/mixed_ab.py --voices 16384 --vpw 8 --rounds 1 2>&1 | grep "one launch" | sed "s|^|$(basename $lib) |"
  done
done
cp /tmp/base_lib.so groove_amd/libgroove_hip.so
