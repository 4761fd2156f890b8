# RECORD of a round-3 measurement job (kept because docs/HISTORY.md cites its log): the build variants and GROOVE_* knobs it names were
# measured, not kept, and removed in round 4 — the script documents how the numbers were taken; it no longer runs against this tree.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-parity --no-watchdog --no-configs --no-shard-curve"
# non-temporal block stores, in-job A/B (materialised million-voice form)
REPS=3 tools/ab_bench.sh "--steps 20 --warmup 5 --materialise" groove_amd/libgroove_hip.so groove_amd/libvar_nt.so 2>&1 | tee gpurun_out/r3_nt_ab.log
# LDS staging A/B: FETCH_SIZE and WRITE_SIZE of fx_run_kernel<4>, separate passes
for v in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/lds_${c}_$v
  GROOVE_FX_LDS_STAGING=$v timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/lds_${c}_$v -- $B --workload chain-4096 --repeats 1 --no-render-ahead > gpurun_out/lds_${c}_$v.log 2>&1
done; done
python3 - <<'PY' | tee -a gpurun_out/r3_lds_ab.log
import csv, glob, collections
for v in (0, 1):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/lds_{c}_{v}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "fx_run_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        if vals: print(f"LDS_STAGING={v} fx_run_kernel {c} mean {sum(vals)/len(vals):.1f} KiB per launch ({len(vals)} launches)")
PY
# liveness gate: fresh processes on the million-voice path, default and safe stream layouts
tools/stress_fresh.sh 16 60
tools/stress_fresh.sh 6 60 GROOVE_SAFE_STREAMS=1
# the round's profiles: every workload + the driver's window
PASS_TIMEOUT=300 tools/profile_round.sh r03 welsh-1m-window welsh-1m chain-4096 mixed-131072 sampler-16384 welsh-256 2>&1 | tail -40
