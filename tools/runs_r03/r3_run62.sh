cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-shard-curve"
GROOVE_BENCH_FORCE_DIST=1 timeout 300 $B 2>gpurun_out/r3_dist62.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist default', round(d['ms_per_step'],4), d.get('rccl_ranks'), d.get('streams',{}).get('comm_before_streams'), d['config'].get('bus_reduce'), d['parity_vs_oracle']['bus_rms_err'])"
GROOVE_BENCH_FORCE_DIST=1 timeout 300 $B --workload mixed-131072 2>>gpurun_out/r3_dist62.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('force-dist mixed', round(d['ms_per_step'],4), d.get('rccl_ranks'))"
timeout 300 python3 -m pytest tests/test_gpu_errors.py -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
