# the GPU suite under the alternative settings: safe stream layout; deferral and event binding off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "== GROOVE_SAFE_STREAMS=1"; GROOVE_SAFE_STREAMS=1 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
echo "== GROOVE_DEFER_BUS=0 GROOVE_BIND_EVENTS=0"; GROOVE_DEFER_BUS=0 GROOVE_BIND_EVENTS=0 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
echo "== GROOVE_TP_VPW2_MIN_VOICES=0 GROOVE_FM_TP_VPW4_MIN_VOICES=0"; GROOVE_TP_VPW2_MIN_VOICES=0 GROOVE_FM_TP_VPW4_MIN_VOICES=0 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
} 2>&1 | tee gpurun_out/r3_alt_settings.log
