#!/bin/bash
# Kernel trace of the DROP-IN path (step-wise API through the C++ host, one frame at a time with the host round trips): rocprofv3 --kernel-trace --stats of
# cslam_step_bench (N = 200, mode capi with the odometry look-ahead).   bash scripts/profile_step.sh <tag> [N] [mode args...]
tag=$1; N=${2:-200}; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 scripts/make_step_scene.py $N 240 /tmp/step_scene > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- cv-monoslam_amd/cslam_step_bench.bin /tmp/step_scene/scene.bin /tmp/step_scene/odo.txt mode=capi hint=1 frames=200 warmup=20 "$@" > gpurun_out/${tag}_stats.log 2>&1
tail -n 2 gpurun_out/${tag}_stats.log
python3 scripts/trace_gaps.py $(ls -t gpurun_out/${tag}_stats/*/*kernel_trace.csv | head -1) | head -30
