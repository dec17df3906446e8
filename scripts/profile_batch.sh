#!/bin/bash
# Profile set of the batched replay (srukf_run_frames_batch, B filters at N = 200, 4 groups): kernel trace + stats (graph replay), MFMA counters and HBM
# traffic counters (eager launches, counters only, separate passes).   bash scripts/profile_batch.sh <tag> <B>
tag=$1; B=${2:-32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 scripts/batch_probe.py $B 1 200 4 > gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_mfma -- python3 scripts/batch_probe.py $B 1 200 4 eager > gpurun_out/${tag}_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 scripts/batch_probe.py $B 1 200 4 eager > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 scripts/batch_probe.py $B 1 200 4 eager > gpurun_out/${tag}_write.log 2>&1
cat gpurun_out/${tag}_stats.log | tail -3
