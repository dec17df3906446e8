#!/bin/bash
# Counters for the split form of the factorisation (BASELINE configs[4]: N = 500, fp32 storage): the pair's operands are recorded from a real frame, then each launch is
# replayed ALONE under rocprofv3 (scripts/split_replay.py: the pair itself cannot run under --pmc, which serialises dispatches).   bash scripts/profile_split.sh <tag> [N]
tag=$1; N=${2:-500}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 scripts/split_replay.py record $N /tmp/split_record.npz 4 > gpurun_out/${tag}_record.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 scripts/split_replay.py replay $N /tmp/split_record.npz 8 > gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 scripts/split_replay.py replay $N /tmp/split_record.npz 4 > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 scripts/split_replay.py replay $N /tmp/split_record.npz 4 > gpurun_out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_mfma -- python3 scripts/split_replay.py replay $N /tmp/split_record.npz 4 > gpurun_out/${tag}_mfma.log 2>&1
tail -n 2 gpurun_out/${tag}_record.log gpurun_out/${tag}_stats.log gpurun_out/${tag}_fetch.log gpurun_out/${tag}_mfma.log
