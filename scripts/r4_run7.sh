#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python scripts/batch_probe.py 8,12,16,24 1 200 1,2,3,4 > gpurun_out/r4g_batch.log 2>&1
cat gpurun_out/r4g_batch.log
