#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python scripts/batch_soak.py 32 3000 500 r04_batch_soak > gpurun_out/r4o_soak.log 2>&1; tail -12 gpurun_out/r4o_soak.log
