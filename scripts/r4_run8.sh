#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -m gpu -q -x -k "batched_filters" > gpurun_out/r4h_pytest.log 2>&1; tail -30 gpurun_out/r4h_pytest.log
timeout 900 python scripts/batch_probe.py 4,8,16,24,32 1 200 1,2,3 > gpurun_out/r4h_batch.log 2>&1
cat gpurun_out/r4h_batch.log
