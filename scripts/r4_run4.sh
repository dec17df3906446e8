#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity_r3.py -m gpu -q -k "fused_tail_agrees" > gpurun_out/r4d_pytest.log 2>&1; tail -3 gpurun_out/r4d_pytest.log
python scripts/tenants_probe.py 4 4,5 106,200 0,1 > gpurun_out/r4d_tenants.log 2>&1
python scripts/tenants_probe.py 3 3,4 106,200 0,1 >> gpurun_out/r4d_tenants.log 2>&1
cat gpurun_out/r4d_tenants.log
