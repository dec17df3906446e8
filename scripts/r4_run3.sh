#!/bin/bash
# round 4, GPU call 3: the round-3/4 GPU tests again (first-frame semantics, batched filters), tenants probe with k_syrk_own / pair_adjacent
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_parity_r2.py -m gpu -q > gpurun_out/r4c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4c_pytest.log
tail -15 gpurun_out/r4c_pytest.log
python scripts/tenants_probe.py > gpurun_out/r4c_tenants.log 2>&1
cat gpurun_out/r4c_tenants.log
