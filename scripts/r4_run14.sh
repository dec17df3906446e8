#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_facade.py tests/test_gpu_parity_r2.py -m gpu -q -s -k "recover_from_flagged or map_changes_mid or batched_filters or three_filters" > gpurun_out/r4n_pytest.log 2>&1; grep -v "^$" gpurun_out/r4n_pytest.log | tail -25
