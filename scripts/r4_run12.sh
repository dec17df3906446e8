#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4l_pytest.log
tail -6 gpurun_out/r4l_pytest.log
