cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for free in 16 24; do timeout 120 python scripts/abort_probe.py 270 1 3 $free; done > gpurun_out/abort_probe2.log 2>&1
timeout 120 python scripts/abort_probe.py 267 1 3 >> gpurun_out/abort_probe2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s > gpurun_out/r5_pytest_new3.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_new3.log
bash scripts/profile_step.sh r05_step 200 > gpurun_out/r05_step_profile.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_r5.py > gpurun_out/r5_pytest_old2.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_old2.log
cat gpurun_out/abort_probe2.log; grep -E "passed|failed|fp32 storage|g9 f32|g10|mixed|rc=" gpurun_out/r5_pytest_new3.log | tail -n 30; tail -n 3 gpurun_out/r5_pytest_old2.log; tail -n 40 gpurun_out/r05_step_profile.log
