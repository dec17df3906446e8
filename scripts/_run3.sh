cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s -k "plan_threshold or other_forms or g9_n500_fp32 or g10 or step_api or frame_view or native_multi" > gpurun_out/r5_pytest_new2.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_new2.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_r5.py > gpurun_out/r5_pytest_old.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_old.log
python - <<'PY' > gpurun_out/step_bench.log 2>&1
import json, sys
sys.path.insert(0, ".")
import bench
import __graft_entry__ as ge
print(json.dumps(bench.step_api_leg(ge.load_package().synth), indent=1))
PY
grep -E "passed|failed|fp32 storage|g9 f32|mixed|rc=" gpurun_out/r5_pytest_new2.log | tail -n 30; tail -n 5 gpurun_out/r5_pytest_old.log; cat gpurun_out/step_bench.log | tail -n 60
