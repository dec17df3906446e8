cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s -k "step_api or frame_view or facade or g8_step or association or smoke" > gpurun_out/r5_pytest_step.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_step.log
python - <<'PY' > gpurun_out/step_bench2.log 2>&1
import json, sys
sys.path.insert(0, ".")
import bench
import __graft_entry__ as ge
print(json.dumps(bench.step_api_leg(ge.load_package().synth), indent=1))
PY
bash scripts/profile_step.sh r05_step 200 > gpurun_out/r05_step_profile.log 2>&1
tail -n 4 gpurun_out/r5_pytest_step.log; grep -E "frames_per_s|us_per_frame|\"n|capi|facade|assoc" gpurun_out/step_bench2.log; tail -n 30 gpurun_out/r05_step_profile.log
