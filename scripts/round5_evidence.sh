#!/bin/bash
# What produced round 5's committed evidence (run through gpurun from the repo root; outputs under gpurun_out/, summaries into profiles/ with scripts/collect_profiles_r05.sh):
#   full GPU tests, driver-style and default bench lines, profile sets of the headline (N = 200) and of configs4 (N = 500, fp32 storage; its counter passes run the memory-tile
#   form: --pmc-serial), the split form's counters (each launch of the pair replayed alone against recorded operands: scripts/profile_split.sh), the kernel trace of the step-wise
#   API through the C++ host (scripts/profile_step.sh), the native multi-GPU host with one device
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
[ -n "$SKIP_PYTEST" ] || { python -m pytest tests -m gpu -q > gpurun_out/r05_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest.log; }
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_style.json 2> gpurun_out/r05_bench.err
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
bash scripts/profile_round.sh r05_a > gpurun_out/r05_profile_n200.log 2>&1
PROFILE_STEPS=40 PROFILE_WARMUP=6 bash scripts/profile_round.sh r05_n500 --landmarks 500 --storage f32 > gpurun_out/r05_profile_n500.log 2>&1
bash scripts/profile_split.sh r05_n500_split 500 > gpurun_out/r05_profile_split.log 2>&1
bash scripts/profile_step.sh r05_step 200 > gpurun_out/r05_profile_step.log 2>&1
python scripts/f32_curve.py 3000 r05_f32_curve_n500 > gpurun_out/r05_f32_curve.log 2>&1
python scripts/make_step_scene.py 200 120 /tmp/multi_scene > /dev/null
cv-monoslam_amd/cslam_replay_multi.bin /tmp/multi_scene/scene.bin /tmp/multi_scene/odo.txt devices=0 frames=100 warmup=10 > gpurun_out/r05_native_multi_1gpu.json 2> gpurun_out/r05_native_multi.err
# the summaries are made HERE (gpurun merges at most 64 MiB back and the raw counter CSVs are larger): profiles/ of this copy -> gpurun_out/profiles_r05/, raw outputs removed
bash scripts/collect_profiles_r05.sh > gpurun_out/r05_collect.log 2>&1
mkdir -p gpurun_out/profiles_r05 && cp profiles/r05_* gpurun_out/profiles_r05/
rm -rf gpurun_out/r05_*_stats gpurun_out/r05_*_fetch gpurun_out/r05_*_write gpurun_out/r05_*_mfma
du -sh gpurun_out
tail -n 3 gpurun_out/r05_pytest.log; tail -c 600 gpurun_out/r05_bench_driver_style.json; tail -n 4 gpurun_out/r05_profile_split.log; cat gpurun_out/r05_native_multi_1gpu.json
