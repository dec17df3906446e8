#!/bin/bash
# What produced round 6's committed evidence (run through gpurun from the repo root; outputs under gpurun_out/, summaries into profiles/ with scripts/collect_profiles_r06.sh):
#   full GPU tests with durations, driver-style and default bench lines (the default line carries the theta_clamp and churn legs), profile sets of the headline (N = 200) and of
#   configs4 (N = 500, fp32 storage; counter passes on the memory-tile form: --pmc-serial), the kernel trace of the step-wise API, the split fold and the gain fold A/B,
#   the pivot's panel stamps at N = 500 with and without the split fold, the mixed downdate in the rank-aware form over 3 000 frames, the map-change timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
[ -n "$SKIP_PYTEST" ] || { timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_pytest.log; }
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style.json 2> gpurun_out/r06_bench.err
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
timeout 900 bash scripts/profile_round.sh r06_a > gpurun_out/r06_profile_n200.log 2>&1
PROFILE_STEPS=40 PROFILE_WARMUP=6 timeout 900 bash scripts/profile_round.sh r06_n500 --landmarks 500 --storage f32 > gpurun_out/r06_profile_n500.log 2>&1
timeout 600 bash scripts/profile_step.sh r06_step 200 > gpurun_out/r06_profile_step.log 2>&1
FRAMES=60 timeout 300 python scripts/split_fold_check.py 400 500:f32 600 > gpurun_out/r06_split_fold.txt 2>&1
timeout 200 python scripts/fold_bench.py > gpurun_out/r06_gain_fold.txt 2>&1
for m in 0 1; do echo "== split_fold $m"; timeout 200 python scripts/persist_stamps.py 500 24 split_fold=$m 2>&1 | grep -v amdgpu.ids | head -26; done > gpurun_out/r06_split_fold_stamps.txt
[ -n "$SKIP_MIXED" ] || timeout 900 python scripts/mixed_rank_study.py > gpurun_out/r06_mixed_rank.log 2>&1
timeout 200 bash scripts/churn_probe.sh 200 200 > gpurun_out/r06_churn.txt 2>&1; grep "map timing" gpurun_out/churn_timing.txt | tail -40 >> gpurun_out/r06_churn.txt
bash scripts/collect_profiles_r06.sh > gpurun_out/r06_collect.log 2>&1
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06_* gpurun_out/profiles_r06/
rm -rf gpurun_out/r06_*_stats gpurun_out/r06_*_fetch gpurun_out/r06_*_write gpurun_out/r06_*_mfma
du -sh gpurun_out
tail -n 3 gpurun_out/r06_pytest.log; tail -c 400 gpurun_out/r06_bench_driver_style.json; cat gpurun_out/r06_split_fold.txt | grep -v amdgpu
