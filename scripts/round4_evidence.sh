#!/bin/bash
# What produced round 4's committed evidence (run through gpurun from the repo root; outputs under gpurun_out/, summaries into profiles/ with
# scripts/summarize_profiles.py and by copying the JSON lines):
#   full GPU tests, driver-style and default bench lines, profile sets of the headline (N = 200), configs4 (N = 500, fp32 storage) and the batched replay,
#   the groups probe and the 3000-frame soak of the batched replay, the 3000-frame soak of the split form at N = 500
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r04_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_style.json 2> gpurun_out/r04_bench.err
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
bash scripts/profile_round.sh r04_a > gpurun_out/r04_profile_n200.log 2>&1
PROFILE_STEPS=40 PROFILE_WARMUP=6 bash scripts/profile_round.sh r04_n500 --landmarks 500 --storage f32 > gpurun_out/r04_profile_n500.log 2>&1
bash scripts/profile_batch.sh r04_batch 32 > gpurun_out/r04_profile_batch.log 2>&1
python scripts/batch_probe.py 8,16,24,32,48 1 200 1,2,3,4 > gpurun_out/r04_batch_probe_groups.txt 2>&1
python scripts/batch_soak.py 32 3000 500 r04_batch_soak > gpurun_out/r04_batch_soak.log 2>&1
python scripts/soak_split.py 3000 r04_split_soak > gpurun_out/r04_split_soak.log 2>&1
