#!/bin/bash
# round 4: full GPU tests, driver-style bench line, profile sets (N = 200 headline, N = 500 configs4, batched replay)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4k_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k_pytest.log
tail -4 gpurun_out/r4k_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4k_bench_driver_style.json 2> gpurun_out/r4k_bench.err; echo "bench rc=$?"
python bench.py > gpurun_out/r4k_bench_default.json 2> gpurun_out/r4k_bench_default.err; echo "bench default rc=$?"
bash scripts/profile_round.sh r04_a > gpurun_out/r4k_profile_n200.log 2>&1
PROFILE_STEPS=40 PROFILE_WARMUP=6 bash scripts/profile_round.sh r04_n500 --landmarks 500 --storage f32 > gpurun_out/r4k_profile_n500.log 2>&1
bash scripts/profile_batch.sh r04_batch 32 > gpurun_out/r4k_profile_batch.log 2>&1
