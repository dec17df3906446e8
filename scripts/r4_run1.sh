#!/bin/bash
# round 4, GPU call 1: tests, driver-style bench line, N = 500 profile set, tenants probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a_pytest.log
tail -3 gpurun_out/r4a_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_driver_style.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
PROFILE_STEPS=40 PROFILE_WARMUP=6 bash scripts/profile_round.sh r04_n500 --landmarks 500 --storage f32 > gpurun_out/r4a_profile_n500.log 2>&1
python scripts/tenants_probe.py > gpurun_out/r4a_tenants.log 2>&1
tail -30 gpurun_out/r4a_tenants.log
