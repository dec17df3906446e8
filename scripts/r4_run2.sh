#!/bin/bash
# round 4, GPU call 2: full GPU tests (null_canonical logic), N = 500 A/B of the memory-tile early2 variants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r4b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4b_pytest.log
tail -5 gpurun_out/r4b_pytest.log
for v in "" cv-monoslam_amd/libsrukf_hip_GMW_MEM_EARLY2_1.so cv-monoslam_amd/libsrukf_hip_GMW_MEM_EARLY2_2.so; do
  echo "== variant ${v:-default(0)}" >> gpurun_out/r4b_n500.log
  for rep in 1 2; do SRUKF_LIB=${v:+$PWD/$v} python scripts/n500_profile.py 500 >> gpurun_out/r4b_n500.log 2>&1; done
done
cat gpurun_out/r4b_n500.log
