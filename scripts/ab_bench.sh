#!/bin/bash
# alternates the product library and a variant in ONE gpurun call: frames/s of the headline workload (bench.py --steps 300), three rounds
#   bash scripts/ab_bench.sh cv-monoslam_amd/libsrukf_hip_HEAD.so
for i in 1 2 3; do for l in "" "$1"; do
  v=$(SRUKF_LIB=$l python bench.py --steps 300 --warmup 20 --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "round $i lib=${l:-product}: $v frames/s"
done; done
