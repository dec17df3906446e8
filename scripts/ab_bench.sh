#!/bin/bash
# alternates the product library and a variant in ONE gpurun call: frames/s of the headline workload (bench.py --steps 300) and the persistent launch's time
# (eager leg, HIP events), three rounds:   bash scripts/ab_bench.sh build/variants/libsrukf_hip_HEAD.so [extra bench.py flags]
lib=$1; shift
for i in 1 2 3; do for l in "" "$lib"; do
  v=$(python bench.py ${l:+--lib $l} --steps 300 --warmup 20 --profile-frames 40 --repetitions 3 --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 --no-step-api --no-theta-clamp "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d.get('value_repetitions'), '; k_gmw_persist', d['kernels_us_per_frame'].get('k_gmw_persist'), 'us')")
  echo "round $i lib=${l:-product}: $v"
done; done
