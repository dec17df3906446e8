#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4i_b32 -- python3 scripts/batch_probe.py 32 1 200 1 > gpurun_out/r4i_b32.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4i_b32/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms  {r['Percentage']}%")
PY
