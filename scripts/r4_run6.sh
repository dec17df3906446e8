#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_b8 -- python3 scripts/batch_probe.py 8 1 > gpurun_out/r4f_b8.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4f_b8/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms  {r['Percentage']}%")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_b16 -- python3 scripts/batch_probe.py 16 1 > gpurun_out/r4f_b16.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4f_b16/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms  {r['Percentage']}%")
PY
