"""Per-kernel duration and the idle gap to the next kernel, from a rocprofv3 --kernel-trace CSV.
  python scripts/trace_gaps.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0]
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gap[k].append(g)
print(f"{'kernel':28s} {'n':>7s} {'dur_us':>8s} {'gap_after_us':>12s}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    print(f"{k:28s} {len(dur[k]):7d} {sum(dur[k]) / len(dur[k]) / 1e3:8.2f} {sum(gap[k]) / max(len(gap[k]), 1) / 1e3:12.2f}")
