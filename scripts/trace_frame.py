"""Start / end of every kernel of one steady frame from a rocprofv3 kernel trace: python scripts/trace_frame.py <kernel_trace.csv> [first kernel name prefix]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_pxy2"
i0 = int(len(rows) * 0.7)
while not rows[i0]["Kernel_Name"].replace("void ", "").startswith(first): i0 += 1
t0 = int(rows[i0]["Start_Timestamp"])
n = 0
for r in rows[i0:]:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if nm.startswith(first): n += 1
    if n > 2: break
    print(f"{nm:28s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f}  end {(int(r['End_Timestamp']) - t0) / 1e3:8.1f}  queue {r.get('Queue_Id', '?')}")
