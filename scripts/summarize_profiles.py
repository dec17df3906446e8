"""Turns the rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.
  python scripts/summarize_profiles.py <tag> <kernel_stats.csv> <pmc_fetch.csv> <pmc_write.csv>
HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KB
(x1024), collected in separate --pmc passes, and FETCH_SIZE reads 1/2 of the bytes of a coalesced
streaming read on gfx950, so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes."""
import collections
import csv
import json
import shutil
import sys

tag, stats, fetch, write = sys.argv[1:5]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


f, w = agg(fetch), agg(write)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches: SRUKF_NO_GRAPH=1), bench.py N=200",
       "unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": {}}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fk = f[k][1] / max(f[k][0], 1)
    wk = w[k][1] / max(w[k][0], 1)
    out["kernels"][k] = {"launches_sampled": f[k][0], "fetch_size_kb": round(fk, 2), "write_size_kb": round(wk, 2),
                         "traffic_bytes": round((2 * fk + wk) * 1024)}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
