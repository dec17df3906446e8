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

import re


def kname(raw):
    """Kernel_Name as rocprofv3 prints it -> bare kernel name: no return type, template or argument list."""
    k = raw.split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    if k.startswith("k_rank_expand<"):                         # <0> (first frame of a fresh state) and <2> (the frame tail that also projects) are different kernels
        return k
    return re.sub(r"<.*>$", "", k)


tag, stats, fetch, write = sys.argv[1:5]
workload = sys.argv[6] if len(sys.argv) > 6 else "bench.py N=200"
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = kname(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


f, w = agg(fetch), agg(write)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches: bench.py --eager), " + workload,
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


# optional 5th argument: counter_collection.csv of the MFMA / VALU pass -> profiles/<tag>_mfma.json
#   MfmaUtil % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) * 100   (rocprofv3's own derived-metric formula)
#   mfma_tflops = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 flop / kernel duration (duration from the kernel-stats average)
if len(sys.argv) > 5:
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(sys.argv[5])):
        k = kname(r["Kernel_Name"])
        c = per[k][r["Counter_Name"]]
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[kname(r["Name"])] = float(r["AverageNs"])
    res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE "
                     "(own pass, eager launches), " + workload + "; per launch averages",
           "formulas": {"mfma_util_pct": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024) * 100", "mfma_gflop": "SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 / 1e9",
                        "mfma_tflops": "mfma_gflop / kernel-trace average duration", "peak_fp64_mfma_tflops": 78.6,
                        "mfma_busy_pct_of_kernel_time": "SQ_VALU_MFMA_BUSY_CYCLES / (kernel-trace duration * 2.4 GHz * 1024 SIMDs) * 100"}, "kernels": {}}
    for k, cs in sorted(per.items()):
        if not k.startswith("k_"):
            continue
        avg = {c: v[1] / max(v[0], 1) for c, v in cs.items()}
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
        fl = avg.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512
        e = {"launches_sampled": max(v[0] for v in cs.values()), **{c: round(v, 1) for c, v in avg.items()},
             "mfma_util_pct": round(100.0 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 2) if gui else None,
             "mfma_gflop": round(fl / 1e9, 4)}
        if k in dur and dur[k] > 0:
            e["avg_duration_us"] = round(dur[k] / 1e3, 2)
            e["mfma_tflops"] = round(fl / dur[k] / 1e3, 2)
            e["mfma_frac_of_peak"] = round(fl / dur[k] / 1e3 / 78.6, 4)
            # busy cycles of the matrix pipes over the kernel's own duration (2.4 GHz x 1024 SIMDs); GRBM_GUI_ACTIVE of a
            # counter pass includes the profiler's serialisation around the dispatch and overstates the denominator
            e["mfma_busy_pct_of_kernel_time"] = round(100.0 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur[k] * 2.4 * 1024), 2)
        res["kernels"][k] = e
    json.dump(res, open(f"profiles/{tag}_mfma.json", "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))
