#!/bin/bash
# Counter passes on the eager replay for the contraction kernels: where k_pxy / k_syrk / k_project spend their cycles.
#   bash scripts/pmc_contractions.sh <tag>     (through gpurun; outputs under gpurun_out/<tag>_pmcX)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"

B="python3 bench.py --eager --no-collectives-check --no-configs4 --steps 12 --warmup 4 --profile-frames 4 --no-cpu-baseline --sequences-per-gpu 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc1 -- $B > gpurun_out/${tag}_pmc1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d gpurun_out/${tag}_pmc2 -- $B > gpurun_out/${tag}_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d gpurun_out/${tag}_pmc3 -- $B > gpurun_out/${tag}_pmc3.log 2>&1
python3 - <<PY
import csv, glob, collections
for q in (1, 2, 3):
    fs = glob.glob("gpurun_out/${tag}_pmc%d/*/*counter_collection.csv" % q)
    if not fs: print("pass", q, "no output"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        c = per[k][r["Counter_Name"]]; c[0] += 1; c[1] += float(r["Counter_Value"])
    for k in ("k_pxy", "k_syrk", "k_project", "k_gmw_persist<false, false>", "k_gain", "k_rank_expand"):
        if k in per: print(q, k, {c: round(v[1] / v[0]) for c, v in per[k].items()})
PY
