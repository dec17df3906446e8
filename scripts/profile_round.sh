#!/bin/bash
# Collects the judged profile set on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh <tag> [bench.py arguments that select the workload, e.g. --landmarks 500 --storage f32]
# 1. rocprofv3 --kernel-trace --stats of the default bench workload (graph replay)
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE) with eager launches (bench.py --eager), counters only; --pmc-serial: counter passes serialise the dispatches, so the
#    split form of the factorisation (N >= 400: two launches that wait for each other) is switched off in them
# 3. one PMC pass with the MFMA / VALU utilisation counters
# Outputs land in gpurun_out/<tag>_*; scripts/summarize_profiles.py turns them into profiles/.
tag=$1; shift; wl="$@"
st=${PROFILE_STEPS:-100}; wu=${PROFILE_WARMUP:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py $wl --steps $st --warmup $wu --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 --no-step-api --no-theta-clamp > gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py $wl --eager --pmc-serial --steps 12 --warmup 4 --profile-frames 4 --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 --no-step-api --no-theta-clamp > gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py $wl --eager --pmc-serial --steps 12 --warmup 4 --profile-frames 4 --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 --no-step-api --no-theta-clamp > gpurun_out/${tag}_write.log 2>&1
# 3. MFMA / VALU utilisation counters (their own pass; SQ block: 8 slots, GRBM: 2)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_mfma -- python3 bench.py $wl --eager --pmc-serial --steps 12 --warmup 4 --profile-frames 4 --no-cpu-baseline --sequences-per-gpu 0 --no-collectives-check --no-configs4 --no-step-api --no-theta-clamp > gpurun_out/${tag}_mfma.log 2>&1
ls gpurun_out/${tag}_stats/*/ gpurun_out/${tag}_fetch/*/ gpurun_out/${tag}_write/*/ gpurun_out/${tag}_mfma/*/
# (--no-step-api --no-theta-clamp above: the step_api leg's child processes would inherit the profiler's preload — ~1 500 counter-profiled step frames per pass and their own
#  <pid>_kernel_trace.csv files next to the parent's; the newest trace is the parent's in any case)
python3 scripts/trace_gaps.py $(ls -t gpurun_out/${tag}_stats/*/*kernel_trace.csv | head -1) | head -14
