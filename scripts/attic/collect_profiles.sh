#!/bin/bash
# Turns what scripts/round4_evidence.sh left under gpurun_out/ into the committed summaries under profiles/ (newest rocprofv3 output of every pass: gpurun_out/
# accumulates the outputs of earlier calls).  Run from the repo root in the build container after the gpurun call has merged its files back.
newest() { ls -t $1 | head -1; }
for t in r04_a r04_n500 r04_batch; do
  wl="bench.py N=200"
  [ $t = r04_n500 ] && wl="bench.py --landmarks 500 --storage f32 (BASELINE configs[4]; counter passes: --pmc-serial, memory-tile form)"
  [ $t = r04_batch ] && wl="srukf_run_frames_batch, 32 filters at N=200 (scripts/profile_batch.sh)"
  python scripts/summarize_profiles.py $t "$(newest "gpurun_out/${t}_stats/runc/*kernel_stats.csv")" "$(newest "gpurun_out/${t}_fetch/runc/*counter_collection.csv")" \
         "$(newest "gpurun_out/${t}_write/runc/*counter_collection.csv")" "$(newest "gpurun_out/${t}_mfma/runc/*counter_collection.csv")" "$wl" > /dev/null
  python scripts/trace_gaps.py "$(newest "gpurun_out/${t}_stats/runc/*kernel_trace.csv")" > profiles/${t}_kernel_gaps.txt 2>&1
done
python scripts/trace_frame.py "$(newest "gpurun_out/r04_n500_stats/runc/*kernel_trace.csv")" > profiles/r04_n500_frame_timeline.txt
tail -1 gpurun_out/r04_bench_driver_style.json > profiles/r04_a_bench_driver_style.json
tail -1 gpurun_out/r04_bench_default.json > profiles/r04_a_bench.json
cp gpurun_out/r04_batch_probe_groups.txt profiles/r04_batch_probe_groups.txt
cp gpurun_out/r04_batch_soak.json profiles/r04_batch_soak.json 2>/dev/null
cp gpurun_out/r04_split_soak.json profiles/r04_n500_split_soak.json 2>/dev/null
ls -la profiles/ | grep r04
