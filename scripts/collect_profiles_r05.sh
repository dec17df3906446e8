#!/bin/bash
# Turns what scripts/round5_evidence.sh left under gpurun_out/ into the committed summaries under profiles/ (newest rocprofv3 output of every pass).  Run from the repo root in
# the build container after the gpurun call has merged its files back.
newest() { ls -t $1 | head -1; }
for t in r05_a r05_n500 r05_n500_split; do
  wl="bench.py N=200"
  [ $t = r05_n500 ] && wl="bench.py --landmarks 500 --storage f32 (BASELINE configs[4]; counter passes: --pmc-serial, memory-tile form — the split form's counters: r05_n500_split)"
  [ $t = r05_n500_split ] && wl="scripts/split_replay.py replay 500: k_gmw_pivslab_persist / k_gmw_tiles_persist each replayed ALONE against the operands and flags of one recorded frame of BASELINE configs[4] (N = 500, fp32 storage); durations are the launches' alone, not the pair's"
  python scripts/summarize_profiles.py $t "$(newest "gpurun_out/${t}_stats/*/*kernel_stats.csv")" "$(newest "gpurun_out/${t}_fetch/*/*counter_collection.csv")" \
         "$(newest "gpurun_out/${t}_write/*/*counter_collection.csv")" "$(newest "gpurun_out/${t}_mfma/*/*counter_collection.csv")" "$wl" > /dev/null
  python scripts/trace_gaps.py "$(newest "gpurun_out/${t}_stats/*/*kernel_trace.csv")" > profiles/${t}_kernel_gaps.txt 2>&1
done
python scripts/trace_frame.py "$(newest "gpurun_out/r05_n500_stats/*/*kernel_trace.csv")" > profiles/r05_n500_frame_timeline.txt
cp "$(newest "gpurun_out/r05_step_stats/*/*kernel_stats.csv")" profiles/r05_step_kernel_stats.csv
python scripts/trace_gaps.py "$(newest "gpurun_out/r05_step_stats/*/*kernel_trace.csv")" > profiles/r05_step_kernel_gaps.txt 2>&1
tail -1 gpurun_out/r05_bench_driver_style.json > profiles/r05_a_bench_driver_style.json
tail -1 gpurun_out/r05_bench_default.json > profiles/r05_a_bench.json
cp gpurun_out/r05_native_multi_1gpu.json profiles/r05_native_multi_1gpu.json
cp gpurun_out/r05_n500_split_record.log profiles/r05_n500_split_record.txt 2>/dev/null
ls -la profiles/ | grep r05
cp gpurun_out/r05_f32_curve_n500.json profiles/r05_f32_curve_n500.json
