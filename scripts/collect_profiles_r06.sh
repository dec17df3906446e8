#!/bin/bash
# Turns what scripts/round6_evidence.sh left under gpurun_out/ into the committed summaries under profiles/ (newest rocprofv3 output of every pass).  Runs on the GPU box at the
# end of the evidence script (the raw counter CSVs are larger than what gpurun merges back) and can be repeated in the build container.
newest() { ls -t $1 2>/dev/null | head -1; }
for t in r06_a r06_n500; do
  wl="bench.py N=200"
  [ $t = r06_n500 ] && wl="bench.py --landmarks 500 --storage f32 (BASELINE configs[4]; kernel trace: the default plan — split form with the split fold; counter passes: --pmc-serial, memory-tile form)"
  python scripts/summarize_profiles.py $t "$(newest "gpurun_out/${t}_stats/*/*kernel_stats.csv")" "$(newest "gpurun_out/${t}_fetch/*/*counter_collection.csv")" \
         "$(newest "gpurun_out/${t}_write/*/*counter_collection.csv")" "$(newest "gpurun_out/${t}_mfma/*/*counter_collection.csv")" "$wl" > /dev/null
  python scripts/trace_gaps.py "$(newest "gpurun_out/${t}_stats/*/*kernel_trace.csv")" > profiles/${t}_kernel_gaps.txt 2>&1
done
python scripts/trace_frame.py "$(newest "gpurun_out/r06_n500_stats/*/*kernel_trace.csv")" > profiles/r06_n500_frame_timeline.txt
cp "$(newest "gpurun_out/r06_step_stats/*/*kernel_stats.csv")" profiles/r06_step_kernel_stats.csv
python scripts/trace_gaps.py "$(newest "gpurun_out/r06_step_stats/*/*kernel_trace.csv")" > profiles/r06_step_kernel_gaps.txt 2>&1
tail -1 gpurun_out/r06_bench_driver_style.json > profiles/r06_a_bench_driver_style.json
tail -1 gpurun_out/r06_bench_default.json > profiles/r06_a_bench.json
grep -v amdgpu.ids gpurun_out/r06_split_fold.txt > profiles/r06_split_fold.txt
grep -v amdgpu.ids gpurun_out/r06_gain_fold.txt > profiles/r06_gain_fold.txt
cp gpurun_out/r06_split_fold_stamps.txt profiles/r06_split_fold_stamps.txt
cp gpurun_out/r06_mixed_rank_n500.json profiles/r06_mixed_rank_n500.json 2>/dev/null
cp gpurun_out/r06_churn.txt profiles/r06_churn.txt
grep -E "passed|failed|slowest|^[0-9.]+s " gpurun_out/r06_pytest.log > profiles/r06_pytest_durations.txt
ls -la profiles/ | grep r06
