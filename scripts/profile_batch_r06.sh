cd "$GRAFT_REPO_ROOT"
timeout 900 bash scripts/profile_batch.sh r06_batch 32 > gpurun_out/r06_batch_profile.log 2>&1
newest() { ls -t $1 2>/dev/null | head -1; }
python scripts/summarize_profiles.py r06_batch "$(newest "gpurun_out/r06_batch_stats/*/*kernel_stats.csv")" "$(newest "gpurun_out/r06_batch_fetch/*/*counter_collection.csv")" "$(newest "gpurun_out/r06_batch_write/*/*counter_collection.csv")" "$(newest "gpurun_out/r06_batch_mfma/*/*counter_collection.csv")" "scripts/batch_probe.py 32 1 200 4: srukf_run_frames_batch, 32 filters at N = 200 in one group per launch (counter passes: eager launches)" > /dev/null
python scripts/trace_gaps.py "$(newest "gpurun_out/r06_batch_stats/*/*kernel_trace.csv")" > profiles/r06_batch_kernel_gaps.txt 2>&1
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06_batch_* gpurun_out/profiles_r06/
rm -rf gpurun_out/r06_batch_stats gpurun_out/r06_batch_fetch gpurun_out/r06_batch_write gpurun_out/r06_batch_mfma
tail -4 gpurun_out/r06_batch_profile.log; ls gpurun_out/profiles_r06 | grep batch
