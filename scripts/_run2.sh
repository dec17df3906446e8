cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in 266 267 270; do timeout 120 python scripts/abort_probe.py $n 1 4; done > gpurun_out/abort_probe.log 2>&1
timeout 120 python scripts/abort_probe.py 270 0 4 >> gpurun_out/abort_probe.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -s > gpurun_out/r5_pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r5_pytest_new.log
timeout 300 python scripts/split_replay.py record 500 /tmp/split_record.npz 4 > gpurun_out/split_record.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r5_b.json 2> gpurun_out/bench_r5_b.err
tail -n 30 gpurun_out/abort_probe.log; tail -n 40 gpurun_out/r5_pytest_new.log; tail -n 5 gpurun_out/split_record.log; tail -c 3000 gpurun_out/bench_r5_b.json
