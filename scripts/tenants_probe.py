"""Aggregate frames/s of B filters in SRUKF_GPU_SHARED at N = 200 against the number of persistent launches that share the GPU."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W = 200, 96, 16
for tenants in (2, 3, 4):
    srukf.debug_set_global("shared_tenants", tenants)
    for B in (3, 4, 6, 8):
        fs = []
        for b in range(B):
            sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
            f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED)
            f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
        srukf.run_frames_batch(fs, 0, W)
        best = 0
        for rep in range(2):
            t0 = time.perf_counter(); srukf.run_frames_batch(fs, W, K); best = max(best, B * K / (time.perf_counter() - t0))
        print(f"tenants={tenants} B={B}: {best:.0f} frames/s aggregate", flush=True)
        for f in fs: f.close()
