"""Stress of bench.py's multi-sequence leg: B = 3 filters in SRUKF_GPU_SHARED beside an idle main filter, repeated; prints the
aggregate rate of every repetition and what was flagged (an abandoned persistent launch shows as a rate ten times lower)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W, B = 200, 100, 10, 3
sc0 = synth.make_scene(N, 40, seed=0, p=synth.scene_params())
ts = torch.cuda.Stream()
main = srukf.Filter(N, sc0["params"], device=0, stream=ts.cuda_stream); main.set_state(sc0["X0"], sc0["S0"]); main.stage_sequence(sc0["odo"], sc0["z"], sc0["matched"])
main.run_frames(0, 30)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    fs = []
    for b in range(B):
        sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
        f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    srukf.run_frames_batch(fs, 0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); srukf.run_frames_batch(fs, W, K); dt = time.perf_counter() - t0
    info = [(f.clamp_info(), f.last_error()[:100] if hasattr(f, "last_error") else "") for f in fs]
    print(f"rep {rep}: {B * K / dt:.0f} frames/s aggregate; clamp info / last error per filter: {info}", flush=True)
    for f in fs: f.close()
