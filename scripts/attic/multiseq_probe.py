"""multi-sequence throughput probe: B contexts on one GPU (what bench.py's multi_sequence leg does), with the filters told
that they share the GPU (one launch per panel) and not told (persistent launches from B streams queue behind each other)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
import torch
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W = int(sys.argv[1]) if len(sys.argv) > 1 else 200, 96, 16


def run(B, mode, chunk):
    fs = []
    for b in range(B):
        sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
        f = srukf.Filter(N, sc["params"], device=0)
        f.set_exclusive(mode)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        fs.append(f)
    for f in fs: f.run_frames_async(0, W)
    rc = [f.synchronize_rc() if hasattr(f, "synchronize_rc") else f.synchronize() for f in fs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k0 in range(0, K, chunk):
        for f in fs: f.run_frames_async(W + k0, min(chunk, K - k0))
    flagged = 0
    for f in fs:
        try: f.synchronize()
        except Exception as e: flagged += 1
    dt = time.perf_counter() - t0
    for f in fs: f.close()
    return B * K / dt, flagged


Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 6, 8]
for B in Bs:
    for name, mode in (("GPU_EXCLUSIVE (ungated persistent launches; not safe for B > 1)", srukf.GPU_EXCLUSIVE), ("GPU_SHARED (half the CUs, two admitted)", srukf.GPU_SHARED),
                       ("GPU_SHARED_PER_PANEL", srukf.GPU_SHARED_PER_PANEL)):
        v, fl = run(B, mode, 16)
        print(f"B={B} {name}: {v:.0f} frames/s aggregate, flagged filters {fl}", flush=True)

# the C entry point for B filters against the Python loop above
for B in (2, 3):
    fs = []
    for b in range(B):
        sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
        f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED)
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    srukf.run_frames_batch(fs, 0, W)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        srukf.run_frames_batch(fs, W, K)
        print(f"B={B} srukf_run_frames_batch rep {rep}: {B * K / (time.perf_counter() - t0):.0f} frames/s aggregate", flush=True)
    for f in fs: f.close()
