"""Stress of bench.py's multi-sequence leg: B = 3 filters in SRUKF_GPU_SHARED beside an idle main filter, repeated; prints the
aggregate rate of every repetition and what was flagged (an abandoned persistent launch shows as a rate ten times lower)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, K, W, B = 200, int(sys.argv[2]) if len(sys.argv) > 2 else 100, 10, 3

sc0 = synth.make_scene(N, 40, seed=0, p=synth.scene_params())
ts = torch.cuda.Stream()
main = srukf.Filter(N, sc0["params"], device=0, stream=ts.cuda_stream); main.set_state(sc0["X0"], sc0["S0"]); main.stage_sequence(sc0["odo"], sc0["z"], sc0["matched"])
main.run_frames(0, 30)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    fs = []
    for b in range(B):
        sc = synth.make_scene(N, W + K, seed=0, p=synth.scene_params(), obs_seed=5000 + b)
        f = srukf.Filter(N, sc["params"], device=0); f.set_exclusive(srukf.GPU_SHARED)
        if len(sys.argv) > 4: f.debug_set("fused_motion", int(sys.argv[4]))
        f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"]); fs.append(f)
    srukf.run_frames_batch(fs, 0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k0 in range(0, K, 16):
        for f in fs: f.run_frames_async(W + k0, min(16, K - k0))
    tl = time.perf_counter() - t0
    ts_ = []
    for f in (fs if rep % 2 == 0 else fs[::-1]):
        try: f.synchronize()
        except Exception as e: print("   flagged:", str(e)[:120])
        ts_.append(round((time.perf_counter() - t0) * 1e3, 2))
    dt = time.perf_counter() - t0
    info = [{k: f.debug_get(k) for k in ("gmw_aborts", "clamp_rows", "gate_timeouts", "gmw_shared", "frame")} for f in fs]
    print(f"rep {rep}: {B * K / dt:.0f} frames/s aggregate; launch {tl * 1e3:.2f} ms, sync done at {ts_} ms; {info if dt > 0.02 else ''}", flush=True)
    for f in fs: f.close()
