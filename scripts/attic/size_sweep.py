"""frames/s of the staged replay over the BASELINE landmark counts (GPU box); one JSON line."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
out = {}
for N in (20, 50, 100, 200, 300, 400, 500):
    F = 200 if N <= 200 else 60
    p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, F); f.set_state(sc["X0"], sc["S0"])     # full-length warm-up: graph capture, first launches, pool growth
    dt = 1e9
    for rep in range(3):                                    # best of three: the first long run of a context can carry a one-off allocator hiccup
        f.set_state(sc["X0"], sc["S0"])
        t = time.perf_counter(); tr = f.run_frames(0, F); dt = min(dt, time.perf_counter() - t)
    f.close()
    out[str(N)] = {"state_dim": 6 * N + 4, "frames_per_s": round(F / dt, 1), "pose_err_vs_truth_max_m": float(np.abs(tr[:, :2] - sc["odo"][1:F + 1, :2]).max())}
print(json.dumps(out))
