"""Soak of the memory-tile persistent factorisation (N >= 340): 300 frames at N = 400 and 200 at N = 500; finiteness, clamp
flag, pose error, frames/s; the run_frames recovery path would show as a frames/s collapse."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N, F in ((400, 300), (500, 200)):
    p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
    f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    f.run_frames(0, 4); f.set_state(sc["X0"], sc["S0"])
    for b in range(0, F, 100):
        t = time.perf_counter(); tr = f.run_frames(b, 100); dt = time.perf_counter() - t
        print(f"N={N} frames {b}-{b+99}: {100/dt:.0f} frames/s, pose err vs truth max {np.abs(tr[:, :2] - sc['odo'][b+1:b+101, :2]).max():.2e}", flush=True)
    X, S = f.get_state()
    print("finite", bool(np.isfinite(X).all() and np.isfinite(S).all()), "clamp info", f.clamp_info())
