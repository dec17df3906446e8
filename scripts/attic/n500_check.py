"""frames/s at large N (GPU box): python scripts/n500_check.py [N ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
for N in [int(a) for a in sys.argv[1:]] or [500]:
    p = synth.scene_params(); F = 24
    sc = synth.make_scene(N, F, seed=0, p=p)
    out = {}
    for excl in (True, False):
        f = srukf.Filter(N, p); f.set_exclusive(excl); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.run_frames(0, 4); f.set_state(sc["X0"], sc["S0"])
        t = time.perf_counter(); tr = f.run_frames(0, F); dt = time.perf_counter() - t
        out[excl] = (F / dt, tr)
    d = np.abs(out[True][1] - out[False][1]).max()
    print(f"N={N}: persistent {out[True][0]:.1f} frames/s, per-panel {out[False][0]:.1f} frames/s, max traj diff {d:.2e}, err vs truth {np.abs(out[True][1][:, :2] - sc['odo'][1:F+1, :2]).max():.2e}", flush=True)
