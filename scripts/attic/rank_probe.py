"""Rank-aware refactorisation probe (GPU box): parity against the oracle and frames/s with SRUKF_RANK_AWARE on / off."""
import sys, os, time
sys.path.insert(0, ".")
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
from oracle import oracle as O
mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
if mode == "parity":
    for N, F in ((20, 6), (50, 3), (100, 2)):
        p = synth.scene_params(); sc = synth.make_scene(N, F, seed=3, p=p)
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
        o = O.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
        for t in range(F):
            f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); o.predict_motion(sc["odo"][t], sc["odo"][t + 1])
            f.predict_measurement(); o.predict_measurement()
            f.update(sc["z"][t], sc["matched"][t]); o.update(sc["z"][t], sc["matched"][t], 1, 0, 1)
            X, S = f.get_state(); Xo, So = o.get_state()
            print(N, t, "|dX|", np.abs(X - Xo).max(), "|dP|", np.abs(S.T @ S - So.T @ So).max(), "tril", np.abs(np.tril(S, -1)).max(), "diag min", np.diag(S).min(), flush=True)
else:
    for N in (100, 200, 300, 500):
        F = 100 if N <= 200 else 40
        p = synth.scene_params(); sc = synth.make_scene(N, F, seed=0, p=p)
        f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
        f.run_frames(0, F); best = 1e9
        for rep in range(3):
            f.set_state(sc["X0"], sc["S0"]); t = time.perf_counter(); tr = f.run_frames(0, F); best = min(best, time.perf_counter() - t)
        print(N, "RANK_AWARE", os.environ.get("SRUKF_RANK_AWARE", "1"), round(F / best, 1), "frames/s; err vs truth", np.abs(tr[:, :2] - sc["odo"][1:F + 1, :2]).max(), "clamp", f.clamp_info(), flush=True)
        f.close()
