"""Where does the full-rank path get flagged in a long N = 500 replay, and what does the state look like there?"""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, F = 500, 2000
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_rank_aware(False); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
g = srukf.Filter(N, p); g.set_state(sc["X0"], sc["S0"]); g.stage_sequence(sc["odo"], sc["z"], sc["matched"])
b = 0
while b < F:
    Xp, Sp = f.get_state()
    try:
        f.run_frames_async(b, 50); f.synchronize()
    except Exception as e:
        fr, row = f.clamp_info()
        print("flagged:", e, "frame", fr, "row", row, "(landmark", row // 6, "component", row % 6, ")")
        X, S = Xp, Sp                                   # state before the block
        d = np.diag(S); e_ = np.sum(S * S, axis=1)
        null = [6 * k + c for k in range(1, N) for c in range(3)]
        print("state at block start", b, ": min/max diag S", d.min(), d.max(), "| null rows: max energy", e_[null].max(), " max |offdiag| in null rows", np.abs(S[null] - np.diag(d)[null]).max())
        Xg, Sg = g.get_state()
        print("rank-aware filter at the same frame: max |dX|", np.abs(Xg - X).max(), " max |dP|", np.abs(Sg.T @ Sg - S.T @ S).max())
        P = S.T @ S
        print("row", row, ": S diag", d[row], " row energy", e_[row], " P[row,row]", P[row, row])
        break
    g.run_frames_async(b, 50); g.synchronize()
    b += 50
else:
    print("no frame flagged in", F)
