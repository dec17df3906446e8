"""How well conditioned is each kept pivot?  D_a (the pivot of the refactorisation = the conditional variance of direction a given the directions before it) over
P_aa (its marginal variance) for the fp64 filter after a few frames: a product S^T S formed with relative error eps can only resolve pivots with D / P >> eps.
  python scripts/pivot_ratio_probe.py [N] [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
F = int(sys.argv[2]) if len(sys.argv) > 2 else 30
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
n = f.n
S0 = sc["S0"]
en = (np.triu(S0) ** 2).sum(axis=1)
kept = [k for k in range(n) if not (k < n - 4 and en[k] < 1e-12)]
r = int(f.debug_get("plan_kept"))
assert r == len(kept), (r, len(kept))
names = ["xi", "yi", "zi", "theta", "phi", "rho"]
for t in (1, 5, F - 1):
    f.set_state(sc["X0"], sc["S0"])
    f.run_frames(0, t)
    X, S = f.get_state()                       # state BEFORE frame t
    Pd = (np.triu(S) ** 2).sum(axis=0)
    f.run_frames(t, 1)
    D = f.debug_copy("D", n)[:r]
    ratio = D / Pd[kept]
    order = np.argsort(ratio)
    print(f"frame {t}: kept {r}; D/P below 1e-7: {(ratio < 1e-7).sum()}, 1e-6: {(ratio < 1e-6).sum()}, 1e-5: {(ratio < 1e-5).sum()}, 1e-4: {(ratio < 1e-4).sum()}, 1e-3: {(ratio < 1e-3).sum()}")
    for a in order[:14]:
        k = kept[a]
        what = f"robot[{k - (n - 4)}]" if k >= n - 4 else f"landmark {k // 6} {names[k % 6]}"
        print(f"    permuted {a:5d} state {k:5d} {what:22s} D = {D[a]:.3e}  P = {Pd[k]:.3e}  D/P = {ratio[a]:.2e}")
f.close()
