"""Per-kernel times of the step-wise API (separate statistics launches, no graph) at N landmarks: what each stage costs alone."""
import sys
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.scene_params(); F = 24
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"])
for t in range(F):
    if t == 4: f.set_profiling(1); f.profile_reset()
    f.predict_motion(sc["odo"][t], sc["odo"][t + 1]); f.predict_measurement(); f.update(sc["z"][t], sc["matched"][t])
pr = f.profile()
print({k: round(v["ms"] / v["launches"] * 1e3, 2) for k, v in pr.items() if v["launches"]})
