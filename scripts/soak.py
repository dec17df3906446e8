"""Long replay at the benchmark size: 3000 frames (the reference's CAPACITY), checks finiteness, the theta-clamp flag and the
pose error against the noise-free odometry; prints frames/s per 500-frame block."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, F = 200, 3000
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames(0, 8); f.set_state(sc["X0"], sc["S0"])
tr = []
for b in range(0, F, 500):
    t = time.perf_counter(); tr.append(f.run_frames(b, 500)); dt = time.perf_counter() - t
    print(f"frames {b}-{b+499}: {500/dt:.0f} frames/s, pose err vs truth max {np.abs(tr[-1][:, :2] - sc['odo'][b+1:b+501, :2]).max():.2e}")
X, S = f.get_state()
print("finite", np.isfinite(X).all() and np.isfinite(S).all(), "| min/max diag S", np.diag(S).min(), np.diag(S).max())
