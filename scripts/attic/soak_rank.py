"""Long replay at the benchmark size with the rank-aware refactorisation against the full-rank path: 3000 frames (the
reference's CAPACITY), the same staged inputs through both; prints the largest pose / robot-covariance difference per 500-frame
block, whether any frame was flagged (theta clamp or null-direction check -> srukf_run_frames repeats it on the exact path and
the aborts counter moves) and frames/s."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
N, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), 3000
p = synth.scene_params()
sc = synth.make_scene(N, F, seed=0, p=p)
fs = []
for on in (True, False):
    f = srukf.Filter(N, p); f.set_rank_aware(on); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    fs.append(f)
print("null directions skipped:", fs[0].null_directions(), "of", 6 * N + 4)
for b in range(0, F, 500):
    tr = []
    for f in fs:
        t = time.perf_counter(); tr.append(f.run_frames(b, 500)); tr.append(500 / (time.perf_counter() - t))
    d = np.abs(tr[0] - tr[2])
    print(f"frames {b}-{b+499}: rank-aware {tr[1]:.0f} / full {tr[3]:.0f} frames/s | max |dpose| {d[:, :4].max():.2e}  max |dP_robot| {d[:, 4:].max():.2e}"
          f" | pose err vs truth {np.abs(tr[0][:, :2] - sc['odo'][b+1:b+501, :2]).max():.2e} | clamp info {fs[0].clamp_info()}")
Xa, Sa = fs[0].get_state(); Xb, Sb = fs[1].get_state()
print("final: max |dX|", np.abs(Xa - Xb).max(), " max |dP|", np.abs(Sa.T @ Sa - Sb.T @ Sb).max(), " null directions now", fs[0].null_directions())
