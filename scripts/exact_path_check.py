"""The exact path's three forms on one matrix: left-looking k_gmw_col (exact_rl 0), right-looking one pivot per launch (exact_rl 2 = B 1), blocked (default: B = 8 at n = 1 204).
Same factor bit for bit between the two right-looking forms; against the left-looking one within rounding.   python scripts/exact_path_check.py [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
synth, srukf = pkg.synth, pkg.srukf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = synth.default_params()                             # the shipped a1 .. a4 = 8: bench.py's theta_clamp scene (the filter it describes diverges after ~10 frames at N = 200: DESIGN.md)
sc = synth.make_scene(N, 12, seed=1, p=p)
res = {}
for name, v in (("left_looking", 0), ("right_looking_b1", 2), ("right_looking_blocked", 1)):
    srukf.debug_set_global("exact_rl", v)
    f = srukf.Filter(N, p)
    f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
    t0 = time.perf_counter(); tr = np.vstack([f.run_frames(t, 1) for t in range(12)]); dt = time.perf_counter() - t0
    X, S = f.get_state()
    res[name] = (X, S, tr)
    print(f"{name}: {12 / dt:.0f} frames/s, exact frames {f.debug_get('exact_frames')}", flush=True)
    f.close()
srukf.debug_set_global("exact_rl", 1)
a, b, c = res["left_looking"], res["right_looking_b1"], res["right_looking_blocked"]
print("blocked == one pivot per launch bit for bit:", all(np.array_equal(x, y) for x, y in zip(b, c)))
for t in range(12):
    print(f"frame {t}: pose left-looking - blocked {np.abs(a[2][t, :4] - c[2][t, :4]).max():.2e}; error vs truth left-looking {np.abs(a[2][t, :2] - sc['odo'][t + 1, :2]).max():.2e}, blocked {np.abs(c[2][t, :2] - sc['odo'][t + 1, :2]).max():.2e}")
